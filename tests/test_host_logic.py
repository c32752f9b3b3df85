"""CPU: scene compiler, host-side Pose mirror, C-ABI library surface, error mapping.  No GPU compute calls."""

import ctypes as C
import os
import re

import numpy as np
import pytest

import rcs_oracle as O
from parity_util import ROOT, SCENE
from rcs_amd import _lib, common
from rcs_amd.mjcf import MjcfError, compile_mjcf

REF_SCENE = "/root/reference/assets/scenes/fr3_empty_world/scene.xml"


@pytest.fixture(scope="module")
def cm():
    return compile_mjcf(SCENE)


def test_scene_sizes_and_names(cm):
    assert (cm.nbody, cm.njnt, cm.nq, cm.nv, cm.nu, cm.ntendon, cm.neq) == (14, 9, 9, 9, 8, 1, 1)
    assert cm.ngeom == 24 and cm.ncam == 2
    for name in [f"fr3_joint{i}_0" for i in range(1, 8)] + ["finger_joint1_0", "finger_joint2_0"]:
        assert cm.name2id("jnt", name) >= 0
    for name in [f"fr3_link{i}_collision_0" for i in range(8)] + ["hand_c_0", "d435i_collision_0", "finger_0_left_0", "finger_0_right_0", "floor"]:
        assert cm.name2id("geom", name) >= 0
    assert cm.name2id("site", "attachment_site_0") >= 0 and cm.name2id("body", "base_0") >= 0
    assert cm.name2id("actuator", "actuator8_0") == 7 and cm.name2id("cam", "wrist_0") >= 0


def test_scene_constants_match_appendix_a(cm):
    assert cm.timestep == 0.002 and cm.integrator == "implicitfast" and cm.cone == "elliptic"
    assert cm.impratio == 20 and cm.noslip_iterations == 5
    assert np.allclose(cm.gravity, [0, 0, -9.81])
    j = cm.arrays
    assert np.allclose(j["jnt_range"][:7], [[-2.7437, 2.7437], [-1.7837, 1.7837], [-2.9007, 2.9007], [-3.0421, -0.1518],
                                            [-2.8065, 2.8065], [0.5445, 4.5169], [-3.0159, 3.0159]])
    assert np.allclose(j["jnt_range"][7:], [[0, 0.04], [0, 0.04]]) and list(j["jnt_type"]) == [3] * 7 + [2, 2]
    assert np.allclose(j["jnt_actfrcrange"][:7, 1], [87, 87, 87, 87, 12, 12, 12]) and list(j["jnt_actgravcomp"]) == [1] * 7 + [0, 0]
    assert np.allclose(j["dof_armature"], 0.1) and np.allclose(j["dof_damping"], 1.0)
    assert np.allclose(j["actuator_gainprm"][:7, 0], [4500, 4500, 3500, 3500, 2000, 2000, 2000])
    assert np.allclose(j["actuator_biasprm"][:7, 1], -j["actuator_gainprm"][:7, 0])
    assert np.allclose(j["actuator_biasprm"][:7, 2], [-450, -450, -350, -350, -200, -200, -200])
    # position actuators inherit the joint range as ctrlrange; the gripper keeps 0..255 and an affine bias
    assert np.allclose(j["actuator_ctrlrange"][:7], j["jnt_range"][:7]) and np.allclose(j["actuator_ctrlrange"][7], [0, 255])
    assert list(j["actuator_biastype"]) == [1] * 8 and np.allclose(j["actuator_biasprm"][7], [0, -100, -10])
    assert np.allclose(j["actuator_gainprm"][7, 0], 0.01568627451) and np.allclose(j["actuator_forcerange"][7], [-100, 100])
    assert np.allclose(j["eq_solimp"][0, :3], [0.95, 0.99, 0.001]) and np.allclose(j["eq_solref"][0], [0.005, 1])
    assert np.allclose(j["body_mass"][3:10], [2.92747, 2.93554, 2.2449, 2.6156, 2.32712, 1.81704, 0.627143])
    assert np.allclose(j["body_gravcomp"][1:], 1.0) and np.allclose(j["wrap_prm"], [0.5, 0.5])
    # quaternions are normalised by the compiler: "1 -1 0 0" -> 90 deg about -x
    assert np.allclose(j["body_quat"][4], np.array([1, -1, 0, 0]) / np.sqrt(2))


@pytest.mark.skipif(not os.path.exists(REF_SCENE), reason="reference checkout not present (GPU box)")
def test_own_scene_equals_reference_scene_tables():
    """The repository's physics-only scene compiles to the same tables as the reference's MJCF (include, default
    classes, childclass, actuator-default inheritance), except the one documented approximation (d435i_0 inertia)."""
    own, ref = compile_mjcf(SCENE), compile_mjcf(REF_SCENE)
    skip_rows = {"body_ipos": [11], "body_mass": [11], "body_inertia": [11], "body_iquat": [11]}
    for key, a in own.arrays.items():
        if key.startswith(("geom_", "cam_", "mesh_")):
            continue
        b = ref.arrays[key]
        assert a.shape == b.shape, key
        mask = np.ones(a.shape[0], dtype=bool) if a.ndim else None
        for r in skip_rows.get(key, []):
            mask[r] = False
        assert np.allclose(a[mask], b[mask]), key
    assert own.jnt_names == ref.jnt_names and own.actuator_names == ref.actuator_names and own.body_names == ref.body_names
    assert set(own.geom_names) <= set(ref.geom_names) | {"camera_mount_collision_0"}


XARM7_SCENE = os.path.join(os.path.dirname(SCENE), "..", "xarm7_empty_world", "scene.xml")
REF_XARM7_SCENE = "/root/reference/assets/scenes/xarm7_empty_world/scene.xml"


def test_xarm7_scene_constants():
    """The xArm7 scene's physics constants as the reference model states them (assets/xarm7/mjcf/xarm7.xml:45-61,161-169)."""
    cm = compile_mjcf(XARM7_SCENE)
    a = cm.arrays
    assert (cm.nbody, cm.njnt, cm.nu, cm.neq, cm.ntendon) == (9, 7, 7, 0, 0)
    assert np.all(a["dof_frictionloss"] == 1.0) and np.all(a["dof_armature"] == 0.1)
    assert a["dof_damping"].tolist() == [10, 10, 5, 5, 5, 2, 2]
    assert a["actuator_gainprm"][:, 0].tolist() == [1500, 1500, 1000, 1000, 1000, 800, 800]
    assert np.all(a["actuator_biasprm"][:, 1] == -a["actuator_gainprm"][:, 0]) and np.all(a["actuator_biastype"] == 1)
    assert a["actuator_forcerange"][:, 1].tolist() == [50, 50, 30, 30, 30, 20, 20] and np.all(a["actuator_forcelimited"] == 1)
    assert np.allclose(a["jnt_range"][1], [-2.059, 2.0944]) and np.allclose(a["actuator_ctrlrange"][3], [-0.19198, 3.927])
    assert np.all(a["jnt_actgravcomp"] == 1) and np.all(a["body_gravcomp"][1:] == 1.0)
    assert np.allclose(a["dof_solref"], [0.02, 1.0]) and np.allclose(a["dof_solimp"], [0.9, 0.95, 0.001, 0.5, 2.0])
    assert cm.jnt_names == [f"joint{i}" for i in range(1, 8)] and cm.actuator_names == [f"act{i}" for i in range(1, 8)]


@pytest.mark.skipif(not os.path.exists(REF_XARM7_SCENE), reason="reference checkout not present (GPU box)")
def test_own_xarm7_scene_equals_reference_scene_tables():
    own, ref = compile_mjcf(XARM7_SCENE), compile_mjcf(REF_XARM7_SCENE)
    for key, a in own.arrays.items():
        if key.startswith(("geom_", "cam_", "mesh_")):
            continue
        b = ref.arrays[key]
        assert a.shape == b.shape, key
        rows = slice(1, None) if key.startswith("body_i") or key in ("body_mass", "body_inertia") else slice(None)
        # (row 0: the reference's world body carries the mass of its pedestal cylinder geom; static, never read)
        assert np.allclose(a[rows], b[rows]), key
    assert own.jnt_names == ref.jnt_names and own.actuator_names == ref.actuator_names and own.body_names == ref.body_names


def test_oracle_dry_friction_rows():
    """Properties of the Huber friction rows (oracle/rcs_physics.c solve_constraints) on the xArm7 model:
    the constraint force on a dof never exceeds frictionloss, saturates against a fast-moving joint, and an arm at rest
    under gravity compensation stays at rest when commanded to stay (stiction)."""
    from rcs_env_oracle import XARM7

    cm = compile_mjcf(XARM7_SCENE)
    s = O.Sim(cm, XARM7["joints"], XARM7["actuators"], XARM7["site"], XARM7["base"], XARM7["q_home"], None, arm_collision_geoms=[])
    s.reset()
    s.robot_reset()
    for _ in range(50):
        s.step(1)
        assert np.all(np.abs(np.asarray(s.s.d.qfrc_constraint[:7])) <= 1.0 + 1e-12)
    q_rest = np.asarray(s.qpos).copy()
    s.step(200)
    assert np.abs(np.asarray(s.qpos) - q_rest).max() < 1e-4 and np.abs(np.asarray(s.qvel)).max() < 1e-4
    # every row obeys the Huber law force = clip(-D * (qacc - aref), +-frictionloss), and a fast joint saturates it
    for i in range(7):
        s.s.d.qvel[i] = 0.5 * (-1) ** i
    s.step(1)
    dd = s.s.d
    assert dd.nefc == 7 and all(dd.efc_type[i] == 2 for i in range(7))
    jar = np.array([dd.qacc_warmstart[i] - dd.efc_aref[i] for i in range(7)])  # qacc of the constraint solve
    D = np.array(dd.efc_D[:7])
    f = np.array(dd.efc_force[:7])
    assert np.allclose(f, np.clip(-D * jar, -1.0, 1.0), atol=1e-9), (f, jar)
    assert np.all(np.abs(np.abs(f) - 1.0) < 1e-12)


def test_compiler_rejects_unknown_filetype(tmp_path):
    with pytest.raises(MjcfError):
        compile_mjcf(str(tmp_path / "scene.mjb"))


def test_host_pose_matches_oracle_pose():
    rng = np.random.default_rng(0)
    for _ in range(200):
        t, rpy = rng.uniform(-1, 1, 3), rng.uniform(-np.pi, np.pi, 3)
        hp, op = common.Pose(translation=t, rpy_vector=rpy), O.Pose(translation=t, rpy_vector=rpy)
        assert np.allclose(hp.rotation_q(), op.rotation_q(), atol=1e-15) and np.allclose(hp.pose_matrix(), op.pose_matrix(), atol=1e-15)
        assert np.allclose(hp.xyzrpy(), op.xyzrpy(), atol=1e-12)
        t2, q2 = rng.uniform(-1, 1, 3), rng.normal(size=4)
        h2, o2 = common.Pose(translation=t2, quaternion=q2), O.Pose(translation=t2, quaternion=q2)
        assert np.allclose((hp * h2).as_vec7(), np.concatenate([(op * o2).translation(), (op * o2).rotation_q()]), atol=1e-14)
        assert np.allclose(hp.inverse().as_vec7(), np.concatenate([op.inverse().translation(), op.inverse().rotation_q()]), atol=1e-14)
        assert abs(hp.total_angle() - op.total_angle()) < 1e-13
        for lim in (0.05, 0.5):
            a, b = hp.limit_rotation_angle(lim).limit_translation_length(lim), op.limit_rotation_angle(lim).limit_translation_length(lim)
            assert np.allclose(a.as_vec7(), np.concatenate([b.translation(), b.rotation_q()]), atol=1e-13)
        m = hp.pose_matrix()
        assert np.allclose(common.Pose(pose_matrix=m).as_vec7(), np.concatenate([O.Pose(pose_matrix=m).translation(), O.Pose(pose_matrix=m).rotation_q()]), atol=1e-13)
    tcp_h, tcp_o = common.Pose(pose_matrix=common.FrankaHandTCPOffset()), O.franka_hand_tcp_offset()
    assert np.allclose(tcp_h.as_vec7(), np.concatenate([tcp_o.translation(), tcp_o.rotation_q()]), atol=1e-15)


def test_robots_meta_config_values():
    fr3 = common.robots_meta_config(common.RobotType.FR3)
    assert fr3.dof == 7 and np.allclose(fr3.q_home, [0, -np.pi / 4, 0, -3 * np.pi / 4, 0, np.pi / 2, np.pi / 4])
    assert np.allclose(fr3.joint_limits[0], [-2.3093, -1.5133, -2.4937, -2.7478, -2.4800, 0.8521, -2.6895])
    assert common.robots_meta_config(common.RobotType.UR5e).dof == 6 and common.robots_meta_config(common.RobotType.SO101).dof == 5


def test_authored_ur5e_and_so101_scenes():
    """The two builder-authored stand-ins for robots the reference names but ships no model of: tree sizes, the archetype the
    host derives (6 hinges / 5 hinges + 2 coupled slides), ranges = robots_meta_config's, and the SO101 entry read in radians."""
    import parity_util as pu
    from rcs_env_oracle import SO101, UR5E

    ur = compile_mjcf(pu.UR5E_SCENE)
    assert (ur.nq, ur.nv, ur.nu, ur.neq, ur.ntendon) == (6, 6, 6, 0, 0)
    meta = common.sim_robots_meta_config(common.RobotType.UR5e)
    assert meta is common.robots_meta_config(common.RobotType.UR5e)
    rng = np.array([ur.arrays["jnt_range"][ur.name2id("jnt", j)] for j in UR5E["joints"]])
    assert np.allclose(rng.T, meta.joint_limits, atol=1e-12)
    # DH data sheet: shoulder height 0.163, upper arm 0.425, forearm 0.392, wrist offsets 0.127 / 0.1 / 0.1
    so = compile_mjcf(pu.SO101_SCENE)
    assert (so.nq, so.nv, so.nu, so.neq, so.ntendon) == (7, 7, 6, 1, 1)
    sm = common.sim_robots_meta_config(common.RobotType.SO101)
    rng = np.array([so.arrays["jnt_range"][so.name2id("jnt", j)] for j in SO101["joints"]])
    assert np.allclose(rng.T, sm.joint_limits, atol=1e-12) and np.allclose(sm.joint_limits, common.SO101_SIM_JOINT_RANGES)
    assert np.allclose(sm.q_home, SO101["q_home"], atol=1e-15) and np.all(sm.q_home > sm.joint_limits[0]) and np.all(sm.q_home < sm.joint_limits[1])
    # -99.66 / +99.91 of the normalised range: 0.17 % / 0.04 % of the span inside the limit
    assert 0 < sm.q_home[1] - sm.joint_limits[0][1] < 0.01 and 0 < sm.joint_limits[1][2] - sm.q_home[2] < 0.01
    # the oracle steps both (arm held against gravity by its servos, gripper follows its command)
    for robot in ("ur5e", "so101"):
        oe = pu.make_oracle_envs(1, True, robot=robot)[0]
        obs, _ = oe.reset()
        for _ in range(10):
            a = {"joints": np.zeros(pu.robot_dof(robot))}
            if oe.has_gripper:
                a["gripper"] = 1.0
            obs, _, _, trunc, info = oe.step(a)
        assert np.abs(obs["joints"] - oe.robot["q_home"]).max() < 0.02 and not trunc, (robot, obs["joints"])
        if oe.has_gripper:
            assert info["gripper_width"] > 0.5


def test_library_loads_and_exports_every_declared_symbol():
    """Every function include/rcs_hip.h declares is exported by librcs_hip.so (built by __graft_entry__.build())."""
    header = open(os.path.join(ROOT, "include", "rcs_hip.h")).read()
    declared = set(re.findall(r"\b(rcsh_[a-z0-9_]+)\s*\(", header))
    declared -= {"rcsh_model_desc", "rcsh_robot_desc", "rcsh_gripper_desc", "rcsh_env_desc", "rcsh_sim"}
    L = _lib.load()
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, missing
    assert set(_lib.EXPORTS) == declared
    assert L.rcsh_abi_version() == 2


def test_no_cpu_fallback_without_gpu():
    """Without a HIP device Sim creation fails loudly (RuntimeError), it never steps on the CPU."""
    L = _lib.load()
    if L.rcsh_device_count() > 0:
        pytest.skip("a GPU is visible")
    from rcs_amd import sim

    with pytest.raises(RuntimeError, match="no HIP device|hip"):
        sim.Sim(SCENE, n_envs=4)


def test_config_mirrors_reference_defaults():
    from rcs_amd import sim
    from rcs_amd.envs import default_sim_gripper_cfg, default_sim_robot_cfg

    c = sim.SimConfig()
    assert (c.async_control, c.realtime, c.frequency, c.max_convergence_steps) == (False, False, 30, 500)
    r = default_sim_robot_cfg()
    assert r.joints == [f"fr3_joint{i}_0" for i in range(1, 8)] and r.attachment_site == "attachment_site_0" and r.base == "base_0"
    assert abs(r.joint_rotational_tolerance - 0.05 * np.pi / 180) < 1e-18 and r.seconds_between_callbacks == 0.1
    g = default_sim_gripper_cfg()
    assert g.joint == "finger_joint1_0" and g.actuator == "actuator8_0" and g.seconds_between_callbacks == 0.05
    assert (g.max_actuator_width, g.min_actuator_width, g.max_joint_width, g.min_joint_width) == (255, 0, 0.04, 0.0)


# ---------------------------------------------------------------- fr3_simple_pick_up: the free box and the task layer
PICKUP_SCENE = os.path.join(os.path.dirname(SCENE), "..", "fr3_simple_pick_up", "scene.xml")
REF_PICKUP_SCENE = "/root/reference/assets/scenes/fr3_simple_pick_up/scene.xml"


def test_pick_up_scene_free_box_constants():
    """The cube of assets/scenes/fr3_simple_pick_up/scene.xml:30-33: half extents, density 50, friction mixed with the
    floor (element-wise max), spawn pose; the robot tables are the empty-world ones."""
    cm, empty = compile_mjcf(PICKUP_SCENE), compile_mjcf(SCENE)
    for k, a in empty.arrays.items():
        assert np.array_equal(a, cm.arrays[k]), k
    assert (cm.nq, cm.nv, cm.njnt) == (9, 9, 9) and len(cm.free_bodies) == 1 and empty.free_bodies == []
    fb = cm.free_bodies[0]
    assert (fb["name"], fb["joint_name"], fb["geom_name"]) == ("box_geom", "box_joint", "box_geom")
    assert np.allclose(fb["size"], [0.032, 0.016, 0.0288]) and np.allclose(fb["qpos0"], [0.44, 0.1, 0.03, 0, 0, 0, 1])
    vol = 8 * 0.032 * 0.016 * 0.0288
    assert np.isclose(fb["mass"], 50 * vol, rtol=1e-14)
    m = fb["mass"]
    assert np.allclose(fb["inertia"], [m / 3 * (0.016**2 + 0.0288**2), m / 3 * (0.032**2 + 0.0288**2), m / 3 * (0.032**2 + 0.016**2)], rtol=1e-14)
    assert np.allclose(fb["friction"], [1, 0.3, 0.1]) and np.allclose(fb["solref"], [0.02, 1]) and fb["plane_z"] == 0.0
    d = _lib.make_free_box_desc(cm)
    assert d.cone_elliptic == 1 and d.noslip_iterations == 5 and d.impratio == 20.0 and _lib.make_free_box_desc(empty) is None


@pytest.mark.skipif(not os.path.exists(REF_PICKUP_SCENE), reason="reference checkout not present (GPU box)")
def test_own_pick_up_scene_equals_reference_scene():
    own, ref = compile_mjcf(PICKUP_SCENE), compile_mjcf(REF_PICKUP_SCENE)
    (a,), (b,) = own.free_bodies, ref.free_bodies
    assert a.keys() == b.keys()
    for k in a:
        assert np.array_equal(a[k], b[k]) if isinstance(a[k], np.ndarray) else a[k] == b[k], k
    assert own.jnt_names == ref.jnt_names and own.body_names == ref.body_names
    for key in ("body_pos", "body_quat", "jnt_range", "actuator_gainprm", "actuator_biasprm", "qpos0"):
        assert np.allclose(own.arrays[key], ref.arrays[key]), key


def test_compiler_rejects_unsupported_free_bodies(tmp_path):
    bad = tmp_path / "s.xml"
    bad.write_text('<mujoco><worldbody><geom type="plane" size="0 0 1"/><body><freejoint/><geom type="sphere" size="0.1"/></body></worldbody></mujoco>')
    with pytest.raises(MjcfError, match="exactly one box geom"):
        compile_mjcf(str(bad))
    bad.write_text('<mujoco><worldbody><body><freejoint/><geom type="box" size="0.1 0.1 0.1"/></body></worldbody></mujoco>')
    with pytest.raises(MjcfError, match="floor plane"):
        compile_mjcf(str(bad))


def _box_oracle():
    cm = compile_mjcf(PICKUP_SCENE)
    m = O.make_model(cm)
    L = O.lib()
    L.orc_box_step1.argtypes = [C.c_void_p, C.c_void_p, C.c_double]
    L.orc_box_step2.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double]
    d = O.OrcBoxData()
    L.orc_box_reset(C.byref(m.box), C.byref(d))
    g = (C.c_double * 3)(0, 0, -9.81)

    def step(n=1):
        for _ in range(n):
            L.orc_box_step1(C.byref(m.box), C.byref(d), 0.002)
            L.orc_box_step2(C.byref(m.box), C.byref(d), g, 0.002, 0.0)

    return m, d, step


def test_oracle_box_rests_where_the_soft_contact_balances_gravity():
    """Four sticking corner contacts carry m g: the penetration solves m g / 4 = D0 K imp(r) r with D0 = imp / ((1 - imp) / m),
    i.e. r = g (1 - imp) / (4 K imp^2) -- independent of the mass.  No drift, no rotation, warm-started Newton converges at once."""
    m, d, step = _box_oracle()
    assert abs(m.box.meaninertia - 0.3195339856057694) < 1e-12 and m.box.nv_total == 15
    step(600)
    assert d.ncon == 4 and list(d.zone) == [2, 2, 2, 2] and d.newton_iter <= 1
    r = 0.0288 - d.qpos[2]
    K = 1 / (0.95**2 * 0.02**2)
    x = r / 0.001
    imp = 0.9 + 0.05 * (x * x / 0.5 if x <= 0.5 else 1 - (1 - x) ** 2 / 0.5)
    assert abs(r - 9.81 * (1 - imp) / (4 * K * imp * imp)) < 1e-9, r
    assert np.abs(np.array(d.qvel[:])).max() < 1e-9
    assert np.allclose(d.qpos[:2], [0.44, 0.1], atol=1e-12) and np.allclose(np.abs(d.qpos[3:]), [0, 0, 0, 1], atol=1e-12)
    assert abs(sum(d.force[3 * c] for c in range(4)) - m.box.mass * 9.81) < 1e-9   # normal forces carry the weight
    assert max(abs(d.force[3 * c + k]) for c in range(4) for k in (1, 2)) < 1e-9


def test_oracle_box_slides_to_a_stop_under_coulomb_friction():
    """A resting box kicked sideways at 0.5 m/s decelerates at ~mu g (mu = 1, sliding zone of the cone) and sticks;
    the stopping distance is v^2 / (2 mu g) up to the soft-contact transients; it never gains energy."""
    m, d, step = _box_oracle()
    step(600)
    x0 = d.qpos[0]
    d.qvel[0] = 0.5
    zones, speeds = set(), []
    for _ in range(200):
        step(1)
        zones.update(d.zone[: d.ncon])
        speeds.append(float(np.linalg.norm(d.qvel[:3])))
    assert 1 in zones and list(d.zone) == [2, 2, 2, 2]          # slid, then stuck
    assert max(speeds) <= 0.5 + 1e-12 and speeds[-1] < 1e-4
    assert abs((d.qpos[0] - x0) - 0.25 / (2 * 9.81)) < 1.5e-3, d.qpos[0] - x0


def test_oracle_box_pops_out_of_the_floor_after_random_cube_pos():
    """RandomCubePos writes z = 0.0288 / 2 -- the cube's centre 14.4 mm too low (python/rcs/envs/sim.py:377): the
    contact pushes it out; it ends at its rest height, upright, at the same x / y."""
    m, d, step = _box_oracle()
    d.qpos[:] = [0.5, -0.03, 0.0144, 0.3, 0, 0, 1]
    step(1)
    assert abs(np.linalg.norm(d.qpos[3:]) - 1) < 1e-15      # mj_kinematics normalised the quaternion in qpos
    yaw0 = 2 * np.arctan2(d.qpos[6], d.qpos[3])
    step(1500)
    assert abs(d.qpos[2] - 0.028692) < 2e-6 and np.abs(np.array(d.qvel[:])).max() < 1e-6
    assert abs(d.qpos[0] - 0.5) < 1e-3 and abs(d.qpos[1] + 0.03) < 1e-3
    assert abs(2 * np.arctan2(d.qpos[6], d.qpos[3]) - yaw0) < 1e-2 and abs(d.qpos[4]) < 1e-6 and abs(d.qpos[5]) < 1e-6


def test_oracle_box_free_flight_and_gyroscopic_term():
    """Above the floor: ballistic centre of mass, angular momentum conserved in the world frame (torque-free tumbling)."""
    m, d, step = _box_oracle()
    d.qpos[:] = [0, 0, 1.0, 1, 0, 0, 0]
    d.qvel[:] = [0.1, -0.2, 0.3, 3.0, -2.0, 1.0]
    I = np.array(m.box.inertia[:])

    def L_world():
        q = np.array(d.qpos[3:])
        R = common.Pose(quaternion=np.array([q[1], q[2], q[3], q[0]])).rotation_m()
        return R @ (I * np.array(d.qvel[3:]))

    L0 = L_world()
    step(100)
    assert d.ncon == 0
    assert np.allclose(d.qpos[:2], [0.1 * 0.2, -0.2 * 0.2], atol=1e-12)
    assert abs(d.qpos[2] - (1.0 + 0.3 * 0.2 - 0.5 * 9.81 * 0.2 * 0.202)) < 1e-9   # semi-implicit Euler: sum of k h^2 g
    assert np.allclose(L_world(), L0, rtol=2e-3)


# ---------------------------------------------------------------- depth renderer: scene tables and the numpy restatement
def test_render_scene_tables():
    """rcs_amd.render: one shape per drawn geom, in the frame of the link it rides on; hull planes contain their hulls."""
    from rcs_amd import render

    d = os.path.dirname(PICKUP_SCENE)
    cm = compile_mjcf(PICKUP_SCENE)
    rs = render.build_render_scene(cm, d)
    assert (rs.znear, rs.zfar) == (0.01, 50.0)  # vis.map defaults x statistic extent = 1
    names = dict(zip(rs.names, range(len(rs.names))))
    assert rs.shape[names["floor"]] == render.SHAPE_PLANE and rs.link[names["floor"]] == render.LINK_WORLD
    assert rs.shape[names["box_geom"]] == render.SHAPE_BOX and rs.link[names["box_geom"]] == render.LINK_FREE_BODY
    assert np.allclose(rs.size[names["box_geom"]], [0.032, 0.016, 0.0288])
    for i in range(1, 8):  # link i's collision hull rides on link i - 1 (joint i), link0's on the world
        assert rs.link[names[f"fr3_link{i}_collision_0"]] == i - 1
    assert rs.link[names["fr3_link0_collision_0"]] == render.LINK_WORLD and rs.link[names["hand_c_0"]] == 6
    assert rs.link[names["finger_0_left_0"]] == 7 and rs.link[names["finger_0_right_0"]] == 8
    assert np.allclose(rs.pos[names["hand_c_0"]], [0, 0, 0.107])  # fr3_link8 + hand flange folded into link 7's frame
    from rcs_amd.mjcf import find_data_file

    assert os.path.dirname(find_data_file(cm.data_dirs, "collision_vertices.npz")).endswith("fr3_empty_world")  # beside the included file
    verts = dict(np.load(find_data_file(cm.data_dirs, "collision_vertices.npz")))
    for g, name in enumerate(rs.names):
        if rs.shape[g] != render.SHAPE_HULL:
            continue
        pl = rs.planes[rs.plane_adr[g]: rs.plane_adr[g] + rs.plane_num[g]]
        v = verts[cm.geom_mesh[cm.name2id("geom", name)]]
        slack = v @ pl[:, :3].T - pl[:, 3]
        assert slack.max() < 1e-9 and np.allclose(np.linalg.norm(pl[:, :3], axis=1), 1)  # every hull vertex inside every plane
        assert np.abs(slack).min(axis=0).max() < 1e-9                                   # and every plane touches the hull
        c, r = rs.sphere[g][:3], rs.sphere[g][3]
        assert (np.linalg.norm(v - c, axis=1) <= r + 1e-12).all() and (np.abs(v - c) <= rs.size[g] + 1e-12).all()
    link, pos, rot, fovy = render.camera_in_link(cm, "bird_eye_cam")
    assert link == render.LINK_WORLD and np.allclose(pos, [0.271, 0, 2.08]) and fovy == 45.0
    assert render.camera_in_link(cm, "wrist_0")[0] == 6
    with pytest.raises(RuntimeError, match="No camera named"):
        render.camera_in_link(cm, "nope")


def test_oracle_depth_render_known_answers():
    """The numpy ray-caster on hand-made frames: a camera looking straight down at the floor and at a box."""
    import rcs_render_oracle as RO
    from rcs_amd import render

    rs = render.RenderScene(
        shape=np.array([0, 1], dtype=np.int32), link=np.array([-1, -2], dtype=np.int32), pos=np.zeros((2, 3)), rot=np.tile(np.eye(3).reshape(9), (2, 1)),
        size=np.array([[0, 0, 0], [0.1, 0.2, 0.05]]), plane_adr=np.zeros(2, dtype=np.int32), plane_num=np.zeros(2, dtype=np.int32),
        sphere=np.array([[0, 0, 0, -1.0], [0, 0, 0, 0.3]]), planes=np.zeros((1, 4)), znear=0.01, zfar=50.0)
    frames = {-1: (np.eye(3), np.zeros(3)), -2: (np.eye(3), np.array([0.0, 0.0, 0.05]))}  # box resting on the floor
    down = np.eye(3).reshape(9)  # a MuJoCo camera looks along -z of its frame: world axes = looking straight down
    W = H = 21
    dgl, mm, cR, cp = RO.render_depth(rs, (-1, np.array([0.0, 0.0, 1.0]), down, 90.0, W, H), frames)
    assert mm.dtype == np.uint16 and dgl.dtype == np.float32 and mm.shape == (H, W)
    assert mm[H // 2, W // 2] in (899, 900)              # top of the box, 0.9 m below the camera (uint16 truncates the float32 metres)
    assert mm[0, 0] in (999, 1000) and mm[H - 1, W - 1] in (999, 1000)  # view depth of the floor is 1 m on every ray that misses the box
    box = (mm < 950)
    assert box.sum() == 3 * 5 or box.sum() == 3 * 4 + 0 or 9 <= box.sum() <= 20  # 0.2 x 0.4 m footprint at 0.9 m under a 90 degree view
    assert np.array_equal(box, box[::-1]) and np.array_equal(box, box[:, ::-1])
    # OpenGL encoding of the raw buffer: d = (1/near - 1/z) / (1/near - 1/far), rows bottom-up
    assert abs(float(dgl[0, 0]) - (100 - 1 / 1.0) / (100 - 0.02)) < 1e-6
    # a camera above the far plane's reach sees background: depth buffer 1, ~far in millimetres
    dgl2, mm2, _, _ = RO.render_depth(rs, (-1, np.array([0.0, 0.0, 60.0]), down, 10.0, 3, 3), frames)
    assert (dgl2 == 1.0).all() and (mm2 > 49900).all()


def test_random_object_pos_draws_follow_the_reference():
    """RandomObjectPos.reset (python/rcs/envs/sim.py:331-354): draw order x, y, (w); z and the rest of the quaternion kept;
    mjData's free-joint layout is [x y z qw qx qy qz] while Pose.rotation_q() is xyzw."""
    from rcs_amd.envs.creators import random_object_qpos

    pose = common.Pose(translation=np.array([0.5, 0.1, 0.03]), quaternion=np.array([0.0, 0.0, np.sin(0.3), np.cos(0.3)]))
    np.random.seed(7)
    got = random_object_qpos(pose, 3, include_position=True, include_rotation=True)
    np.random.seed(7)
    for e in range(3):
        x = 0.5 + np.random.random() * 0.2 - 0.1
        y = 0.1 + np.random.random() * 0.2 - 0.1
        w = 2 * np.random.random() - np.cos(0.3)
        assert np.allclose(got[e], [x, y, 0.03, w, 0, 0, np.sin(0.3)], atol=1e-15)
    np.random.seed(7)
    state = np.random.get_state()[1].copy()
    fixed = random_object_qpos(pose, 2, include_position=False, include_rotation=False)
    assert np.array_equal(np.random.get_state()[1], state)  # nothing drawn
    assert np.allclose(fixed, np.tile([0.5, 0.1, 0.03, np.cos(0.3), 0, 0, np.sin(0.3)], (2, 1)))


def test_default_free_camera_pose():
    """mjv_defaultFreeCamera as rcs_amd.render restates it: looks at stat.center from 1.5 x extent away, along the
    scene's vis.global azimuth / elevation (fr3 scenes: 120 / -20 degrees), camera -z forward and +y up."""
    from rcs_amd import render

    cm = compile_mjcf(SCENE)
    assert np.allclose(cm.stat_center, [0.3, 0, 0.4]) and (cm.vis_azimuth, cm.vis_elevation, cm.vis_fovy) == (120.0, -20.0, 45.0)
    link, pos, rot, fovy = render.default_free_camera(cm)
    R = rot.reshape(3, 3)
    assert link == render.LINK_WORLD and fovy == 45.0
    assert np.allclose(R.T @ R, np.eye(3), atol=1e-12) and np.isclose(np.linalg.det(R), 1.0)
    forward = -R[:, 2]
    assert np.allclose(pos + 1.5 * forward, [0.3, 0, 0.4])            # the optical axis passes through the centre, 1.5 m away
    assert np.isclose(np.degrees(np.arcsin(forward[2])), -20.0) and np.isclose(np.degrees(np.arctan2(forward[1], forward[0])), 120.0)
    assert R[2, 1] > 0 and abs(R[2, 0]) < 1e-12                       # +y of the camera points up, +x is horizontal


def test_pybind_module_matches_the_reference_api():
    """The compiled binding `rcs_hip._core.sim` (extensions/rcs_hip) against the NAMES of the reference's `rcs._core.sim`
    (tests/golden/core_sim_api.json, generated from the reference's stub files by tools/make_core_api_fixture.py): every
    class, every method with its argument names, every field -- including what SimRobot / SimGripper inherit from
    common.Robot / common.Gripper.  No GPU here: constructing a Sim must fail with the reference's exception type."""
    import json
    import re
    import sys

    sys.path.insert(0, os.path.join(ROOT, "extensions", "rcs_hip"))
    import rcs_hip
    from rcs_hip import _core

    api = json.load(open(os.path.join(ROOT, "tests", "golden", "core_sim_api.json")))
    assert _core.abi_version() == 2
    own = {"SimRobotConfig": {"mjcf_scene_path"}, "Sim": set()}
    for cls, spec in api.items():
        if cls in ("Robot", "Gripper", "BaseCameraConfig"):
            continue
        k = getattr(_core.sim, cls)
        methods = dict(spec["methods"])
        for base in spec["bases"]:
            if base in api:  # inherited from common.Robot / common.Gripper
                methods = {**api[base]["methods"], **methods}
        for name, args in methods.items():
            assert hasattr(k, name), (cls, name)
            if cls == "CameraType" and name.startswith("__"):
                continue  # pybind11's enum machinery (its docstrings name no arguments)
            if name == "__init__" and cls == "Sim":
                continue  # raw mjModel* / mjData* cannot cross without MuJoCo: Sim(model, n_envs, device, free_box)
            sig = getattr(k, name).__doc__.split("\n")[0]
            have = re.findall(r"(\w+): ", sig)
            assert [a for a in args if a not in have] == [], (cls, name, args, sig)
        fields = list(spec["fields"])
        for base in spec["bases"]:
            if base in api:
                fields += api[base]["fields"]
        for f in fields:
            assert hasattr(k, f), (cls, f)
    assert [int(getattr(_core.sim.CameraType, m)) for m in ("free", "tracking", "fixed", "default_free")] == [0, 1, 2, 3]  # camera.h:19-24
    cc = _core.sim.SimCameraConfig("wrist_0", 30, 64, 48)
    assert (cc.identifier, cc.frame_rate, cc.resolution_width, cc.resolution_height, cc.type) == ("wrist_0", 30, 64, 48, _core.sim.CameraType.fixed)
    from rcs_amd.mjcf import compile_mjcf

    cm = compile_mjcf(os.path.join(ROOT, "robot-control-stack_amd", "rcs_amd", "scenes", "fr3_empty_world", "scene.xml"))
    with pytest.raises(RuntimeError):
        _core.sim.Sim(rcs_hip.model_tables(cm), 4)
    c = _core.sim.SimRobotConfig()
    c.add_id("0")
    assert c.joints[0] == "fr3_joint1_0" and c.attachment_site == "attachment_site_0" and c.base == "base_0"
    g = _core.sim.SimGripperConfig()
    g.add_id("0")
    assert g.joint == "finger_joint1_0" and g.collision_geoms[0] == "hand_c_0" and g.max_actuator_width == 255


def test_pybind_common_module_matches_the_reference_api():
    """`rcs_hip._core.common` against the NAMES of the reference's `rcs._core.common` (tests/golden/core_common_api.json, from
    the reference's common.pyi by tools/make_core_api_fixture.py): Pose, RPY, Kinematics, Pin, RobotType, RobotPlatform,
    RobotMetaConfig, RobotConfig, BaseCameraConfig, GraspType -- every method, every overload's argument names, every field --
    the module functions and the exported enum constants, with the reference's enum values (Robot.h:22-23)."""
    import json
    import re
    import sys

    sys.path.insert(0, os.path.join(ROOT, "extensions", "rcs_hip"))
    from rcs_hip import _core

    c = _core.common
    api = json.load(open(os.path.join(ROOT, "tests", "golden", "core_common_api.json")))
    for cls, spec in api["classes"].items():
        k = getattr(c, cls)
        for name, args in spec["methods"].items():
            assert hasattr(k, name), (cls, name)
            if cls in ("RobotType", "RobotPlatform", "GraspType") and name.startswith("__"):
                continue  # pybind11's enum machinery
            doc = getattr(k, name).__doc__ or ""
            for overload in spec["overloads"].get(name, [args]):
                # some line of the docstring (one per overload) names all of this overload's arguments
                assert any(all(re.search(rf"\b{a}: ", line) for a in overload) for line in doc.split("\n")), (cls, name, overload, doc)
        for f in spec["fields"]:
            assert hasattr(k, f), (cls, f)
        for base in spec["bases"]:
            assert issubclass(k, getattr(c, base)), (cls, base)
    for fn, args in api["functions"].items():
        sig = getattr(c, fn).__doc__.split("\n")[0]
        assert [a for a in args if a not in re.findall(r"(\w+): ", sig)] == [], (fn, sig)
    for name in api["constants"]:
        assert hasattr(c, name), name
    assert [int(c.RobotType.FR3), int(c.RobotType.UR5e), int(c.RobotType.SO101), int(c.RobotType.XArm7)] == [0, 1, 2, 3]
    assert [int(c.RobotPlatform.SIMULATION), int(c.RobotPlatform.HARDWARE)] == [0, 1]
    # values: the compiled tables equal the host mirror's (include/rcs/Robot.h:24-95), the helpers the reference's constants
    from rcs_amd import common as H

    for t in ("FR3", "UR5e", "SO101", "XArm7"):
        a, b = c.robots_meta_config(getattr(c.RobotType, t)), H.robots_meta_config(getattr(H.RobotType, t))
        assert a.dof == b.dof and np.array_equal(a.q_home, b.q_home) and np.array_equal(a.joint_limits, b.joint_limits), t
    assert np.array_equal(c.FrankaHandTCPOffset(), H.FrankaHandTCPOffset()) and np.array_equal(c.IdentityRotQuatVec(), [0, 0, 0, 1])
    cfg = c.RobotConfig()
    assert (cfg.robot_type, cfg.robot_platform, cfg.attachment_site) == (c.RobotType.FR3, c.RobotPlatform.SIMULATION, "attachment_site")
    assert cfg.tcp_offset.is_close(c.Pose())
    # the RL IK class of the reference's rcs_robotics_library extension (rl.pyi): constructible by name, a Kinematics
    for cls, spec in api["rl"].items():
        k = getattr(_core.rl, cls)
        assert issubclass(k, c.Kinematics), cls
        for name, args in spec["methods"].items():
            doc = getattr(k, name).__doc__ or ""
            assert all(re.search(rf"\b{a}: ", doc) for a in args), (cls, name, args, doc)
    with pytest.raises((RuntimeError, OSError)):
        c.Pin("no_such_robot.urdf")  # (the reference's default is urdf=True: the file is read as a URDF)


def test_urdf_chain_equals_the_mjcf_chain_on_the_oracle():
    """`Pin(path, frame_id="fr3_link8", urdf=True)` is the reference's default constructor (src/pybind/rcs.cpp:296-300,
    src/rcs/Kinematics.cpp:12-19) and the reference ships assets/fr3/urdf/fr3.urdf.  The URDF reader (rcs_amd.urdf: the chain
    rewritten as MJCF, fixed joints folded into their moving ancestor, a frame per link) must give the same kinematics as the
    MJCF model of the same robot: forward map of `fr3_link8` == the MJCF's attachment site at 32 configurations (round-off), and
    the CLIK's iterates on the two models -- the oracle's restatement of Pin::inverse -- agree to the last iteration."""
    import rcs_oracle as O
    from parity_util import SCENE
    from rcs_amd.mjcf import compile_mjcf
    from rcs_amd.urdf import compile_urdf, urdf_to_mjcf
    from rcs_env_oracle import FR3_Q_HOME

    urdf = os.path.join(os.path.dirname(SCENE), "fr3.urdf")
    text, info = urdf_to_mjcf(urdf)
    assert info["joints"] == [f"fr3_joint{i}" for i in range(1, 8)] and info["root"] == "fr3_link0" and info["leaves"] == ["fr3_link8"]
    assert text.count("<body ") == 8 and 'site name="fr3_link8" pos="0.0 0.0 0.107"' in text  # link8 rides on link7's body
    cu, _ = compile_urdf(urdf)
    joints = info["joints"]
    ou = O.Sim(cu, joints, ["act_" + j for j in joints], "fr3_link8", "fr3_link0", FR3_Q_HOME, None, arm_collision_geoms=[])
    arm = [f"fr3_joint{i}_0" for i in range(1, 8)]
    om = O.Sim(compile_mjcf(SCENE), arm, arm, "attachment_site_0", "base_0", FR3_Q_HOME, None, "finger_joint1_0", "actuator8_0")
    rng = np.random.default_rng(0)
    for _ in range(32):
        q = np.asarray(FR3_Q_HOME) + rng.uniform(-0.6, 0.6, 7)
        a, b = ou.ik_forward(q), om.ik_forward(q)
        assert np.abs(a.translation() - b.translation()).max() < 1e-14 and np.abs(a.rotation_q() - b.rotation_q()).max() < 1e-14
    for k in range(8):
        target = om.ik_forward(np.asarray(FR3_Q_HOME) + rng.uniform(-0.3, 0.3, 7))
        (qa, ia), (qb, ib) = ou.ik_inverse(target, FR3_Q_HOME), om.ik_inverse(target, FR3_Q_HOME)
        assert ia == ib and qa is not None and np.abs(qa[:7] - qb[:7]).max() < 1e-12, (k, ia, ib)
    # the joint limits the URDF states are the robot's (robots_meta_config, Robot.h:28-43)
    from rcs_amd import common as H

    lim = H.robots_meta_config(H.RobotType.FR3).joint_limits
    assert np.allclose(np.asarray(cu.arrays["jnt_range"])[:, 0], lim[0]) and np.allclose(np.asarray(cu.arrays["jnt_range"])[:, 1], lim[1])


def test_compiled_pose_equals_the_host_mirror():
    """Every Pose / RPY operation of the compiled classes (csrc/pose.h on the host) against the Python mirror that the
    reference's own known answers pin (tests/test_oracle_pins.py), on random inputs incl. improper matrices: round-off."""
    import pickle
    import sys

    sys.path.insert(0, os.path.join(ROOT, "extensions", "rcs_hip"))
    from rcs_hip import _core
    from rcs_amd import common as H

    c = _core.common
    rng = np.random.default_rng(0)
    worst = 0.0
    for _ in range(300):
        q, t, m = rng.normal(size=4), rng.normal(size=3), rng.normal(size=(4, 4))
        m[3] = [0, 0, 0, 1]
        a, b = c.Pose(quaternion=q, translation=t), H.Pose(quaternion=q, translation=t)
        a2, b2 = c.Pose(pose_matrix=m), H.Pose(pose_matrix=m)
        worst = max(worst, np.abs(a.xyzrpy() - b.xyzrpy()).max(), np.abs((a * a2).pose_matrix() - (b * b2).pose_matrix()).max(),
                    np.abs(a2.rotation_q() - b2.rotation_q()).max(), abs(a.total_angle() - b.total_angle()),
                    np.abs(a.inverse().pose_matrix() - b.inverse().pose_matrix()).max(),
                    np.abs(a.interpolate(a2, 0.3).pose_matrix() - b.interpolate(b2, 0.3).pose_matrix()).max(),
                    np.abs(a.limit_rotation_angle(0.2).rotation_q() - b.limit_rotation_angle(0.2).rotation_q()).max(),
                    np.abs(a.limit_translation_length(0.1).translation() - b.limit_translation_length(0.1).translation()).max(),
                    np.abs(c.Pose(rpy_vector=t, translation=t).rotation_q() - H.Pose(rpy_vector=t, translation=t).rotation_q()).max(),
                    np.abs(c.Pose(rpy=c.RPY(rpy=t)).rotation_q() - H.Pose(rpy=H.RPY(rpy=t)).rotation_q()).max(),
                    np.abs(c.Pose(rotation=a.rotation_m()).rotation_q() - H.Pose(rotation=b.rotation_m()).rotation_q()).max(),
                    np.abs(c.RPY(rpy=t).rotation_matrix() - H.RPY(rpy=t).rotation_matrix()).max(),
                    np.abs(c.RPY(rpy=t).as_quaternion_vector() - H.RPY(rpy=t).as_quaternion_vector()).max(),
                    np.abs(a.rotation_rpy().as_vector() - b.rotation_rpy().as_vector()).max())
        assert a.is_close(a2, 0.3, 0.3) == b.is_close(b2, 0.3, 0.3)
    assert worst < 1e-12, worst
    p = c.Pose(quaternion=rng.normal(size=4), translation=rng.normal(size=3))
    assert pickle.loads(pickle.dumps(p)).is_close(p, 1e-12, 1e-12) and c.Pose(pose=p).is_close(p)
    r = pickle.loads(pickle.dumps(c.RPY(0.1, 0.2, 0.3)))
    assert (r.roll, r.pitch, r.yaw) == (0.1, 0.2, 0.3) and (r + r).is_close(c.RPY(0.2, 0.4, 0.6)) and "roll" in str(p)


def test_hull_edges_and_the_outline_method_against_the_plane_walk():
    """The ray caster's outline method (csrc/render.h: k_hull_views) rests on the polytope edges the library works out from a
    hull's face planes (csrc/model.cpp: build_hull_edges, exported host-only as rcsh_hull_edges).  For every drawn hull of the
    scenes: Euler's formula holds on what it finds, every edge lies on both its planes, and -- restated in numpy -- "inside the
    cone over the outline, entry depth from the front planes" sees exactly what walking all planes for both ends of the ray's
    interval sees, for thousands of rays from random eye points."""
    import ctypes as C

    from rcs_amd import _lib

    L = _lib.load()
    rng = np.random.default_rng(0)
    scenes = os.path.join(ROOT, "robot-control-stack_amd", "rcs_amd", "scenes")
    n_hulls = rays = 0
    for scene in ("fr3_empty_world", "xarm7_pick_world"):
        hulls = np.load(os.path.join(scenes, scene, "render_hulls.npz"))
        for name in hulls.files:
            pl = np.ascontiguousarray(hulls[name], dtype=np.float64)
            ne, centre = C.c_int32(0), np.zeros(3)
            _lib.check(L.rcsh_hull_edges(pl.ctypes.data_as(_lib._F64P), len(pl), 0, None, None, C.byref(ne), centre.ctypes.data_as(_lib._F64P)))
            assert ne.value > len(pl), (scene, name, ne.value)  # a polytope with f faces has at least 3 f / 2 edges
            ep, ev = np.zeros((ne.value, 2), dtype=np.int32), np.zeros((ne.value, 6))
            _lib.check(L.rcsh_hull_edges(pl.ctypes.data_as(_lib._F64P), len(pl), ne.value, ep.ctypes.data_as(_lib._I32P), ev.ctypes.data_as(_lib._F64P),
                                         C.byref(ne), centre.ctypes.data_as(_lib._F64P)))
            for side in (0, 1):  # both end points on both planes
                v = ev[:, :3], ev[:, 3:]
                for w in v:
                    assert np.abs((pl[ep[:, side], :3] * w).sum(axis=1) - pl[ep[:, side], 3]).max() < 1e-7
            assert (pl[:, :3] @ centre - pl[:, 3]).max() < -1e-4  # strictly inside
            # Euler: vertices = distinct end points
            verts = np.unique(np.round(np.concatenate([ev[:, :3], ev[:, 3:]]) / 1e-6).astype(np.int64), axis=0)
            faces = len(np.unique(ep))
            assert len(verts) - ne.value + faces == 2, (scene, name, len(verts), ne.value, faces)
            r = np.linalg.norm(ev[:, :3] - centre, axis=1).max()
            for _ in range(6):
                u = rng.normal(size=3); u /= np.linalg.norm(u)
                o = centre + u * rng.uniform(1.3 * r, 2.0)
                D = centre + rng.normal(size=(1500, 3)) * 0.8 * r - o
                D /= np.abs(D @ u)[:, None]
                nd, no = D @ pl[:, :3].T, pl[:, 3] - pl[:, :3] @ o
                with np.errstate(divide="ignore", invalid="ignore"):
                    t = no / nd
                t0 = np.maximum(np.where(nd < 0, t, -np.inf).max(axis=1), 0.01)
                t1 = np.minimum(np.where(nd > 0, t, np.inf).min(axis=1), 50.0)
                hit_walk = (t0 <= t1) & (t0 > 0.01)
                front = no < 0
                sil = front[ep[:, 0]] != front[ep[:, 1]]
                m = np.cross(ev[sil, :3] - o, ev[sil, 3:] - o)
                m[m @ (centre - o) < 0] *= -1
                inside = (D @ m.T >= 0).all(axis=1)
                with np.errstate(divide="ignore", invalid="ignore"):
                    tf = np.where(nd[:, front] < 0, no[front] / nd[:, front], -np.inf)
                t0o = np.maximum(tf.max(axis=1), 0.01)
                hit_out = inside & (t0o > 0.01) & (t0o < 50.0)
                assert 8 <= sil.sum() <= 64, sil.sum()
                assert (hit_walk != hit_out).sum() <= 1 and np.abs(t0[hit_walk & hit_out] - t0o[hit_walk & hit_out]).max() < 1e-12
                assert hit_walk.sum() > 100
                rays += len(D)
            n_hulls += 1
    assert n_hulls == 19 and rays > 100000


def test_stepping_kernels_leave_room_for_four_workgroups_per_cu():
    """A CU of the MI355X has 160 KB of LDS and four SIMDs; the stepping kernels run one wavefront per SIMD, so a batch of 4096
    environments is exactly one round of 1024 workgroups -- IF four of them fit a CU's LDS.  Round 5 found the lean detection kernel of
    the 9-joint robots at 41,144 bytes (184 too many): 768 + 256 workgroups in two rounds, step_until_convergence at half speed.  Every
    k_run_team instantiation of the built library stays at or under 160 KB / 4 -- except the contact-resolving kernel of per-environment
    escalation (no free box, contacts resolved: <T, *, false, true, *>), which since round 6 keeps records for 64 contacts instead of
    48 and runs one workgroup per escalated environment, never the whole batch: three per CU are what it gets."""
    import re
    import subprocess
    import tempfile

    llvm = "/opt/rocm/lib/llvm/bin"
    if not os.path.exists(os.path.join(llvm, "clang-offload-bundler")):
        pytest.skip("no ROCm LLVM tools to read the code object with")
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "gfx950.co")
        subprocess.check_call([f"{llvm}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", _lib.LIB_PATH, fat])
        subprocess.check_call([f"{llvm}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", f"--output={co}",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"])
        notes = subprocess.check_output([f"{llvm}/llvm-readelf", "--notes", co], text=True)
    sizes = {}
    for m in re.finditer(r"- \.agpr_count:\s+(\d+)(.*?)\.wavefront_size", notes, re.S):
        f = dict(re.findall(r"\.(\w+):\s+(\S+)", m.group(0)))
        if "k_run_team" in f.get("name", ""):
            sizes[f["name"]] = int(f["group_segment_fixed_size"])
    assert len(sizes) >= 20, len(sizes)
    names = subprocess.check_output(["c++filt"] + list(sizes), text=True).split("\n")
    too_big = {}
    for (k, v), nm in zip(sizes.items(), names):
        args_ = re.search(r"k_run_team(?:_occ2)?<rcsh::Topo<\d+, (?:true|false)>, (true|false), (true|false), (true|false), (true|false)>", nm)
        assert args_, nm
        boxless_contact = args_.group(2) == "false" and args_.group(3) == "true"
        if v > (160 * 1024 // 3 if boxless_contact else 160 * 1024 // 4):
            too_big[nm] = v
    assert not too_big, too_big
