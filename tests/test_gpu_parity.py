"""GPU parity: HIP path (through the C-ABI) vs the CPU oracle on the same seeded inputs.

Tolerances (north star: 1e-5 on joint positions / velocities, flags bit-exact).  Both sides run the same FP64
algorithm in different formulations and agree to round-off; measured on the headline workload (4096-class rollouts,
async 17 substeps and until-convergence): joint positions 2e-15, joint VELOCITIES 7e-15, finger slides 7e-18, observations
2e-15.  The bars below leave three orders of magnitude for longer rollouts and other GPUs' instruction scheduling:
  * joint positions / TCP observations / CLIK targets <= 1e-9 (TOL), joint velocities <= 1e-8 (VTOL) -- 3 / 4 orders of
    magnitude inside the north star's 1e-5 -- finger slides <= 1e-9 m, every flag and substep count bit-exact;
  * contact scenes (cube tumbling on the floor, the pinch): cube pose <= 1e-7, since impacts amplify round-off.
(Round 1 carried 1e-4 / 1e-6 bars: those were set by a second, one-lane-per-environment kernel whose fingers, resting exactly
on a joint limit with zero actuator force, flipped a one-sided limit row on 1e-17 round-off.  That kernel is gone.)
"""

import os

import numpy as np
import pytest

from parity_util import run_cartesian_rollout_parity, run_joint_rollout_parity

pytestmark = pytest.mark.gpu

TOL = 1e-9
VTOL = 1e-8
FINGER_TOL = 1e-9


@pytest.fixture(autouse=True, params=["team"])
def kernel(request):
    """The kernel variant every test pins (rcsh_sim_set_kernel).  One variant is left: 16 lanes per environment."""
    import parity_util

    parity_util.KERNEL = request.param
    yield request.param
    parity_util.KERNEL = "auto"


@pytest.mark.parametrize("gripper", [True, False])
def test_joints_async_17_substeps(gripper):
    rep = run_joint_rollout_parity(n_envs=96, n_steps=6, async_control=True, seed=1, gripper=gripper)
    assert rep["max_abs_qpos"] < TOL and rep["max_abs_qvel"] < VTOL and rep["max_abs_obs"] < TOL, rep
    assert rep["max_abs_finger"] < FINGER_TOL and rep["max_abs_gripper_width"] < 1e-7, rep
    assert rep["flag_mismatches"] == 0, rep


def test_physics_all_joints_round_off():
    from parity_util import run_physics_parity_mid_stroke

    rep = run_physics_parity_mid_stroke(n_envs=64, n_calls=6, k=17, seed=2)
    assert rep["max_abs_qpos"] < 1e-10 and rep["max_abs_qvel"] < 1e-8 and rep["max_abs_cart"] < 1e-10, rep
    assert rep["flag_mismatches"] == 0, rep


@pytest.mark.parametrize("n_over", [2, 6])
def test_physics_many_limit_rows(n_over):
    """2 penetrating arm limit rows: the team kernel's active-set vote; 6: its Newton + line-search path."""
    from parity_util import run_physics_parity_at_joint_limits

    rep = run_physics_parity_at_joint_limits(n_envs=32, n_calls=8, k=17, seed=4, n_over=n_over)
    assert rep["max_rows"] >= n_over, rep
    assert rep["max_abs_qpos"] < 1e-9 and rep["max_abs_qvel"] < 1e-7, rep


def test_joints_until_convergence():
    rep = run_joint_rollout_parity(n_envs=40, n_steps=3, async_control=False, seed=7, gripper=True)
    assert rep["max_abs_qpos"] < TOL and rep["max_abs_obs"] < TOL and rep["max_abs_finger"] < FINGER_TOL, rep
    assert rep["flag_mismatches"] == 0 and rep["substep_mismatches"] == 0, rep


def test_sim_config_variants():
    """SimConfig fields other than the defaults: 10 Hz async control (50 substeps per env-step, envs/sim.py:52-53) and a
    convergence cap of 77 substeps that every environment hits (sim.cpp:84-106: converged stays false)."""
    rep = run_joint_rollout_parity(n_envs=20, n_steps=3, async_control=True, seed=23, frequency=10)
    assert rep["max_abs_qpos"] < TOL and rep["max_abs_finger"] < FINGER_TOL and rep["flag_mismatches"] == 0, rep
    rep = run_joint_rollout_parity(n_envs=20, n_steps=3, async_control=False, seed=23, max_convergence_steps=77)
    assert rep["max_abs_qpos"] < TOL and rep["max_abs_finger"] < FINGER_TOL, rep
    assert rep["flag_mismatches"] == 0 and rep["substep_mismatches"] == 0, rep


def test_two_episodes_reset_quirks():
    # prev_action survives reset (Q2), gripper reset is overwritten by sim.reset (Q1)
    rep = run_joint_rollout_parity(n_envs=33, n_steps=4, async_control=True, seed=3, gripper=True, episodes=2)
    assert rep["max_abs_qpos"] < TOL and rep["max_abs_finger"] < FINGER_TOL and rep["flag_mismatches"] == 0, rep


@pytest.mark.parametrize("mode", ["xyzrpy", "tquat"])
def test_cartesian_relative_clik(mode):
    # tolerance: the CLIK stops at |err| < 1e-4 after ~70-100 damped steps; both sides run the same iteration, so the
    # solutions agree far below that (differences come from sin/cos implementations only)
    rep = run_cartesian_rollout_parity(n_envs=32, n_steps=5, async_control=True, seed=11, mode=mode)
    assert rep["max_abs_target"] < TOL and rep["max_abs_qpos"] < TOL and rep["max_abs_tquat"] < TOL, rep
    assert rep["flag_mismatches"] == 0, rep
    # the xyzrpy observation, component by component: after +-0.1 rad steps the TCP is tilted away from the roll = +-pi
    # seam of the reference's Euler extraction, where an equivalent-but-different branch would be an O(1) error
    assert rep["max_abs_xyzrpy"] < TOL and rep["rpy_componentwise"] > rep["steps"] * 32 // 2, rep


@pytest.mark.parametrize("mode", ["joints", "xyzrpy", "tquat"])
def test_relative_to_configured_origin(mode):
    """RelativeTo.CONFIGURED_ORIGIN (base.py:490-565): actions are offsets from the origin fixed at reset, and the step
    limit applies to the CHANGE of the offset (`_last_action`), not to the offset itself."""
    if mode == "joints":
        rep = run_joint_rollout_parity(n_envs=32, n_steps=6, async_control=True, seed=13, relative_to="configured_origin")
        assert rep["max_abs_obs"] < TOL and rep["max_abs_finger"] < FINGER_TOL, rep
    else:
        rep = run_cartesian_rollout_parity(n_envs=24, n_steps=6, async_control=True, seed=13, mode=mode, relative_to="configured_origin")
        assert rep["max_abs_target"] < TOL and rep["max_abs_tquat"] < TOL, rep
    assert rep["max_abs_qpos"] < TOL and rep["flag_mismatches"] == 0, rep


def test_cartesian_absolute_until_convergence():
    rep = run_cartesian_rollout_parity(n_envs=16, n_steps=2, async_control=False, seed=5, mode="xyzrpy", relative=False, gripper=False)
    assert rep["max_abs_target"] < TOL and rep["max_abs_qpos"] < TOL, rep
    assert rep["flag_mismatches"] == 0, rep


def test_ik_kernels_match_oracle_and_round_trip():
    import rcs_oracle as O
    from parity_util import make_vec_env

    venv = make_vec_env(8, True, gripper=False, relative=False)
    venv.reset()
    ik = venv.robot.get_ik()
    rng = np.random.default_rng(0)
    q0 = np.tile(venv.robot.get_joint_position()[0], (8, 1))
    qt = q0 + rng.uniform(-0.3, 0.3, size=q0.shape)
    tcp = O.franka_hand_tcp_offset()
    tcp7 = np.concatenate([tcp.translation(), tcp.rotation_q()])
    pose = ik.forward(qt, tcp7)  # Pin::forward semantics: frame * tcp^-1
    # target for inverse(): frame * tcp, so that inverse() recovers the frame (reference quirk Q7)
    from rcs_amd.common import Pose

    target = np.stack([(Pose(translation=p[:3], quaternion=p[3:]) * Pose(translation=tcp7[:3], quaternion=tcp7[3:]) *
                        Pose(translation=tcp7[:3], quaternion=tcp7[3:])).as_vec7() for p in pose])
    q, ok, iters = ik.inverse(target, q0, tcp7)
    assert ok.all() and (iters > 5).all()
    back = ik.forward(q[:, :7], tcp7)
    assert np.abs(back[:, :3] - pose[:, :3]).max() < 2e-4
    assert np.abs(q[:, 7:]).max() == 0.0
    venv.close()


@pytest.mark.parametrize("robot", ["fr3", "xarm7"])
def test_kinematics_api_matches_oracle(robot, kernel):
    """rcs.common.Kinematics.forward / inverse (`rcsh_ik_*`) against the oracle's Pin restatement on both chains:
    same iteration counts, joint solutions <= 1e-9, poses <= 1e-12."""
    import rcs_oracle as O
    from parity_util import make_oracle_envs, make_vec_env

    n = 16
    venv = make_vec_env(n, True, gripper=False, relative=False, robot=robot)
    o = make_oracle_envs(1, True, gripper=False, relative=False, robot=robot)[0]
    venv.reset()
    o.reset()
    ik = venv.robot.get_ik()
    rng = np.random.default_rng(3)
    q0 = venv.robot.get_joint_position()
    qt = q0 + rng.uniform(-0.25, 0.25, size=q0.shape)
    tcp = O.franka_hand_tcp_offset() if robot == "fr3" else O.Pose()
    tcp7 = np.concatenate([tcp.translation(), tcp.rotation_q()])
    # the base frame (xArm7's base body sits 0.12 m above its world; the FR3's at the origin)
    bp, obp = venv.robot.get_base_pose_in_world_coordinates(), o.sim.get_base_pose()
    assert np.abs(bp.translation() - obp.translation()).max() < 1e-15 and np.abs(bp.rotation_q() - obp.rotation_q()).max() < 1e-15
    if robot == "xarm7":
        assert abs(bp.translation()[2] - 0.12) < 1e-12
    fwd = ik.forward(qt, tcp7)
    for e in range(n):
        of = o.sim.ik_forward(qt[e], tcp)
        assert np.abs(fwd[e] - np.concatenate([of.translation(), of.rotation_q()])).max() < 1e-12
    q, ok, iters = ik.inverse(fwd, q0, tcp7)  # Pin.inverse(Pin.forward(q)) is NOT the identity (quirk Q7): just compare
    for e in range(n):
        oq, oit = o.sim.ik_inverse(O.Pose(translation=fwd[e][:3], quaternion=fwd[e][3:]), q0[e], tcp)
        assert bool(ok[e]) == (oq is not None) and int(iters[e]) == oit
        if oq is not None:
            assert np.abs(q[e] - oq).max() < 1e-9
    venv.close()


def test_fine_grained_api_sequence_with_masks(kernel):
    """The 1:1 Sim / SimRobot / SimGripper surface (boundary row b) driven call by call, half of the environments
    masked out of some calls, against one oracle per environment doing exactly the calls its mask lets through:
    step(k), step_until_convergence, set_joint_position, set_joints_hard, move_home, robot reset, gripper shut / open /
    set_normalized_width / reset, sim reset; states, flags and convergence step counts compared after every stage."""
    import parity_util as pu
    import rcs_oracle as O
    from rcs_amd import sim as S
    from rcs_amd.envs import default_sim_gripper_cfg, default_sim_robot_cfg
    from rcs_amd.mjcf import compile_mjcf
    from rcs_env_oracle import FR3_Q_HOME

    n = 12
    cfg = default_sim_robot_cfg("fr3_empty_world")
    simu = S.Sim(cfg.mjcf_scene_path, S.SimConfig(), n_envs=n)
    simu.set_kernel(kernel)
    robot = S.SimRobot(simu, None, cfg)
    grip = S.SimGripper(simu, default_sim_gripper_cfg())
    cm = compile_mjcf(pu.SCENE)
    arm = [f"fr3_joint{i}_0" for i in range(1, 8)]
    osims = [O.Sim(cm, arm, arm, "attachment_site_0", "base_0", FR3_Q_HOME, None, "finger_joint1_0", "actuator8_0") for _ in range(n)]
    rng = np.random.default_rng(8)
    odd = np.arange(n) % 2 == 1

    def check(stage, conv=False):
        q, v, st, gs = simu.qpos, simu.qvel, robot.get_state(), grip.get_state()
        w, grasped = grip.get_normalized_width(), grip.is_grasped()
        steps = simu.convergence_steps() if conv else None
        done = simu.is_converged() if conv else None
        for e, o in enumerate(osims):
            assert np.abs(q[e][:7] - np.asarray(o.qpos)[:7]).max() < TOL and np.abs(q[e][7:] - np.asarray(o.qpos)[7:]).max() < FINGER_TOL, (stage, e)
            assert np.abs(v[e][:7] - np.asarray(o.qvel)[:7]).max() < VTOL, (stage, e)
            assert bool(st.is_moving[e]) == bool(o.s.is_moving) and bool(st.is_arrived[e]) == bool(o.s.is_arrived), (stage, e)
            assert bool(st.ik_success[e]) == bool(o.s.ik_success) and bool(st.collision[e]) == bool(o.s.robot_collision), (stage, e)
            assert np.abs(st.target_angles[e] - np.asarray(o.s.target_angles[:7])).max() < TOL, (stage, e)
            assert abs(w[e] - o.gripper_get_normalized_width()) < 1e-7 and bool(grasped[e]) == o.gripper_is_grasped(), (stage, e)
            assert abs(gs.last_commanded_width[e] - o.s.last_commanded_width) < 1e-15, (stage, e)
            if conv:
                assert int(steps[e]) == int(o.s.convergence_steps) and bool(done[e]) == bool(o.s.converged), (stage, e)

    # reset everything, then home the robot (RobotEnv.reset order)
    simu.reset(); robot.reset(); grip.reset()
    for o in osims:
        o.reset(); o.robot_reset(); o.gripper_reset()
    simu.step(1)
    [o.step(1) for o in osims]
    check("reset")
    # SimRobot::get_base_pose_in_world_coordinates (SimRobot.cpp:207-213; mjData.xpos / xquat of the base body: valid after a step)
    bp, obp = robot.get_base_pose_in_world_coordinates(), osims[0].get_base_pose()
    assert np.abs(bp.translation() - obp.translation()).max() < 1e-15 and np.abs(bp.rotation_q() - obp.rotation_q()).max() < 1e-15
    # joint targets on the odd environments only, gripper shut on the even ones
    tgt = np.tile(FR3_Q_HOME, (n, 1)) + rng.uniform(-0.08, 0.08, size=(n, 7))
    robot.set_joint_position(tgt, mask=odd)
    grip.shut(mask=~odd)
    for e, o in enumerate(osims):
        if odd[e]:
            o.set_joint_position(tgt[e])
        else:
            o.gripper_grasp()
    simu.step(34)
    [o.step(34) for o in osims]
    check("masked targets")
    simu.step_until_convergence()
    [o.step_until_convergence() for o in osims]
    check("until convergence", conv=True)
    # hard joint reset on the even ones, move_home on the odd ones, a half-open gripper everywhere
    hard = np.tile(FR3_Q_HOME, (n, 1)) + rng.uniform(-0.2, 0.2, size=(n, 7))
    robot.set_joints_hard(hard, mask=~odd)
    robot.move_home(mask=odd)
    grip.set_normalized_width(np.full(n, 0.5))
    for e, o in enumerate(osims):
        if odd[e]:
            o.move_home()
        else:
            o.set_joints_hard(hard[e])
        o.gripper_set_normalized_width(0.5)
    simu.step(51)
    [o.step(51) for o in osims]
    check("hard reset / move_home")
    # sim reset of the odd environments only
    simu.reset(mask=odd)
    robot.reset(mask=odd)
    grip.open(mask=odd)
    for e, o in enumerate(osims):
        if odd[e]:
            o.reset(); o.robot_reset(); o.gripper_open()
    simu.step(17)
    [o.step(17) for o in osims]
    check("masked sim reset")
    with pytest.raises(ValueError):
        grip.set_normalized_width(1.5)
    simu.close()


@pytest.mark.parametrize("async_control", [True, False])
def test_snapshot_restore_replays_bit_for_bit(async_control):
    """rcsh_sim_get_state / set_state: restoring a snapshot and repeating the same env-steps reproduces observations,
    joint state, flags and substep counts exactly (the launch is deterministic: no atomics, fixed reduction orders)."""
    from parity_util import make_vec_env, synthetic_actions

    n = 40
    venv = make_vec_env(n, async_control)
    j, g = synthetic_actions(n, 7, 31)
    venv.reset()
    for t in range(2):
        venv.step({"joints": j[t], "gripper": g[t]})
    snap = venv.sim.get_state()

    def run():
        out = []
        for t in range(2, 7):
            obs, _, _, trunc, info = venv.step({"joints": j[t], "gripper": g[t]})
            out.append((obs["joints"].copy(), obs["tquat"].copy(), venv.sim.qpos.copy(), venv.sim.qvel.copy(), trunc.copy(),
                        info["substeps"].copy(), info["is_sim_converged"].copy()))
        return out

    first = run()
    venv.sim.set_state(snap)
    second = run()
    for a, b in zip(first, second):
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
    with pytest.raises(ValueError):
        venv.sim.set_state(snap[:-8])
    # (round 6: the blob begins with a header -- layout version, n_envs, number of state fields; one that says otherwise is refused)
    bad = snap.copy()
    bad[8] ^= 1
    with pytest.raises((ValueError, RuntimeError)):
        venv.sim.set_state(bad)
    venv.close()


def test_error_behaviour_of_the_boundary():
    """SURVEY 8b "Errors": bad names raise RuntimeError with the reference's wording (SimRobot.cpp:57-93,
    SimGripper.cpp:16-28), width / force out of range raises ValueError (SimGripper.cpp:80-83), IK failure is NOT an error
    (ik_success = False in the state), an unsupported scene is refused at construction, handles survive errors."""
    import dataclasses

    from rcs_amd import sim as S
    from rcs_amd.envs import default_sim_gripper_cfg, default_sim_robot_cfg

    cfg = default_sim_robot_cfg("fr3_empty_world")
    simu = S.Sim(cfg.mjcf_scene_path, S.SimConfig(), n_envs=4)
    bad = dataclasses.replace(cfg, joints=[*cfg.joints[:-1], "no_such_joint"])
    with pytest.raises(RuntimeError, match="No joint named no_such_joint"):
        S.SimRobot(simu, None, bad)
    bad = dataclasses.replace(cfg, attachment_site="no_such_site")
    with pytest.raises(RuntimeError, match="No site named no_such_site"):
        S.SimRobot(simu, None, bad)
    robot = S.SimRobot(simu, None, cfg)  # the handle is still usable
    gcfg = default_sim_gripper_cfg()
    with pytest.raises(RuntimeError, match="No actuator named"):
        S.SimGripper(simu, dataclasses.replace(gcfg, actuator="no_such_actuator"))
    grip = S.SimGripper(simu, gcfg)
    for w, f in ((1.2, 0.0), (-0.1, 0.0), (0.5, -1.0)):
        with pytest.raises(ValueError):
            grip.set_normalized_width(w, f)
    simu.reset(); robot.reset(); grip.reset(); simu.step(1)
    # an unreachable Cartesian target: no exception, ik_success False, joint targets untouched
    before = robot.get_state().target_angles.copy()
    far = np.tile(np.array([3.0, 0.0, 0.5, 0.0, 0.0, 0.0, 1.0]), (4, 1))
    robot.set_cartesian_position(far)
    st = robot.get_state()
    assert not st.ik_success.any() and np.array_equal(st.target_angles, before)
    simu.step(5)
    assert np.isfinite(simu.qpos).all()
    with pytest.raises((RuntimeError, ValueError)):
        S.Sim(cfg.mjcf_scene_path, S.SimConfig(), n_envs=0)
    simu.close()


def test_collision_flags_match_oracle():
    """Reference collision pins (test_sim_envs.py:136-151,347-360) through the HIP path: folded arm in JOINTS mode,
    TCP target below the ground in Cartesian mode; every flag and substep count equals the oracle's."""
    from parity_util import make_oracle_envs, make_vec_env
    from rcs_amd.envs import ControlMode

    n = 8
    venv = make_vec_env(n, False, gripper=True, relative=False)
    oenvs = make_oracle_envs(n, False, gripper=True, relative=False)
    venv.reset()
    [o.reset() for o in oenvs]
    rng = np.random.default_rng(0)
    for step in range(3):
        a = np.tile(np.array([0, 1.78, 0, -1.45, 0, 0, 0.0]), (n, 1)) + rng.uniform(-0.02, 0.02, size=(n, 7)) * (np.arange(n)[:, None] > 0)
        g = np.ones(n, dtype=np.float32)
        _, _, _, trunc, info = venv.step({"joints": a, "gripper": g})
        for e, o in enumerate(oenvs):
            _, _, _, otr, oi = o.step({"joints": a[e], "gripper": g[e]})
            assert bool(trunc[e]) == bool(otr) and bool(info["collision"][e]) == bool(oi["collision"])
            assert int(info["substeps"][e]) == int(o.sim.s.convergence_steps)
        if step == 0:
            assert info["collision"].all() and info["ik_success"].all()
    assert trunc.all()
    venv.close()

    venv = make_vec_env(n, False, gripper=True, relative=False, control_mode=ControlMode.CARTESIAN_TRPY)
    oenvs = make_oracle_envs(n, False, gripper=True, relative=False, mode="xyzrpy")
    obs, _ = venv.reset()
    [o.reset() for o in oenvs]
    a = obs["xyzrpy"].copy()
    a[:, 0], a[:, 2] = 0.4, -0.05
    _, _, _, trunc, info = venv.step({"xyzrpy": a, "gripper": np.zeros(n, dtype=np.float32)})
    assert info["collision"].all() and info["ik_success"].all()
    for e, o in enumerate(oenvs):
        _, _, _, otr, oi = o.step({"xyzrpy": a[e], "gripper": 0})
        assert bool(trunc[e]) == bool(otr) and bool(info["collision"][e]) == bool(oi["collision"])
        assert int(info["substeps"][e]) == int(o.sim.s.convergence_steps)
    venv.close()


@pytest.mark.parametrize("async_control", [True, False])
def test_xarm7_joints_with_dry_friction(async_control, kernel):
    """Second archetype: 7-dof xArm7, <general> affine actuators with force ranges, no gripper, and a
    dry-friction row (frictionloss = 1) on every joint -- the Huber-cost rows of the constraint solve.  Team kernel only."""
    rep = run_joint_rollout_parity(n_envs=48, n_steps=6 if async_control else 3, async_control=async_control, seed=9, robot="xarm7")
    assert rep["max_abs_qpos"] < TOL and rep["max_abs_qvel"] < VTOL and rep["max_abs_obs"] < TOL, rep
    assert rep["flag_mismatches"] == 0 and rep["substep_mismatches"] == 0, rep


@pytest.mark.parametrize("n_over", [1, 3])
def test_xarm7_friction_rows_and_limit_rows_together(n_over, kernel):
    """Dry-friction rows and penetrating joint-limit rows in the same constraint solves: the zone candidates of the
    factorisation slot carry the limit rows' states, the serial routine's line search crosses both kinds of boundary."""
    from parity_util import run_xarm7_at_joint_limits_parity

    rep = run_xarm7_at_joint_limits_parity(n_envs=48, n_calls=8, k=17, seed=6, n_over=n_over)
    assert rep["max_rows"] >= n_over, rep
    assert rep["max_abs_qpos"] < 1e-9 and rep["max_abs_qvel"] < 1e-7, rep


def test_xarm7_cartesian_relative_clik(kernel):
    """The CLIK on the xArm7 chain (7 joints, attachment site on link7) + its friction-row physics."""
    rep = run_cartesian_rollout_parity(n_envs=24, n_steps=5, async_control=True, seed=17, mode="xyzrpy", robot="xarm7")
    assert rep["max_abs_target"] < TOL and rep["max_abs_qpos"] < TOL and rep["max_abs_tquat"] < TOL, rep
    assert rep["flag_mismatches"] == 0, rep


@pytest.mark.parametrize("n_envs", [1, 5, 33])
def test_ragged_batch_sizes(n_envs):
    """Batches that do not fill a wavefront / a team group / the 8-workgroup XCD rounding."""
    rep = run_joint_rollout_parity(n_envs=n_envs, n_steps=3, async_control=True, seed=21, gripper=True)
    assert rep["max_abs_qpos"] < TOL and rep["max_abs_finger"] < FINGER_TOL and rep["flag_mismatches"] == 0, rep


def test_headline_batch_is_position_independent():
    """BASELINE configs[1] size (4096 environments), size-independent property: an environment's trajectory depends
    on its own inputs only.  64 distinct action streams are tiled 64x over the batch; every copy must equal the first
    BIT FOR BIT wherever it sits (lane, team, wavefront, XCD), a masked reset must leave the unmasked environments
    untouched, and the first 64 environments are checked against the oracle."""
    from parity_util import make_oracle_envs, make_vec_env, synthetic_actions

    n, base, steps = 4096, 64, 4
    joints, grip = synthetic_actions(base, steps, 5)
    venv = make_vec_env(n, True)
    oenvs = make_oracle_envs(base, True)
    venv.reset()
    [o.reset() for o in oenvs]
    for t in range(steps):
        obs, _, _, _, info = venv.step({"joints": np.tile(joints[t], (n // base, 1)), "gripper": np.tile(grip[t], n // base)})
        q, v = venv.sim.qpos, venv.sim.qvel
        for arr in (q, v, obs["tquat"], obs["joints"], info["gripper_width"]):
            a = np.asarray(arr).reshape(n // base, base, -1)
            assert np.array_equal(a, np.broadcast_to(a[0], a.shape)), "replicas diverged"
        for e, o in enumerate(oenvs):
            o.step({"joints": joints[t, e], "gripper": grip[t, e]})
            assert np.abs(q[e][:7] - o.sim.qpos[:7]).max() < TOL and np.abs(q[e][7:] - o.sim.qpos[7:]).max() < FINGER_TOL
    # masked reset: only every third environment restarts
    before_q = venv.sim.qpos.copy()
    mask = (np.arange(n) % 3 == 0)
    venv.reset(mask=mask)
    after_q = venv.sim.qpos
    assert np.array_equal(after_q[~mask], before_q[~mask])
    assert np.array_equal(after_q[mask], np.broadcast_to(after_q[0], after_q[mask].shape))
    assert not np.array_equal(after_q[0], before_q[0])
    venv.close()


def test_free_box_matches_oracle(kernel):
    """The free box of fr3_simple_pick_up (plane-box contacts, elliptic cones, noslip) against the oracle: tumbling,
    sliding, popping out of the floor and coming to rest.  The contact problem is strictly convex, both sides solve it
    to ~1e-13; what is left is round-off amplified by impacts."""
    import parity_util as pu

    rep = pu.run_free_box_parity(n_envs=32, n_calls=12, k=25, seed=3)
    assert rep["max_ncon"] == 4 and rep["zones"] == {0, 1, 2}, rep  # separating, sliding and sticking contacts all occurred
    assert rep["max_abs_pos"] < 1e-6 and rep["max_abs_quat"] < 1e-5 and rep["max_abs_vel"] < 1e-3, rep
    assert rep["max_abs_robot_qpos"] < 1e-9, rep
    # a batch that fills neither a wavefront (4 environments) nor a grid row (8 workgroups)
    rep = pu.run_free_box_parity(n_envs=5, n_calls=6, k=25, seed=9)
    assert rep["max_abs_pos"] < 1e-6 and rep["max_abs_quat"] < 1e-5 and rep["max_abs_vel"] < 1e-3 and rep["max_abs_robot_qpos"] < 1e-9, rep


@pytest.mark.parametrize("async_control", [True, False])
def test_pick_task_env_matches_oracle(kernel, async_control):
    """rcs/FR3SimplePickUpSim-v0 (FR3SimplePickUpSimEnvCreator: RandomCubePos reset, relative TRPY control through the
    CLIK, PickCubeSuccessWrapper reward / success) against the oracle's restatement of that wrapper stack."""
    import parity_util as pu

    rep = pu.run_pick_task_parity(n_envs=16, n_steps=6 if async_control else 3, seed=1, episodes=2, async_control=async_control)
    assert rep["flag_mismatches"] == 0, rep
    assert rep["max_abs_obs"] < TOL and rep["max_abs_box"] < 1e-6 and rep["max_abs_reward"] < 1e-6, rep
    assert 0.0 < rep["min_reward"] and rep["max_reward"] < 1.0, rep


@pytest.mark.parametrize("double_precision", [False, True], ids=["float32", "float64"])
def test_depth_render_matches_oracle(kernel, double_precision):
    """SimCameraSet depth images (wrist camera on the moving hand, fixed bird's-eye camera) of the pick-up scene: the
    ray-casting kernel against the numpy restatement (float64) on the oracle's frames.  Pixels on a silhouette edge may fall on
    either side (a ray grazing a hull face decides by round-off); everything else is identical to the millimetre in the
    double-precision instantiation, and in the product's float32 rays -- the reference's image is a float32 z-buffer quantised
    to uint16 millimetres -- within ONE millimetre (a depth within 1e-7 of a millimetre boundary truncates to the other side).
    The fused uint16 path equals the reference's Python conversion of the raw depth buffer bit for bit in both."""
    import parity_util as pu

    rep = pu.run_depth_render_parity(n_envs=6, width=64, height=48, seed=2, double_precision=double_precision)
    assert rep["fused_mismatch"] == 0, rep
    if double_precision:
        assert rep["mismatched_mm"] <= 2e-4 * rep["pixels"], rep
    else:
        assert rep["mismatched_mm"] <= 5e-3 * rep["pixels"] and rep["off_by_more_than_1mm"] <= 4e-4 * rep["pixels"], rep
    assert rep["max_abs_extrinsics"] < 1e-12, rep
    assert rep["robot_pixels"] > 100 and rep["wrist_min_mm"] < 700, rep  # robot and cube from above; the floor under the hand in the wrist view
    # colour frames of the same rays: flat-shaded shape colours, identical to the restatement up to the rounding of a level
    # (pixels whose ray fell on the other side of a silhouette edge aside)
    # (float32: a ray that meets a hull within 1e-7 of the edge between two faces may take the other face's shade)
    assert rep["rgb_off_by_more_than_one"] <= (0 if double_precision else 2e-4 * rep["pixels"]), rep
    assert rep["rgb_mismatched_pixels"] <= (3e-3 if double_precision else 8e-3) * rep["pixels"], rep
    assert rep["green_pixels"] > 20 and rep["white_pixels"] > 100, rep  # the cube and the robot are in the pictures
    assert rep["capsule_pixels"] > 10, rep  # and so is the wrist camera's body (a capsule on the hand), seen from above


def test_free_and_tracking_camera_types():
    """CameraType.free is an untouched mjvCamera (camera.cpp:36-47): it looks at the world origin from 2 m away, azimuth 90,
    elevation -45 degrees; CameraType.tracking fails as it does in the reference (no body to track is ever set)."""
    from rcs_amd import sim as S
    from rcs_amd.camera import CameraType, SimCameraConfig, SimCameraSet
    from rcs_amd.envs import default_sim_robot_cfg

    cfg = default_sim_robot_cfg("fr3_empty_world")
    simu = S.Sim(cfg.mjcf_scene_path, S.SimConfig(), n_envs=2)
    S.SimRobot(simu, None, cfg)
    simu.step(1)
    cs = SimCameraSet(simu, {"free": SimCameraConfig(identifier="", type=CameraType.free, resolution_width=32, resolution_height=24)}, physical_units=True)
    f = cs.get_latest_frames().frames["free"].camera
    ext = f.depth.extrinsics[0]  # world -> camera (z in front)
    cam_pos = -ext[:3, :3].T @ ext[:3, 3]
    assert np.allclose(cam_pos, [0.0, -np.sqrt(2.0), np.sqrt(2.0)], atol=1e-12)  # 2 m from the origin, 45 degrees up, on -y
    assert np.allclose(ext[:3, :3] @ (np.zeros(3) - cam_pos), [0, 0, 2.0], atol=1e-12)  # the origin is straight ahead
    centre = f.depth.data[0, 12, 16, 0]
    assert f.color.data.shape == (2, 24, 32, 3) and f.color.data.dtype == np.uint8 and 500 < centre < 2100  # the robot's base, or the floor behind it
    with pytest.raises(RuntimeError, match="track body id"):
        SimCameraSet(simu, {"t": SimCameraConfig(identifier="wrist_0", type=CameraType.tracking)})
    simu.close()


def test_camera_set_semantics():
    """Frame-set buffer and geometry of SimCameraSet (src/sim/camera.cpp:54-140, python/rcs/camera/sim.py:88-115): frames
    rendered at one simulation time share a frame set, clear_buffer forgets it, intrinsics follow fovy and the
    resolution, the bird's-eye view of an empty floor is the plane's depth along each pixel's ray."""
    from rcs_amd import sim as S
    from rcs_amd.camera import SimCameraConfig, SimCameraSet
    from rcs_amd.envs import default_sim_robot_cfg

    cfg = default_sim_robot_cfg("fr3_empty_world")
    simu = S.Sim(cfg.mjcf_scene_path, S.SimConfig(), n_envs=3)
    S.SimRobot(simu, None, cfg)
    W, H = 40, 30
    cs = SimCameraSet(simu, {"top": SimCameraConfig(identifier="bird_eye_cam", resolution_width=W, resolution_height=H)}, physical_units=True)
    assert cs.buffer_size() == 0 and cs.camera_names == ["top"] and cs.name_to_identifier == {"top": "bird_eye_cam"}
    f1 = cs.get_latest_frames()
    f2 = cs.get_latest_frames()
    assert cs.buffer_size() == 1  # same simulation time: the frame set is overwritten, not appended
    simu.step(1)
    f3 = cs.get_latest_frames()
    assert cs.buffer_size() == 2 and float(f3.avg_timestamp[0]) == 0.002
    assert cs.get_timestamp_frames(f1.avg_timestamp) is not None and cs.get_timestamp_frames(np.full(3, 7.0)) is None
    cs.clear_buffer()
    assert cs.buffer_size() == 0
    d = f1.frames["top"].camera.depth
    c = f1.frames["top"].camera.color
    assert d.data.shape == (3, H, W, 1) and d.data.dtype == np.uint16 and c.data.shape == (3, H, W, 3) and c.data.dtype == np.uint8
    assert np.array_equal(d.data, f2.frames["top"].camera.depth.data)
    K = d.intrinsics
    assert np.isclose(K[0, 0], 0.5 * H / np.tan(np.deg2rad(45) / 2)) and K[0, 0] == K[1, 1] and K[0, 2] == (W - 1) / 2 and K[1, 2] == (H - 1) / 2
    # corner pixels see only floor: depth along the view axis of the plane z = 0 from the camera pose in the extrinsics
    E = np.linalg.inv(d.extrinsics[0])  # camera (z forward, y down) in the world
    for (r, c) in ((0, 0), (0, W - 1), (H - 1, 0)):
        ray = E[:3, :3] @ np.array([(c - K[0, 2]) / K[0, 0], (r - K[1, 2]) / K[1, 1], 1.0])
        z = -E[2, 3] / ray[2]
        assert abs(int(d.data[0, r, c, 0]) - 1000 * z) <= 1.0, (r, c, d.data[0, r, c, 0], z)
    simu.close()


def test_env_with_cameras_returns_depth_frames(kernel):
    """SimEnvCreator(cameras=...) / FR3SimplePickUpSimEnvCreator(cam_list=...): the observation carries the cameras'
    depth frames (CameraSetWrapper.observation, base.py:633-674), rendered on demand at the step's simulation time."""
    from rcs_amd.envs import FR3SimplePickUpSimEnvCreator

    env = FR3SimplePickUpSimEnvCreator()(n_envs=4, resolution=(32, 24), cam_list=["wrist_0", "bird_eye_cam"])
    obs, info = env.reset()
    assert info["camera_available"] and set(obs["frames"]) == {"wrist_0", "bird_eye_cam"}
    d0 = obs["frames"]["bird_eye_cam"]["depth"]
    assert d0["data"].shape == (4, 24, 32, 1) and d0["data"].dtype == np.uint16 and d0["extrinsics"].shape == (4, 4, 4)
    rng = np.random.default_rng(0)
    obs, reward, term, trunc, info = env.step({"xyzrpy": rng.uniform(-0.05, 0.05, (4, 6)), "gripper": np.ones(4)})
    assert np.allclose(info["frame_timestamp"], 2 * 0.002 + 17 * 0.002) and env.camera_set.buffer_size() == 2
    w0, w1 = obs["frames"]["wrist_0"]["depth"], d0
    assert not np.array_equal(w0["extrinsics"], env.camera_set.get_timestamp_frames(np.full(4, 0.004)).frames["wrist_0"].camera.depth.extrinsics)  # the hand moved
    env.close()


def test_error_behaviour_of_free_body_task_and_camera_calls():
    """The entry points added for the pick-up scene refuse what they cannot do, with the reference's exception types:
    unknown names -> the KeyError / RuntimeError the mujoco bindings / mj_name2id wrappers raise, wrong call order ->
    RuntimeError, out-of-range arguments -> ValueError; handles stay usable."""
    import ctypes as C

    from rcs_amd import _lib
    from rcs_amd import sim as S
    from rcs_amd.camera import CameraType, SimCameraConfig, SimCameraSet
    from rcs_amd.envs import default_sim_gripper_cfg, default_sim_robot_cfg

    empty = S.Sim(default_sim_robot_cfg("fr3_empty_world").mjcf_scene_path, S.SimConfig(), n_envs=2)
    with pytest.raises(KeyError, match="box_joint"):       # sim.data.joint("box_joint") on a scene without it
        empty.free_joint_qpos("box_joint")
    L = empty._L
    q = np.zeros((2, 7))
    assert L.rcsh_sim_get_free_qpos(empty._h, _lib.ptr(q)) == _lib.RCSH_ERR_STATE
    t = _lib.PickTaskDesc()
    S.SimRobot(empty, None, default_sim_robot_cfg("fr3_empty_world"))
    assert L.rcsh_env_configure_pick_task(empty._h, C.byref(t)) == _lib.RCSH_ERR_STATE   # no free box in this scene
    assert L.rcsh_camera_render(empty._h, 0, None, None, None) == _lib.RCSH_ERR_ARG      # no such camera
    with pytest.raises(RuntimeError, match="No camera named nope"):
        SimCameraSet(empty, {"x": SimCameraConfig(identifier="nope")})
    with pytest.raises(RuntimeError, match="track body id"):  # what mjv_updateCamera says to the reference's tracking camera
        SimCameraSet(empty, {"x": SimCameraConfig(identifier="bird_eye_cam", type=CameraType.tracking)})
    empty.step(1)
    empty.close()

    cfg = default_sim_robot_cfg("fr3_simple_pick_up")
    pick = S.Sim(cfg.mjcf_scene_path, S.SimConfig(), n_envs=2)
    with pytest.raises(ValueError, match="removed"):  # the one-lane-per-environment kernel of ABI 1
        pick.set_kernel("lane")
    box = _lib.make_free_box_desc(pick.model)
    assert L.rcsh_sim_add_free_box(pick._h, C.byref(box)) == _lib.RCSH_ERR_STATE          # already attached by Sim()
    S.SimRobot(pick, None, cfg)
    S.SimGripper(pick, default_sim_gripper_cfg())
    obs = np.zeros((2, 21))
    assert L.rcsh_env_reset_task(pick._h, None, _lib.ptr(q), _lib.ptr(obs), None, None) == _lib.RCSH_ERR_STATE  # rcsh_env_configure first
    pick.set_free_joint_qpos("box_joint", [0.5, 0.0, 0.1, 1, 0, 0, 0])
    pick.step(3)
    assert np.isfinite(pick.free_joint_qpos("box_joint")).all() and (pick.free_joint_qvel("box_joint")[:, 2] < 0).all()  # falling
    pick.close()


def test_task_env_with_random_object_pos(kernel):
    """SimTaskEnvCreator(random_pos_args=...) (creators.py:160-167): RandomObjectPos places the cube around the given pose
    instead of RandomCubePos' ISO-cube centre; options["RandomObjectPos.init_object_pose"] replaces the pose for good."""
    from rcs_amd import common
    from rcs_amd.envs import SimTaskEnvCreator, default_sim_robot_cfg

    rc = default_sim_robot_cfg(scene="fr3_simple_pick_up")
    with pytest.raises(TypeError):
        SimTaskEnvCreator()(rc, random_pos_args={"joint_name": "box_joint"}, n_envs=2)
    with pytest.raises(KeyError):
        SimTaskEnvCreator()(rc, random_pos_args={"joint_name": "no_such_joint", "init_object_pose": common.Pose()}, n_envs=2)
    pose = common.Pose(translation=np.array([0.55, -0.05, 0.0288]), quaternion=np.array([0.0, 0.0, 0.0, 1.0]))
    env = SimTaskEnvCreator()(rc, random_pos_args={"joint_name": "box_joint", "init_object_pose": pose, "include_position": False}, n_envs=3)
    env.reset()
    q = env.sim.free_joint_qpos("box_joint")
    assert np.allclose(q[:, :2], [0.55, -0.05], atol=1e-6) and np.allclose(q[:, 3:], [1, 0, 0, 0], atol=1e-9) and (np.abs(q[:, 2] - 0.0288) < 1e-3).all()
    other = common.Pose(translation=np.array([0.4, 0.1, 0.0288]), quaternion=np.array([0.0, 0.0, 0.0, 1.0]))
    env.reset(options={"RandomObjectPos.init_object_pose": other})
    env.reset()
    assert np.allclose(env.sim.free_joint_qpos("box_joint")[:, :2], [0.4, 0.1], atol=1e-6)
    env.close()


def test_xarm7_with_free_box_and_camera(kernel):
    """The builder-authored xArm7 + cube scene (no reference scene exists): friction-row Newton for the
    arm and the cube's contact solve in one launch, and a depth frame of both."""
    import parity_util as pu

    rep = pu.run_xarm7_box_parity(n_envs=16, n_calls=8, k=25, seed=4)
    assert rep["max_ncon"] == 4 and {1, 2} <= rep["zones"], rep
    assert rep["max_abs_box"] < 1e-6 and rep["max_abs_robot_qpos"] < 1e-6, rep
    assert rep["depth_mismatch"] <= 3 and rep["cube_pixels"] > 16, rep


def test_xarm7_box_batch_of_baseline_config_3(kernel):
    """BASELINE configs[3] at its per-GPU size -- 8192 environments split over 2 GPUs = 4096 each -- AS WRITTEN: "xarm7 pick-place
    scene with object contacts + SimCameraSet depth render" is scenes/xarm7_pick_world (the xArm7 with dry joint friction and a
    two-finger gripper next to the pick-up cube; the reference ships no such scene).  All 4096 environments pinch their cube, lift
    it and hold it, with a depth AND colour frame of the fixed camera per launch; 16 distinct cube placements tiled 256x: every
    copy equals the first bit for bit wherever it sits, and every cube ends up in the air.  (The same script against the oracle:
    test_xarm7_picks_the_cube_up_matches_oracle.)"""
    from rcs_amd import sim as S
    from rcs_amd.camera import SimCameraConfig, SimCameraSet
    from rcs_amd.envs import xarm7_pick_sim_gripper_cfg, xarm7_pick_sim_robot_cfg

    n, base = 4096, 16
    cfg = xarm7_pick_sim_robot_cfg()
    simu = S.Sim(cfg.mjcf_scene_path, S.SimConfig(), n_envs=n)
    assert simu.resolve_robot_contacts
    robot = S.SimRobot(simu, None, cfg)
    grip = S.SimGripper(simu, xarm7_pick_sim_gripper_cfg())
    cs = SimCameraSet(simu, {"side": SimCameraConfig(identifier="side_cam", resolution_width=32, resolution_height=24)}, physical_units=True)
    rng = np.random.default_rng(6)
    qb = np.tile(np.array([0.40, 0.0, 0.0288, 0, 0, 0, 1.0]), (base, 1))
    qb[:, 0] += rng.uniform(-0.004, 0.004, base)
    qb[:, 1] += rng.uniform(-0.004, 0.004, base)
    yaw = rng.uniform(-0.1, 0.1, base)
    qb[:, 3], qb[:, 6] = np.cos((np.pi + yaw) / 2), np.sin((np.pi + yaw) / 2)
    simu.reset(); robot.reset(); grip.reset()
    simu.set_free_joint_qpos("box_joint", np.tile(qb, (n // base, 1)))
    simu.step(1)

    def tiled(arr):
        a = np.asarray(arr).reshape(n // base, base, -1)
        return np.array_equal(a, np.broadcast_to(a[0], a.shape))

    from rcs_amd import common

    down = common.Pose(rotation=np.diag([1.0, -1.0, -1.0])).rotation_q()

    def move(z):
        robot.set_cartesian_position(np.tile(np.concatenate([[0.40, 0.0, z - 0.12], down]), (n, 1)))  # (robot frame: base 0.12 m up)

    def run(k):
        simu.step(k)
        f = cs.get_latest_frames().frames["side"].camera
        for arr in (simu.qpos, simu.qvel, simu.free_joint_qpos("box_joint"), simu.free_joint_qvel("box_joint"), f.depth.data, f.color.data):
            assert tiled(arr), "replicas diverged"
        return f

    grip.open(); move(0.20); run(500)
    move(0.035); run(700)
    grip.shut(); run(250)
    assert (grip.get_normalized_width() > 0.3).all()  # the fingers stopped on the cube
    move(0.30); run(600)
    f = run(200)
    z = simu.free_joint_qpos("box_joint")[:, 2]
    assert (z > 0.25).all() and np.isfinite(simu.qpos).all(), (z.min(), z.max())  # every one of the 4096 cubes hangs in its gripper
    assert (f.depth.data[:base] < 2500).any()
    simu.close()


def test_pick_task_batch_is_position_independent(kernel):
    """The pick-up task at BASELINE's batch size (4096 environments): 32 distinct (cube placement, action stream) pairs tiled
    128x over the batch; every copy equals the first BIT FOR BIT wherever it sits (lane, team, wavefront, XCD) -- robot,
    cube, reward, depth pixels -- and a masked reset leaves the unmasked environments' cubes untouched."""
    from rcs_amd.envs import FR3SimplePickUpSimEnvCreator

    n, base, steps = 4096, 32, 3
    env = FR3SimplePickUpSimEnvCreator()(n_envs=n, resolution=(16, 12), cam_list=["wrist_0"])
    rng = np.random.default_rng(11)
    np.random.seed(3)
    box = np.tile(env.draw_box_qpos()[:base], (n // base, 1))
    obs, info = env.reset(options={"box_qpos": box})

    def tiled(arr):
        a = np.asarray(arr).reshape(n // base, base, -1)
        return np.array_equal(a, np.broadcast_to(a[0], a.shape))

    assert tiled(obs["frames"]["wrist_0"]["depth"]["data"]) and tiled(env.sim.free_joint_qpos("box_joint"))
    for t in range(steps):
        a = np.tile(np.concatenate([rng.uniform(-0.05, 0.05, (base, 3)), rng.uniform(-0.1, 0.1, (base, 3))], axis=1), (n // base, 1))
        g = np.tile(rng.uniform(0, 1, base).astype(np.float32), n // base)
        obs, reward, term, trunc, info = env.step({"xyzrpy": a, "gripper": g})
        for arr in (env.sim.qpos, env.sim.qvel, env.sim.free_joint_qpos("box_joint"), env.sim.free_joint_qvel("box_joint"), reward,
                    info["box_qpos"], obs["tquat"], obs["frames"]["wrist_0"]["depth"]["data"]):
            assert tiled(arr), "replicas diverged"
    before = env.sim.free_joint_qpos("box_joint").copy()
    mask = np.arange(n) % 5 == 0
    env.reset(options={"box_qpos": box}, mask=mask)
    after = env.sim.free_joint_qpos("box_joint")
    assert np.array_equal(after[~mask], before[~mask]) and not np.array_equal(after[mask], before[mask])
    env.close()


@pytest.mark.parametrize("async_control", [True, False])
def test_arm6_joints(async_control, kernel):
    """Third archetype, `Topo<6, false>`: the builder-authored 6-dof arm (scenes/arm6_empty_world).  Its joints turn about
    y and about a skew axis, with anchors off the link origin -- the general-axis path that no FR3 / xArm7 joint takes."""
    rep = run_joint_rollout_parity(n_envs=40, n_steps=6 if async_control else 3, async_control=async_control, seed=13, robot="arm6")
    assert rep["max_abs_qpos"] < TOL and rep["max_abs_qvel"] < VTOL and rep["max_abs_obs"] < TOL, rep
    assert rep["flag_mismatches"] == 0 and rep["substep_mismatches"] == 0, rep


def test_arm6_cartesian_relative_clik(kernel):
    """The CLIK on the 6-dof chain (6 Jacobian columns, attachment site on the last link)."""
    rep = run_cartesian_rollout_parity(n_envs=24, n_steps=5, async_control=True, seed=19, mode="xyzrpy", robot="arm6")
    assert rep["max_abs_target"] < TOL and rep["max_abs_qpos"] < TOL and rep["max_abs_tquat"] < TOL, rep
    assert rep["flag_mismatches"] == 0, rep


@pytest.mark.parametrize("async_control", [True, False])
def test_ur5e_joints(async_control, kernel):
    """The UR5e-proportioned 6-dof arm (scenes/ur5e_empty_world: public DH lengths and link masses, robots_meta_config's UR5e
    home pose and limits; the reference ships no UR5e model): `Topo<6, false>` on a second chain, actuators holding the arm
    against gravity (no gravity compensation)."""
    rep = run_joint_rollout_parity(n_envs=40, n_steps=6 if async_control else 3, async_control=async_control, seed=27, robot="ur5e")
    assert rep["max_abs_qpos"] < TOL and rep["max_abs_qvel"] < VTOL and rep["max_abs_obs"] < TOL, rep
    assert rep["flag_mismatches"] == 0 and rep["substep_mismatches"] == 0, rep


def test_ur5e_cartesian_relative_clik(kernel):
    rep = run_cartesian_rollout_parity(n_envs=24, n_steps=5, async_control=True, seed=29, mode="xyzrpy", robot="ur5e")
    assert rep["max_abs_target"] < TOL and rep["max_abs_qpos"] < TOL and rep["max_abs_tquat"] < TOL, rep
    assert rep["flag_mismatches"] == 0, rep


@pytest.mark.parametrize("async_control", [True, False])
def test_so101_joints_and_gripper(async_control, kernel):
    """Fourth archetype, `Topo<5, true>`: the SO-101-proportioned 5-dof arm with the two-finger gripper
    (scenes/so101_empty_world; fingers on lanes 5 and 6, so the frame scan selects its receiving lanes by predicate instead
    of by DPP bank).  Its home pose -- robots_meta_config's SO101 entry -- sits 0.3 % of the range inside two joint limits and
    the arm is not gravity-compensated, so limit rows come and go in almost every substep."""
    rep = run_joint_rollout_parity(n_envs=40, n_steps=6 if async_control else 3, async_control=async_control, seed=31, robot="so101")
    assert rep["max_abs_qpos"] < TOL and rep["max_abs_qvel"] < VTOL and rep["max_abs_obs"] < TOL and rep["max_abs_finger"] < FINGER_TOL, rep
    assert rep["flag_mismatches"] == 0 and rep["substep_mismatches"] == 0 and rep["max_abs_gripper_width"] < 1e-7, rep


def test_so101_cartesian_relative_clik(kernel):
    """The CLIK on 5 joints: 6 x 5 Jacobian, damped 6 x 6 normal equations; most 6-dof targets are unreachable, so the iteration
    runs to its cap and `ik_success` is false -- in the oracle and in the kernel alike (flags compared bit for bit)."""
    rep = run_cartesian_rollout_parity(n_envs=16, n_steps=4, async_control=True, seed=33, mode="xyzrpy", robot="so101")
    assert rep["flag_mismatches"] == 0, rep
    assert rep["max_abs_qpos"] < 1e-7 and rep["max_abs_tquat"] < 1e-7, rep  # (a thousand CLIK iterations amplify round-off)


@pytest.mark.parametrize("async_control", [True, False])
def test_seven_dof_arm_without_gripper_or_friction(async_control, kernel):
    """`Topo<7, false>` without the friction variant (team kernel) and on the lane kernel: the xArm7 chain with
    frictionloss = 0 -- a combination no shipped scene selects (the xArm7 has friction, the FR3 scene has fingers)."""
    rep = run_joint_rollout_parity(n_envs=40, n_steps=5 if async_control else 3, async_control=async_control, seed=21, robot="xarm7_nofric")
    assert rep["max_abs_qpos"] < TOL and rep["max_abs_qvel"] < VTOL and rep["max_abs_obs"] < TOL, rep
    assert rep["flag_mismatches"] == 0 and rep["substep_mismatches"] == 0, rep


@pytest.mark.parametrize("robot", ["fr3_fric", "arm6_fric"])
def test_joint_friction_on_the_other_archetypes(robot, kernel):
    """Dry joint friction where the shipped scenes have none: FR3 + hand (friction rows together with the fingers' coupling
    equality, their limit rows and the tendon actuator -- `newton_rows<Topo<7,true>, FRIC>`) and the 6-dof arm."""
    for async_control in (True, False):
        rep = run_joint_rollout_parity(n_envs=32, n_steps=5 if async_control else 2, async_control=async_control, seed=23, robot=robot)
        # (stiction: a 15 g finger under 0.5 N of dry friction sticks wherever it stops -- the quadratic zone of its Huber row is
        # 1e-6 m/s^2 wide, so round-off decides the zone and the resting place to ~2e-6 m; nothing else in the suite is this loose)
        assert rep["max_abs_qpos"] < 1e-8 and rep["max_abs_qvel"] < 1e-6 and rep["max_abs_obs"] < 1e-8 and rep["max_abs_finger"] < 1e-5, rep
        assert rep["flag_mismatches"] == 0 and rep["substep_mismatches"] == 0, rep


def test_seven_dof_arm_without_gripper_cartesian(kernel):
    """The CLIK kernels of `Topo<7, false>` on both kernel variants (the shipped xArm7 scene only reaches the team one)."""
    rep = run_cartesian_rollout_parity(n_envs=24, n_steps=4, async_control=True, seed=25, mode="tquat", robot="xarm7_nofric")
    assert rep["max_abs_target"] < TOL and rep["max_abs_qpos"] < TOL and rep["max_abs_tquat"] < TOL, rep
    assert rep["flag_mismatches"] == 0, rep


def test_device_pointer_forms_equal_host_forms(kernel):
    """The resident-rollout entry points (`*_dev`, device pointers from rcsh_dev_alloc / upload / download -- what a host language
    without a HIP binding uses) against the host-buffer forms of the same calls, bit for bit: task reset and step, a depth frame."""
    import ctypes as C

    from rcs_amd import _lib
    from rcs_amd.envs import FR3SimplePickUpSimEnvCreator

    n, W, H = 6, 24, 16
    envs = [FR3SimplePickUpSimEnvCreator()(n_envs=n, resolution=(W, H), cam_list=["wrist_0"]) for _ in range(2)]
    host, dev = envs
    L, h = dev._L, dev.sim._h
    assert L.rcsh_sim_num_envs(h) == n and L.rcsh_sim_stream(h) is not None
    cfg = (C.c_int32 * 4)()
    assert L.rcsh_sim_get_config(h, C.byref(cfg, 0), C.byref(cfg, 4), C.byref(cfg, 8), C.byref(cfg, 12)) == 0 and list(cfg) == [1, 0, 30, 500]

    def dalloc(nbytes):
        p = C.c_void_p()
        _lib.check(L.rcsh_dev_alloc(h, nbytes, C.byref(p)))
        return p

    def up(p, a):
        a = np.ascontiguousarray(a)
        _lib.check(L.rcsh_dev_upload(h, p, C.c_void_p(a.ctypes.data), a.nbytes))

    def down(p, a):
        _lib.check(L.rcsh_dev_download(h, C.c_void_p(a.ctypes.data), p, a.nbytes))
        return a

    np.random.seed(5)
    box = host.draw_box_qpos()
    rng = np.random.default_rng(6)
    act = np.concatenate([rng.uniform(-0.05, 0.05, (n, 3)), rng.uniform(-0.1, 0.1, (n, 3))], axis=1)
    grip = rng.uniform(0, 1, n).astype(np.float32)
    ow = dev.obs_width
    d_box, d_act, d_grip = dalloc(n * 7 * 8), dalloc(n * 6 * 8), dalloc(n * 4)
    d_obs, d_info, d_gw, d_sub, d_task, d_img = dalloc(n * ow * 8), dalloc(n * 8), dalloc(n * 8), dalloc(n * 4), dalloc(n * 9 * 8), dalloc(n * W * H * 2)
    up(d_box, box); up(d_act, act); up(d_grip, grip)
    # reset
    h_obs, h_info = host.reset(options={"box_qpos": box})
    _lib.check(L.rcsh_env_reset_task_dev(h, None, d_box, d_obs, d_info, d_gw))
    obs = down(d_obs, np.zeros((n, ow)))
    assert np.array_equal(obs[:, :7], h_obs["tquat"]) and np.array_equal(down(d_gw, np.zeros(n)), h_info["gripper_width"])
    # step
    h_obs, h_rew, h_term, h_trunc, h_info = host.step({"xyzrpy": act, "gripper": grip})
    _lib.check(L.rcsh_env_step_task_dev(h, d_act, d_grip, d_obs, d_info, d_gw, d_sub, d_task))
    obs, task = down(d_obs, np.zeros((n, ow))), down(d_task, np.zeros((n, 9)))
    assert np.array_equal(obs[:, :7], h_obs["tquat"]) and np.array_equal(obs[:, 7:14], h_obs["joints"])
    assert np.array_equal(task[:, :7], h_info["box_qpos"]) and np.array_equal(task[:, 7], h_rew) and np.array_equal(task[:, 8] != 0, h_term)
    assert np.array_equal(down(d_sub, np.zeros(n, dtype=np.int32)), h_info["substeps"])
    # depth frame of the same instant
    _lib.check(L.rcsh_camera_render_dev(h, dev.camera_set._ids["wrist_0"], None, d_img, None))
    _lib.check(L.rcsh_sim_synchronize(h))
    img = down(d_img, np.zeros((n, H, W), dtype=np.uint16))
    assert np.array_equal(img, h_obs["frames"]["wrist_0"]["depth"]["data"][..., 0])
    # mj_resetData of the box alone
    _lib.check(L.rcsh_sim_reset_free_box(h))
    assert np.array_equal(dev.sim.free_joint_qpos("box_joint"), np.tile([0.44, 0.1, 0.03, 0, 0, 0, 1.0], (n, 1)))
    for p in (d_box, d_act, d_grip, d_obs, d_info, d_gw, d_sub, d_task, d_img):
        _lib.check(L.rcsh_dev_free(h, p))
    [e.close() for e in envs]


def test_grasp_lift_swing_matches_oracle(kernel):
    """Robot <-> cube contacts: a scripted pinch through the 1:1 API -- finger pads against the cube
    (box-box, up to 36 contacts), one constraint problem over the robot's 9 and the cube's 6 dofs, noslip -- then a lift and a
    swing above PickCubeSuccessWrapper's 1.002 m.  Kernel vs oracle stage by stage; flags (robot / gripper collision,
    is_grasped, convergence step counts) bit for bit."""
    from parity_util import run_grasp_parity

    rep = run_grasp_parity(n_envs=4, seed=0)
    assert rep["max_ncon"] >= 36 and rep["coupled_substeps"] > 1000 and rep["max_noslip"] >= 1, rep
    # (measured 7e-13 / 2e-12 / 6e-10 / 2e-8 over three seeds of eight environments: profiles/r2_soak_parity.log)
    assert rep["max_abs_qpos"] < 1e-10 and rep["max_abs_qvel"] < 1e-9 and rep["max_abs_box"] < 1e-8 and rep["max_abs_box_vel"] < 5e-7, rep
    assert rep["flag_mismatches"] == 0, rep
    st = rep["stages"]
    assert (st["closed"]["box_z"] < 0.03).all() and (st["lifted"]["box_z"] > 0.28).all(), st       # on the floor, then in the hand
    assert (st["held"]["box_z"] > 1.002).all() and (st["released"]["box_z"] < st["held"]["box_z"] - 0.3).all(), st
    assert ((st["held"]["width"] > 0.3) & (st["held"]["width"] < 0.5)).all(), st                  # fingers stopped by the 32 mm cube


def test_hard_pinch_is_independent_of_the_launch_split(kernel):
    """A placement whose closing pads made the coupled solve's line search cycle (tests/test_contacts_cpu.py): inside a
    launch of several substeps the solve started from the previous minimiser, stalled, and the cube was squashed through the
    floor, while launches of one substep agreed with the oracle.  Any split must, now that the line search is safeguarded."""
    from parity_util import run_hard_pinch_parity

    rep = run_hard_pinch_parity()
    assert 2 <= rep["max_newton"] <= 12, rep
    for ch, r in rep["splits"].items():
        assert r["qpos"] < 1e-9 and r["qvel"] < 1e-7 and r["box"] < 1e-8 and r["box_vel"] < 1e-6, (ch, rep)
        assert (r["box_z"] > 0.0275).all(), (ch, rep)


def test_batch_scale_pinch_is_launch_split_independent(kernel):
    """BASELINE's batch size, every environment in contact at once: 4096 cubes at random offsets pinched, lifted and released
    with the stepping cut into launches of 17 and of 100 substeps.  Inside a launch the coupled solve starts elsewhere than
    after a fresh launch, so agreement of EVERY environment says that none of the ~2.7 million solves stalled or ran into its
    cap -- a property the oracle cannot check at this size (the first run of this kind found the line search's cycle)."""
    from parity_util import run_split_consistency

    rep = run_split_consistency(4096, 17, 100)
    for tag, r in rep.items():
        assert r["envs_over_1e-8"] == 0 and r["max_dq"] < 1e-10 and r["max_dbox"] < 1e-9, (tag, rep)
    assert rep["closed"]["box_z"][0] > 0.027 and rep["lifted"]["box_z"][0] > 0.28 and rep["released"]["box_z"][1] < 0.03, rep


def test_pick_task_reaches_success(kernel):
    """rcs/FR3SimplePickUpSim-v0's wrapper stack (RandomCubePos, PickCubeSuccessWrapper) with absolute joint actions: the
    scripted pinch ends in `success` / `terminated` with reward 1, in the kernel and in the oracle alike."""
    from parity_util import run_pick_success_parity

    rep = run_pick_success_parity(n_envs=3, seed=1)
    assert rep["flag_mismatches"] == 0 and rep["truncated"] == 0, rep
    assert rep["success_steps"] >= 3 * 10 and rep["grasped_steps"] > 100 and rep["max_box_z"] > 1.002, rep
    assert rep["max_abs_obs"] < 1e-8 and rep["max_abs_box"] < 1e-7 and rep["max_abs_reward"] < 1e-8, rep
    assert rep["contact_overflows"] == 0, rep  # no contact phase ran out of contact (48) or link (4) slots: info["contact_overflow"]


def test_pybind_module_equals_ctypes_host_layer(kernel):
    """The compiled pybind11 binding (extensions/rcs_hip: rcs_hip._core.sim, the reference-side plugin) drives the C-ABI
    exactly like the ctypes host layer: same calls, same numbers bit for bit; exception types as the reference raises them."""
    import os
    import sys

    import parity_util as pu
    from rcs_amd import sim as S
    from rcs_amd.envs import default_sim_gripper_cfg, default_sim_robot_cfg
    from rcs_env_oracle import FR3_Q_HOME

    sys.path.insert(0, os.path.join(pu.ROOT, "extensions", "rcs_hip"))
    import rcs_hip
    from rcs_hip import _core

    n = 8
    cfg = default_sim_robot_cfg("fr3_empty_world")
    a = S.Sim(cfg.mjcf_scene_path, S.SimConfig(), n_envs=n)
    a.set_kernel(kernel)
    ra, ga = S.SimRobot(a, None, cfg), S.SimGripper(a, default_sim_gripper_cfg())
    os.environ["RCSH_KERNEL"] = kernel
    try:
        b = _core.sim.Sim(rcs_hip.model_tables(a.model), n)
    finally:
        os.environ.pop("RCSH_KERNEL")
    rc = _core.sim.SimRobotConfig()
    rc.add_id("0")
    rc.q_home = np.asarray(FR3_Q_HOME)
    rb = _core.sim.SimRobot(b, None, rc)
    gc = _core.sim.SimGripperConfig()
    gc.add_id("0")
    gb = _core.sim.SimGripper(b, gc)
    rng = np.random.default_rng(5)
    for simu, rob, grp in ((a, ra, ga), (b, rb, gb)):
        simu.reset(); rob.reset(); grp.reset()
        simu.step(1)
    odd = np.arange(n) % 2 == 1
    tgt = np.tile(FR3_Q_HOME, (n, 1)) + rng.uniform(-0.1, 0.1, (n, 7))
    ra.set_joint_position(tgt, mask=odd); rb.set_joint_position(tgt, mask=odd)
    ga.shut(mask=~odd); gb.shut(mask=~odd)
    a.step(34); b.step(34)
    assert np.array_equal(a.qpos, b.qpos) and np.array_equal(a.qvel, b.qvel)
    a.step_until_convergence(); b.step_until_convergence()
    assert np.array_equal(a.qpos, b.qpos) and np.array_equal(a.is_converged(), b.is_converged()) and np.array_equal(a.convergence_steps(), b.convergence_steps())
    sa, sb = ra.get_state(), rb.get_state()
    assert np.array_equal(sa.target_angles, sb.target_angles) and np.array_equal(sa.is_arrived, sb.is_arrived) and np.array_equal(sa.is_moving, sb.is_moving)
    assert np.array_equal(ra.get_cartesian_position(), rb.get_cartesian_position())
    assert np.array_equal(ga.get_normalized_width(), gb.get_normalized_width()) and np.array_equal(ga.is_grasped(), gb.is_grasped())
    pose = ra.get_cartesian_position()
    pose[:, 0] += 0.05
    ra.set_cartesian_position(pose); rb.set_cartesian_position(pose)
    a.step(17); b.step(17)
    assert np.array_equal(a.qpos, b.qpos)
    assert np.allclose(rb.to_pose_in_world_coordinates(rb.to_pose_in_robot_coordinates(pose)), pose, atol=1e-15)
    with pytest.raises(ValueError):
        gb.set_normalized_width(1.5)  # std::invalid_argument in the reference (SimGripper.cpp:80-83)
    bad = _core.sim.SimRobotConfig()
    with pytest.raises(RuntimeError, match="No geom named fr3_link0_collision"):
        _core.sim.SimRobot(b, None, bad)  # names without the "_0" suffix: the reference's runtime_error
    a.close()
    del rb, gb, b


def test_robot_on_the_floor_resolved_contacts(kernel):
    """Robot <-> floor contacts as FORCES (opt-in in scenes without a free body): the reference's folded-arm collision case
    and variations; hull / pad / capsule contacts with the plane, the coupled solve without a box (phantom box), noslip."""
    from parity_util import run_floor_contact_parity

    rep = run_floor_contact_parity(n_envs=8, seed=2)
    assert rep["coupled_substeps"] > 500 and rep["max_ncon"] >= 2 and rep["collisions"] == 8, rep
    assert rep["max_abs_qpos"] < 1e-7 and rep["max_abs_qvel"] < 1e-5 and rep["flag_mismatches"] == 0, rep
    assert rep["tracking_error"] > 0.05, rep  # the floor keeps the arm from reaching its target


@pytest.mark.parametrize("scene,resolve", [("fr3_empty_world", None), ("fr3_empty_world", True), ("fr3_simple_pick_up", None)])
def test_self_collision_flags_match_oracle(scene, resolve, kernel):
    """SimRobot / SimGripper collision callbacks see contacts between two geoms of the robot (fingers, pads
    and hand against links 1 and 2 when the arm folds onto itself).  All three instantiations that carry the detection: the
    lean kernel, the contact-resolving one without a free body, and the pick-up scene's."""
    from parity_util import run_self_collision_parity

    rep = run_self_collision_parity(n_envs=48, seed=1, scene=scene, resolve=resolve)
    assert rep["flag_mismatches"] == 0 and rep["substep_mismatches"] == 0 and rep["max_abs_qpos"] < TOL, rep
    assert rep["self_only"] >= 8 and rep["robot_hits"] >= 4 and rep["gripper_hits"] >= 4, rep


def test_rate_driven_cameras_match_oracle(kernel):
    """SURVEY 8 row f4, the rendering leg of Sim::step: SimCameraSet(render_on_demand=False) delivers frames at the cameras'
    frame rates from inside Sim.step and step_until_convergence, first frame in the first substep after construction / reset;
    timestamps per environment equal the restated callback rule exactly, pixels equal the numpy ray-caster's."""
    from parity_util import run_rate_driven_camera_parity

    rep = run_rate_driven_camera_parity(n_envs=4, seed=3)
    assert rep["timestamp_mismatches"] == 0 and rep["camera_set_mismatches"] == 0 and rep["max_abs_qpos"] < TOL, rep
    assert rep["events"] >= 4 * 12 and rep["latest_timestamp_ok"] and rep["obs_keys"] == ["bird_eye_cam", "wrist_0"], rep
    assert rep["mismatched_mm"] <= 5e-4 * rep["pixels"] and rep["rgb_off_by_more_than_one"] == 0, rep


def test_cube_against_the_robot_base(kernel):
    """Collision geoms welded to the world (link 0's hull) against the free cube: contacts between the world body and the cube
    that are NOT the floor's.  The cube is thrown at the base from all around, spinning; it bounces off in the kernel as in the
    oracle, in EVERY environment to round-off.  (Round 2 let 3 of 12 environments differ by centimetres: the portal refinement
    broke exact support ties -- a portal normal is perpendicular to a box edge by construction -- by the sign of round-off;
    ties are broken by rule now, in the oracle and in the kernel, and the refinement is compiled without multiply-add
    contraction: oracle rcs_contact.c SUPPORT_TIE, kernel contact_team.h kSupportTie.)"""
    from parity_util import run_cube_against_base_parity

    for seed in (5, 11, 15):
        rep = run_cube_against_base_parity(seed=seed)
        assert rep["base_contact_envs"] >= 6 and rep["max_abs_robot_qpos"] < TOL, rep
        assert (rep["env_pos_err"] < 1e-9).all() and rep["max_abs_quat"] < 1e-8, rep
        assert (rep["final_radius"] > 0.08).all(), rep  # nowhere did the cube pass through the base


def test_pybind_camera_set_equals_ctypes_host_layer(kernel):
    """The compiled module's SimCameraSet (rcs_hip._core.sim: the reference's class and method names) against the ctypes
    host's: the same rgb / depth buffers bit for bit on demand, the same frame-set rule (one set per simulation time), and
    -- render_on_demand=False -- the same frames with the same timestamps from inside Sim.step."""
    import os
    import sys

    import parity_util as pu

    sys.path.insert(0, os.path.join(pu.ROOT, "extensions", "rcs_hip"))
    import rcs_hip
    from rcs_hip import _core
    from rcs_amd import sim as S
    from rcs_amd.camera import SimCameraConfig, SimCameraSet
    from rcs_amd.envs import default_sim_robot_cfg
    from rcs_amd.mjcf import compile_mjcf
    from rcs_env_oracle import FR3_Q_HOME

    n, W, H = 3, 32, 24
    cfg = default_sim_robot_cfg("fr3_simple_pick_up")
    cm = compile_mjcf(cfg.mjcf_scene_path)
    tgt = np.tile(np.array([0.1, -0.6, 0.1, -2.2, 0.1, 1.7, 0.9]), (n, 1)) + 0.05 * np.arange(n)[:, None]

    def host(on_demand):
        simu = S.Sim(cfg.mjcf_scene_path, S.SimConfig(), n_envs=n)
        robot = S.SimRobot(simu, None, cfg)
        cams = {"w": SimCameraConfig(identifier="wrist_0", frame_rate=30, resolution_width=W, resolution_height=H),
                "b": SimCameraConfig(identifier="bird_eye_cam", frame_rate=10, resolution_width=W, resolution_height=H)}
        cs = SimCameraSet(simu, cams, physical_units=True, render_on_demand=on_demand, max_framesets=100)
        robot.set_joint_position(tgt)
        simu.step(40)
        if on_demand:
            raw = {k: cs.render_raw_rgbd(k) for k in cams}
            out = {k: (raw[k][0].reshape(n, -1), raw[k][1].reshape(n, -1)) for k in cams}, simu.time
        else:
            out = [(ev["timestamp"], {k: (ev["color"][k].reshape(n, -1), ev["depth"][k].reshape(n, -1), ev["have"][k]) for k in ev["have"]}) for ev in cs._buffer], None
        simu.close()
        return out

    def bound(on_demand):
        simu = _core.sim.Sim(rcs_hip.model_tables(cm), n, 0, rcs_hip.free_box_tables(cm))
        simu.set_render_scene(rcs_hip.render_tables(cm, os.path.dirname(cfg.mjcf_scene_path)))
        rc = _core.sim.SimRobotConfig()
        rc.add_id("0")
        rc.q_home = np.asarray(FR3_Q_HOME)
        robot = _core.sim.SimRobot(simu, None, rc)
        cams = {"w": _core.sim.SimCameraConfig("wrist_0", 30, W, H), "b": _core.sim.SimCameraConfig("bird_eye_cam", 10, W, H)}
        cs = _core.sim.SimCameraSet(simu, cams, render_on_demand=on_demand)
        robot.set_joint_position(tgt)
        simu.step(40)
        return simu, cs

    (want, t_want) = host(True)
    simu, cs = bound(True)
    assert cs.buffer_size() == 0
    fs = cs.get_latest_frameset()
    fs2 = cs.get_latest_frameset()
    assert cs.buffer_size() == 1 and np.array_equal(fs.timestamp, t_want) and np.array_equal(fs2.timestamp, t_want)
    for k in ("w", "b"):
        assert np.array_equal(fs.color_frames[k], want[k][0]) and np.array_equal(fs.depth_frames[k], want[k][1])
    assert cs.get_timestamp_frameset(fs.timestamp) is not None and cs.get_timestamp_frameset(fs.timestamp + 1) is None
    cs.clear_buffer()
    assert cs.buffer_size() == 0 and cs._sim is simu
    del cs, simu

    events, _ = host(False)
    simu, cs = bound(False)
    assert cs.buffer_size() == len(events) >= 3
    for i, (ts, cams) in enumerate(events):
        fs = cs.get_timestamp_frameset(ts) if not np.isnan(ts).any() else None
        assert fs is not None and sorted(fs.color_frames) == sorted(cams)
        for k, (rgb, depth, have) in cams.items():
            assert np.array_equal(fs.rendered[k].astype(bool), have)
            assert np.array_equal(fs.color_frames[k][have], rgb[have]) and np.array_equal(fs.depth_frames[k][have], depth[have])
    with pytest.raises(RuntimeError, match="No camera named"):
        _core.sim.SimCameraSet(simu, {"x": _core.sim.SimCameraConfig("nope", 0, 8, 8)})
    with pytest.raises(RuntimeError, match="track body id"):
        _core.sim.SimCameraSet(simu, {"x": _core.sim.SimCameraConfig("wrist_0", 0, 8, 8, _core.sim.CameraType.tracking)})


def test_c_host_equals_python_host(kernel, tmp_path):
    """The drop-in boundary from a host that is not Python: examples/c_host/rollout.c -- plain C against include/rcs_hip.h and
    librcs_hip.so, the scene's tables as C initialisers -- creates the batch, attaches robot and gripper, configures the
    Gymnasium loop and steps it; the same calls through the ctypes host give the same observations bit for bit."""
    import os
    import subprocess
    import sys

    import parity_util as pu
    from parity_util import make_vec_env

    root = pu.ROOT
    inc = subprocess.run([sys.executable, os.path.join(root, "tools", "export_model_c.py")], check=True, capture_output=True, text=True).stdout
    (tmp_path / "model.inc").write_text(inc)
    libdir = os.path.join(root, "robot-control-stack_amd", "rcs_amd")
    exe = str(tmp_path / "rollout")
    subprocess.run(["gcc", "-O2", "-std=c11", "-I" + os.path.join(root, "include"), "-I" + str(tmp_path), "-iquote", str(tmp_path),
                    os.path.join(root, "examples", "c_host", "rollout.c"), "-L" + libdir, "-lrcs_hip", "-Wl,-rpath," + libdir, "-lm", "-o", exe], check=True)
    n, steps = 64, 5
    out = subprocess.run([exe, str(n), str(steps)], check=True, capture_output=True, text=True).stdout.strip().splitlines()
    assert len(out) == 2 * (steps + 1)

    state = 0x9E3779B97F4A7C15

    def lcg_unit():
        nonlocal state
        state = (state * 6364136223846793005 + 1442695040888963407) % (1 << 64)
        return (state >> 11) / 9007199254740992.0 * 2.0 - 1.0

    env = make_vec_env(n, True, gripper=True, relative=True)
    obs, info = env.reset()
    rows = iter(out)
    mov = float(np.deg2rad(5))
    for t in range(steps + 1):
        if t > 0:
            a = np.zeros((n, 7))
            g = np.zeros(n, dtype=np.float32)
            for e in range(n):
                for k in range(7):
                    a[e, k] = mov * lcg_unit()
                g[e] = 1.0 if lcg_unit() > 0 else 0.0
            obs, _, _, _, info = env.step({"joints": a, "gripper": g})
        for e in (0, n - 1):
            tok = next(rows).split()
            assert tok[:4] == ["step", str(t), "env", str(e)]
            got = np.array([float.fromhex(x) for x in tok[tok.index("obs") + 1:]])
            want = np.concatenate([obs["tquat"][e], obs["joints"][e], obs["xyzrpy"][e], [obs["gripper"][e]]])
            assert np.array_equal(got, want), (t, e, got - want)
            assert float.fromhex(tok[tok.index("width") + 1]) == float(info["gripper_width"][e])
            if t > 0:
                assert int(tok[tok.index("substeps") + 1]) == int(info["substeps"][e]) == 17
    env.close()


def test_masked_reset_keeps_the_reset_frames_of_rate_driven_cameras(kernel):
    """env.reset(mask) with SimCameraSet(render_on_demand=False): the reference renders in the first substep after a reset
    (clocks at -1 / rate, sim.cpp:131-137), so the reset environments must come back with a frame stamped with their new
    time.  (Advisor, round 2: the observation-only pass over the UNmasked rows used to wipe those records.)"""
    from rcs_amd import sim as S
    from rcs_amd.camera import SimCameraConfig, SimCameraSet
    from rcs_amd.envs import ControlMode, RelativeTo, default_sim_gripper_cfg, default_sim_robot_cfg
    from rcs_amd.envs.creators import VecSimEnv

    n = 6
    cfg = default_sim_robot_cfg("fr3_empty_world")
    simu = S.Sim(cfg.mjcf_scene_path, S.SimConfig(async_control=True, frequency=30), n_envs=n)
    robot = S.SimRobot(simu, None, cfg)
    grip = S.SimGripper(simu, default_sim_gripper_cfg())
    cams = {"wrist_0": SimCameraConfig(identifier="wrist_0", frame_rate=30, resolution_width=16, resolution_height=12)}
    cs = SimCameraSet(simu, cams, physical_units=True, render_on_demand=False)
    env = VecSimEnv(simu, robot, grip, ControlMode.JOINTS, float(np.deg2rad(5)), RelativeTo.LAST_STEP, camera_set=cs)
    obs, info = env.reset()
    assert info["camera_available"] and np.allclose(info["frame_timestamp"], 0.002)
    for _ in range(3):
        obs, _, _, _, info = env.step({"joints": np.full((n, 7), 0.01), "gripper": np.ones(n)})
    t_before = simu.time.copy()
    mask = np.array([1, 0, 1, 0, 0, 1], dtype=bool)
    obs, info = env.reset(mask=mask)
    assert info["camera_available"], "the reset environments' first frames were dropped"
    ts = info["frame_timestamp"]
    assert np.allclose(ts[mask], 0.002) and np.all(np.isnan(ts[~mask])), ts  # (clear_buffer: the others wait for their next frame)
    assert np.allclose(simu.time[mask], 0.002) and np.array_equal(simu.time[~mask], t_before[~mask])
    obs, _, _, _, info = env.step({"joints": np.zeros((n, 7)), "gripper": np.ones(n)})
    assert not np.isnan(info["frame_timestamp"]).any()
    env.close()


def test_contact_table_capacity_is_fatal_only_when_contacts_are_resolved(kernel, tmp_path):
    """A scene with more box geoms than the contact phase holds (the two fingers' pads use all ten) still loads and steps --
    the extra geom is merely invisible to geom-geom detection -- and asking for its robot contacts to be RESOLVED is the
    model error.  (Advisor, round 2.)"""
    import shutil

    from rcs_amd import sim as S
    from rcs_amd.envs import default_sim_robot_cfg
    from parity_util import SCENE

    xml = open(SCENE).read()
    marker = "<worldbody>"
    assert marker in xml
    xml = xml.replace(marker, marker + '\n    <geom name="table_block" type="box" size="0.1 0.1 0.02" pos="0.9 0 0.02"/>', 1)
    scene = tmp_path / "scene.xml"
    scene.write_text(xml)
    for extra in ("collision_vertices.npz", "render_hulls.npz"):
        shutil.copy(os.path.join(os.path.dirname(SCENE), extra), tmp_path)
    cfg = default_sim_robot_cfg("fr3_empty_world")
    # (advisor, round 3) the overflow is not silent for DETECTION either: the host is told which geoms geom-geom detection
    # cannot see and warns, and a collision callback cannot be registered on one of them
    # (the block comes first in geom order and takes a box slot: the last finger pad is the geom left out)
    with pytest.warns(RuntimeWarning, match="fingertip_pad_collision_5_right_0"):
        simu = S.Sim(str(scene), S.SimConfig(), n_envs=4)
    assert [simu.model.geom_names[g] for g in simu.undetected_collision_geoms] == ["fingertip_pad_collision_5_right_0"]
    robot = S.SimRobot(simu, None, cfg)
    robot.set_joint_position(np.tile(cfg_home(robot), (4, 1)))
    simu.step_until_convergence()
    assert simu.is_converged().all()
    simu.close()
    with pytest.warns(RuntimeWarning), pytest.raises(RuntimeError, match="box geoms"):
        S.Sim(str(scene), S.SimConfig(), n_envs=4, resolve_robot_contacts=True)
    import copy

    bad = copy.deepcopy(cfg)
    bad.arm_collision_geoms = list(bad.arm_collision_geoms) + ["fingertip_pad_collision_5_right_0"]
    with pytest.warns(RuntimeWarning):
        simu = S.Sim(str(scene), S.SimConfig(), n_envs=4)
    with pytest.raises(RuntimeError, match="not in the contact table"):
        S.SimRobot(simu, None, bad)
    simu.close()


def cfg_home(robot):
    from rcs_amd import common

    return np.asarray(common.sim_robots_meta_config(robot.get_config().robot_type).q_home)[: robot.dof]


@pytest.mark.parametrize("robot", ["fr3", "xarm7"])
def test_compiled_pin_kinematics_match_oracle(robot, kernel):
    """`rcs_hip._core.common.Pin(path, frame_id, urdf=False)` -- the reference's `common.Pin` constructor and
    `Kinematics.forward / inverse` signatures (src/pybind/rcs.cpp:289-300) over rcsh_ik_* -- against the oracle's restatement of
    Pin (src/rcs/Kinematics.cpp:28-82): poses <= 1e-12, joint solutions <= 1e-9 with model.nq entries (quirk Q7), None where
    the CLIK runs into its cap."""
    import sys

    import rcs_oracle as O
    from parity_util import ROOT, SCENE, XARM7_SCENE, make_oracle_envs

    sys.path.insert(0, os.path.join(ROOT, "extensions", "rcs_hip"))
    from rcs_hip import _core

    c = _core.common
    pin = c.Pin(SCENE if robot == "fr3" else XARM7_SCENE, "attachment_site_0" if robot == "fr3" else "attachment_site", False)
    assert isinstance(pin, c.Kinematics)
    o = make_oracle_envs(1, True, gripper=False, relative=False, robot=robot)[0]
    o.reset()
    rng = np.random.default_rng(7)
    q_home = np.asarray(c.robots_meta_config(c.RobotType.FR3 if robot == "fr3" else c.RobotType.XArm7).q_home)
    tcp_o = O.franka_hand_tcp_offset() if robot == "fr3" else O.Pose()
    tcp_c = c.Pose(pose_matrix=c.FrankaHandTCPOffset()) if robot == "fr3" else c.Pose()
    assert np.abs(tcp_c.rotation_q() - tcp_o.rotation_q()).max() < 1e-15
    solved = 0
    for _ in range(12):
        qt = q_home + rng.uniform(-0.25, 0.25, size=q_home.shape)
        f, of = pin.forward(qt, tcp_c), o.sim.ik_forward(qt, tcp_o)
        assert np.abs(f.translation() - of.translation()).max() < 1e-12 and np.abs(f.rotation_q() - of.rotation_q()).max() < 1e-12
        q = pin.inverse(f, q_home, tcp_c)
        oq, _ = o.sim.ik_inverse(O.Pose(translation=of.translation(), quaternion=of.rotation_q()), q_home, tcp_o)
        assert (q is None) == (oq is None)
        if q is not None:
            assert q.shape == oq.shape and np.abs(q - oq).max() < 1e-9
            solved += 1
    assert solved >= 8
    far = c.Pose(translation=np.array([5.0, 0.0, 0.5]))  # out of reach: the CLIK hits its 1000-iteration cap
    assert pin.inverse(far, q_home) is None and pin.forward(q_home).is_close(pin.forward(q_home, c.Pose()))


def test_xarm7_picks_the_cube_up_matches_oracle(kernel):
    """BASELINE configs[3] as written -- "xarm7 pick-place scene with object contacts": the xArm7 of the reference's xarm7.xml
    (dry joint friction on all seven arm joints) with the Franka hand on its flange pinches the pick-up scene's cube, lifts it
    20 cm and holds it; kernel (`k_run_team<Topo<7,true>, FRIC, BOX, CON>`: friction-dof rows inside the coupled 15-dof solve)
    vs oracle: every flag bit-equal, arm joints <= 1e-9, cube pose <= 1e-8 over ~1700 substeps in contact."""
    from parity_util import run_xarm7_pick_parity

    rep = run_xarm7_pick_parity(n_envs=4, seed=0)
    assert rep["flag_mismatches"] == 0 and rep["ik_failures"] == 0, rep
    assert rep["max_abs_qpos"] < TOL and rep["max_abs_qvel"] < VTOL and rep["max_abs_box"] < 1e-8, rep
    assert rep["coupled_substeps"] >= 4 * 1000 and rep["max_ncon"] >= 20, rep
    st = rep["stages"]
    assert (st["down"]["box_z"] < 0.03).all() and (st["lifted"]["box_z"] > 0.25).all() and (st["held"]["box_z"] > 0.25).all(), st  # picked up and held
    assert (st["released"]["box_z"] < 0.06).all() and (st["held"]["width"] > 0.3).all(), st  # dropped again; the fingers stopped on the cube


def test_xarm7_arm_links_rest_on_the_floor_and_show_in_the_frame(kernel):
    """scenes/xarm7_pick_world carries the reference's convex collision mesh on every arm link (xarm7.xml:103-156): with the
    shoulder sent forward the forearm comes down on the floor -- hull-plane contacts on ARM links, resolved in the coupled solve
    next to the arm's dry-friction rows -- kernel vs oracle <= 1e-9 on the joints; the arm comes to rest short of its target,
    held up by the floor; and the fixed camera's depth frame, arm included, equals the numpy ray caster's on the oracle's frames
    (silhouette pixels aside)."""
    from parity_util import run_xarm7_links_on_the_floor_parity

    rep = run_xarm7_links_on_the_floor_parity(n_envs=3, seed=0)
    assert rep["max_abs_qpos"] < TOL and rep["max_abs_qvel"] < VTOL and rep["max_abs_box"] < 1e-8, rep
    assert rep["coupled_substeps"] > 3 * 300 and rep["arm_link_contacts"] > 3 * 300 and rep["max_links_in_contact"] <= 4, rep
    assert (np.abs(rep["qpos"] - rep["target"]).max(axis=1) > 0.05).all(), rep  # the floor is in the way
    assert rep["mismatched_mm"] <= 2e-3 * rep["pixels"] and rep["arm_pixels"] > 3 * 40, rep


def test_render_schedule_grows_with_the_launch_and_rejects_a_second_set(kernel):
    """Advisor, round 2: the render schedule's capacity was frozen at construction (from max_convergence_steps at that moment) and a
    longer launch -- Sim.step(k) with a large k, a raised cap -- failed after the state had advanced; a second rate-driven
    SimCameraSet silently replaced the first one's schedule.  Now the schedule grows before a longer launch, keeping the cameras'
    clocks (no camera becomes due again by re-registration), and a second set is refused."""
    import warnings

    from rcs_amd import sim as S
    from rcs_amd.camera import SimCameraConfig, SimCameraSet
    from rcs_amd.envs import default_sim_gripper_cfg, default_sim_robot_cfg

    n = 3
    cfg = default_sim_robot_cfg("fr3_empty_world")
    simu = S.Sim(cfg.mjcf_scene_path, S.SimConfig(max_convergence_steps=40), n_envs=n)
    S.SimRobot(simu, None, cfg)
    S.SimGripper(simu, default_sim_gripper_cfg())
    cams = {"wrist_0": SimCameraConfig(identifier="wrist_0", frame_rate=30, resolution_width=8, resolution_height=6)}
    cs = SimCameraSet(simu, cams, physical_units=True, render_on_demand=False, max_framesets=10000)
    small = cs._capacity
    with pytest.raises(RuntimeError, match="already has a SimCameraSet"):
        SimCameraSet(simu, cams, physical_units=True, render_on_demand=False)
    with warnings.catch_warnings():
        warnings.simplefilter("error")  # a dropped frame would warn
        simu.step(10)
        first = cs.buffer_size()
        simu.step(1500)  # 3 s of simulated time in one launch: 90 frames at 30 Hz, far beyond the initial capacity
    assert cs._capacity > small and first == 1
    ts = np.array([ev["timestamp"][0] for ev in cs._buffer])
    # (a 30 Hz camera is due when MORE than 1/30 s has passed: every 17th substep of 2 ms, Sim::invoke_rendering_callbacks)
    assert len(ts) == 1 + int((1510 - 1) // 17), len(ts)
    gaps = np.diff(ts)
    assert np.all(gaps > 1 / 30) and np.all(gaps < 1 / 30 + 2 * 0.002 + 1e-12), gaps  # every frame one period (plus <= a substep) after the last
    simu.close()


def test_depth_frames_at_the_size_of_baseline_config_3(kernel):
    """The ray caster at BASELINE configs[3]'s per-GPU size: 4096 environments x one 256 x 256 depth frame of the fixed camera of
    scenes/xarm7_pick_world -- 268 million rays, a million workgroups numbered per XCD, 230 MB of per-environment hull views.  32
    distinct arm poses / cube placements tiled 128x over the batch: every copy of a frame equals the first bit for bit wherever
    its environment sits, and the first two frames equal the numpy ray caster's on the oracle's kinematics (silhouette pixels aside;
    the product's float32 rays: within one millimetre, identical in all but a few pixels per thousand)."""
    import rcs_oracle as O
    import rcs_render_oracle as RO
    from parity_util import XARM7_PICK_SCENE
    from rcs_amd import render
    from rcs_amd import sim as S
    from rcs_amd.camera import SimCameraConfig, SimCameraSet
    from rcs_amd.envs import xarm7_pick_sim_gripper_cfg, xarm7_pick_sim_robot_cfg
    from rcs_amd.mjcf import compile_mjcf
    from rcs_env_oracle import XARM7_PICK as R

    n, base, W = 4096, 32, 256
    cfg = xarm7_pick_sim_robot_cfg()
    simu = S.Sim(cfg.mjcf_scene_path, S.SimConfig(), n_envs=n)
    robot = S.SimRobot(simu, None, cfg)
    S.SimGripper(simu, xarm7_pick_sim_gripper_cfg())
    cs = SimCameraSet(simu, {"side": SimCameraConfig(identifier="side_cam", frame_rate=0, resolution_width=W, resolution_height=W)},
                      physical_units=True, render_on_demand=True)
    rng = np.random.default_rng(3)
    q = np.asarray(R["q_home"]) + rng.uniform(-0.5, 0.5, (base, 7))
    qb = np.tile(np.array([0.40, 0.0, 0.0288, 1.0, 0, 0, 0]), (base, 1))
    qb[:, :2] += rng.uniform(-0.1, 0.1, (base, 2))
    simu.reset(); robot.reset()
    simu.set_free_joint_qpos("box_joint", np.tile(qb, (n // base, 1)))
    robot.set_joints_hard(np.tile(q, (n // base, 1)))
    simu.step(2)
    mm = cs.render_depth_mm("side")
    assert mm.shape == (n, W, W)
    tiles = mm.reshape(n // base, base, W, W)
    assert all(np.array_equal(tiles[k], tiles[0]) for k in range(1, n // base)), "replicas of a frame differ"
    assert len({int(f.astype(np.uint64).sum()) for f in tiles[0]}) == base  # 32 different pictures
    cm = compile_mjcf(XARM7_PICK_SCENE)
    link, pos, rot, fovy = render.camera_in_link(cm, "side_cam")
    for e in range(2):
        o = O.Sim(cm, R["joints"], R["actuators"], R["site"], R["base"], R["q_home"], O.Pose(translation=np.array([0.0, 0.0, 0.1034])),
                  R["gripper_joint"], R["gripper_actuator"], arm_collision_geoms=[], gripper_cfg=R["gripper_cfg"])
        o.reset(); o.robot_reset()
        o.box_qpos = qb[e]
        o.set_joints_hard(q[e])
        o.step(2)
        _, omm, _, _ = RO.render_depth(cs._scene, (link, pos, rot, fovy, W, W), RO.oracle_frames(o, cm))
        diff = np.abs(mm[e].astype(np.int64) - omm.astype(np.int64))
        assert (diff != 0).sum() <= 5e-3 * W * W and (diff > 1).sum() <= 4e-4 * W * W, (int((diff != 0).sum()), int((diff > 1).sum()))
        assert (omm < 1500).sum() > 2000  # the arm fills a good part of the picture
    simu.close()


def test_outline_method_and_plane_walk_draw_the_same_frames(kernel, monkeypatch):
    """The ray caster's two ways through a hull -- the outline as the camera sees it plus the front planes (k_hull_views; the default)
    and the walk over every face plane for both ends of the ray's interval (RCSH_RENDER_OUTLINE=0; also the fallback for hulls
    without a clean edge table) -- on the same 24 poses of the pick-up scene, both cameras, depth and colour: the same pixels
    (a ray that grazes an outline may fall on either side: at most one pixel in ten thousand)."""
    from rcs_amd import sim as S
    from rcs_amd.camera import SimCameraConfig, SimCameraSet
    from rcs_amd.envs import default_sim_gripper_cfg, default_sim_robot_cfg
    from rcs_env_oracle import FR3_Q_HOME

    n, W, H = 24, 96, 64
    rng = np.random.default_rng(12)
    q = np.asarray(FR3_Q_HOME) + rng.uniform(-0.5, 0.5, (n, 7))
    qb = np.tile(np.array([0.5, 0.0, 0.0288, 1, 0, 0, 0.0]), (n, 1))
    qb[:, :2] += rng.uniform(-0.15, 0.15, (n, 2))
    qb[:, 6] = rng.uniform(-1, 1, n)
    frames = {}
    for outline in ("1", "0"):
        monkeypatch.setenv("RCSH_RENDER_OUTLINE", outline)
        cfg = default_sim_robot_cfg("fr3_simple_pick_up")
        simu = S.Sim(cfg.mjcf_scene_path, S.SimConfig(), n_envs=n)
        robot = S.SimRobot(simu, None, cfg)
        S.SimGripper(simu, default_sim_gripper_cfg())
        cs = SimCameraSet(simu, {c: SimCameraConfig(identifier=c, frame_rate=0, resolution_width=W, resolution_height=H) for c in ("wrist_0", "bird_eye_cam")},
                          physical_units=True, render_on_demand=True)
        cs.set_double_precision(True)  # (the two methods round differently: compared where round-off is 1e-16, not 1e-7)
        simu.set_free_joint_qpos("box_joint", qb)
        robot.set_joints_hard(q)
        simu.step(2)
        frames[outline] = {c: cs.render_raw_rgbd(c)[:2] for c in ("wrist_0", "bird_eye_cam")}
        simu.close()
    for c in ("wrist_0", "bird_eye_cam"):
        (rgb1, d1), (rgb0, d0) = frames["1"][c], frames["0"][c]
        assert (d1 != d0).sum() <= 1e-4 * d1.size and (rgb1 != rgb0).any(axis=-1).sum() <= 1e-4 * d1.size, ((d1 != d0).sum(), c)
        assert (d1 < 1.0).mean() > 0.5  # something is in the pictures
