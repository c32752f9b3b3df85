"""Shared helpers for the parity tests, smoke() and bench.py's checks (uses the oracle as the checker)."""

from __future__ import annotations

import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCENE = os.path.join(ROOT, "robot-control-stack_amd", "rcs_amd", "scenes", "fr3_empty_world", "scene.xml")
MAX_JOINT_MOV = float(np.deg2rad(5))


def rpy_distance(a, b) -> float:
    """Angle between the rotations two roll-pitch-yaw triples describe."""
    from rcs_amd.common import Pose

    pa, pb = Pose(rpy_vector=np.asarray(a)), Pose(rpy_vector=np.asarray(b))
    return float((pa * pb.inverse()).total_angle())


def rpy_error(a, b) -> float:
    """Difference of two roll-pitch-yaw observations: COMPONENT-WISE wherever that is meaningful.  The reference's Euler
    extraction (Eigen eulerAngles(2,1,0): yaw in [0, pi], src/rcs/Pose.cpp) jumps by 2 pi / pi where a component sits on
    +-pi or 0 -- at the FR3's home pose roll is exactly +-pi -- so a triple with a component within 1e-6 of such a seam is
    compared as a rotation instead (round-off decides the side of the seam); everywhere else a different-but-equivalent
    Euler branch on the device would show up here as an error of order 1."""
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    near_seam = any(min(abs(abs(x) - s) for s in (0.0, np.pi)) < 1e-6 for x in (*a, *b))
    dist = rpy_distance(a, b)
    return dist if near_seam else max(dist, float(np.abs(a - b).max()))


def synthetic_actions(n_envs: int, n_steps: int, seed: int = 0, dof: int = 7):
    """SURVEY 8d action tensor: joints ~ U(+-5 deg)^dof f64, gripper ~ U(0,1) f32, one RNG stream per env."""
    joints = np.zeros((n_steps, n_envs, dof))
    grip = np.zeros((n_steps, n_envs), dtype=np.float32)
    for e in range(n_envs):
        rng = np.random.default_rng(seed + e)
        joints[:, e, :] = rng.uniform(-MAX_JOINT_MOV, MAX_JOINT_MOV, size=(n_steps, dof))
        grip[:, e] = rng.uniform(0, 1, size=n_steps).astype(np.float32)
    return joints, grip


# kernel variant every make_vec_env() pins (the GPU tests run each of them, see test_gpu_parity.py)
KERNEL = "auto"


XARM7_SCENE = os.path.join(ROOT, "robot-control-stack_amd", "rcs_amd", "scenes", "xarm7_empty_world", "scene.xml")
ARM6_SCENE = os.path.join(ROOT, "robot-control-stack_amd", "rcs_amd", "scenes", "arm6_empty_world", "scene.xml")
UR5E_SCENE = os.path.join(ROOT, "robot-control-stack_amd", "rcs_amd", "scenes", "ur5e_empty_world", "scene.xml")
SO101_SCENE = os.path.join(ROOT, "robot-control-stack_amd", "rcs_amd", "scenes", "so101_empty_world", "scene.xml")
GRIPPER_ROBOTS = ("fr3", "fr3_fric", "so101")


def robot_dof(robot: str) -> int:
    return 5 if robot == "so101" else 6 if robot.startswith(("arm6", "ur5e")) else 7


PICKUP_SCENE = os.path.join(ROOT, "robot-control-stack_amd", "rcs_amd", "scenes", "fr3_simple_pick_up", "scene.xml")


def run_xarm7_box_parity(n_envs=16, n_calls=8, k=25, seed=0, width=48, height=32):
    """xarm7_box_world: the 7-dof arm with dry joint friction next to the free cube (elliptic cones, no noslip pass), kernel vs
    oracle, plus a depth frame of the fixed camera (floor, cube; the xArm7 scene carries no robot shapes)."""
    from rcs_amd import render
    from rcs_amd import sim as S
    from rcs_amd.camera import SimCameraConfig, SimCameraSet
    from rcs_amd.envs import xarm7_sim_robot_cfg
    from rcs_amd.mjcf import compile_mjcf
    import rcs_oracle as O
    import rcs_render_oracle as RO
    from rcs_env_oracle import XARM7

    scene = os.path.join(ROOT, "robot-control-stack_amd", "rcs_amd", "scenes", "xarm7_box_world", "scene.xml")
    cfg = xarm7_sim_robot_cfg("xarm7_box_world")
    simu = S.Sim(cfg.mjcf_scene_path, S.SimConfig(), n_envs=n_envs)
    robot = S.SimRobot(simu, None, cfg)
    cs = SimCameraSet(simu, {"side": SimCameraConfig(identifier="side_cam", resolution_width=width, resolution_height=height)}, physical_units=True)
    cs.set_double_precision(True)  # (this comparison is about WHAT is drawn and when: exact pixels against the float64 restatement)
    cm = compile_mjcf(scene)
    osims = [O.Sim(cm, XARM7["joints"], XARM7["actuators"], XARM7["site"], XARM7["base"], XARM7["q_home"], None, arm_collision_geoms=[]) for _ in range(n_envs)]
    rng = np.random.default_rng(seed)
    qb = np.zeros((n_envs, 7))
    qb[:, 0] = 0.45 + rng.uniform(-0.1, 0.1, n_envs)
    qb[:, 1] = rng.uniform(-0.1, 0.1, n_envs)
    qb[:, 2] = rng.uniform(0.0144, 0.08, n_envs)
    qb[:, 3:] = rng.normal(size=(n_envs, 4)) * np.array([1.0, 0.2, 0.2, 1.0])
    vb = np.concatenate([rng.uniform(-0.4, 0.4, (n_envs, 3)), rng.uniform(-3, 3, (n_envs, 3))], axis=1)
    simu.set_free_joint_qpos("box_joint", qb)
    simu.set_free_joint_qvel("box_joint", vb)
    for e, o in enumerate(osims):
        o.box_qpos, o.box_qvel = qb[e], vb[e]
    rep = {"max_abs_box": 0.0, "max_abs_robot_qpos": 0.0, "max_ncon": 0, "zones": set(), "depth_mismatch": 0, "cube_pixels": 0}
    for _ in range(n_calls):
        tgt = np.asarray(XARM7["q_home"]) + rng.uniform(-0.2, 0.2, (n_envs, 7))
        robot.set_joint_position(tgt)
        simu.step(k)
        qk, qr = simu.free_joint_qpos("box_joint"), simu.qpos
        for e, o in enumerate(osims):
            o.set_joint_position(tgt[e])
            o.step(k)
            bd = o.s.d.box
            rep["max_ncon"] = max(rep["max_ncon"], int(bd.ncon))
            rep["zones"].update(int(z) for z in bd.zone[: bd.ncon])
            rep["max_abs_box"] = max(rep["max_abs_box"], float(np.abs(qk[e] - o.box_qpos).max()))
            rep["max_abs_robot_qpos"] = max(rep["max_abs_robot_qpos"], float(np.abs(qr[e] - o.qpos[:7]).max()))
    data = cs.get_latest_frames().frames["side"].camera.depth.data
    link, pos, rot, fovy = render.camera_in_link(cm, "side_cam")
    for e, o in enumerate(osims):
        dgl, mm, _, _ = RO.render_depth(cs._scene, (link, pos, rot, fovy, width, height), RO.oracle_frames(o, cm))
        rep["depth_mismatch"] += int((data[e, ..., 0] != mm).sum())
        floor_only = RO.render_depth(cs._scene, (link, pos, rot, fovy, width, height), {**RO.oracle_frames(o, cm), -2: (np.eye(3), np.array([0, 0, -9.0]))})[1]
        rep["cube_pixels"] += int((mm != floor_only).sum())
    simu.close()
    return rep


def run_free_box_parity(n_envs=32, n_calls=10, k=25, seed=0, kick=True):
    """The free box of the pick-up scene, kernel vs oracle: every environment starts the box at a random pose near the
    floor (some penetrating, some tilted, some in the air) with a random twist, the arm holds its home pose; Sim.step(k)
    n_calls times.  Returns max abs differences of the box state and the largest contact count / zones seen."""
    from rcs_amd import sim as S
    from rcs_amd.envs import default_sim_gripper_cfg, default_sim_robot_cfg
    from rcs_amd.mjcf import compile_mjcf
    import rcs_oracle as O
    from rcs_env_oracle import FR3_Q_HOME

    cfg = default_sim_robot_cfg("fr3_simple_pick_up")
    simu = S.Sim(cfg.mjcf_scene_path, S.SimConfig(), n_envs=n_envs)
    robot = S.SimRobot(simu, None, cfg)
    S.SimGripper(simu, default_sim_gripper_cfg())
    cm = compile_mjcf(PICKUP_SCENE)
    arm = [f"fr3_joint{i}_0" for i in range(1, 8)]
    osims = [O.Sim(cm, arm, arm, "attachment_site_0", "base_0", FR3_Q_HOME, None, "finger_joint1_0", "actuator8_0") for _ in range(n_envs)]
    rng = np.random.default_rng(seed)
    qb = np.zeros((n_envs, 7))
    qb[:, 0] = 0.45 + rng.uniform(-0.1, 0.1, n_envs)
    qb[:, 1] = rng.uniform(-0.1, 0.1, n_envs)
    qb[:, 2] = rng.uniform(0.0144, 0.06, n_envs)        # RandomCubePos drops it 14.4 mm INTO the floor (sim.py:377)
    qb[:, 3:] = rng.normal(size=(n_envs, 4)) * np.array([1.0, 0.15, 0.15, 1.0])  # unnormalised, mostly yaw
    qb[: n_envs // 4, 3:] = np.array([2 * rng.uniform(size=n_envs // 4) - 1, 0 * qb[: n_envs // 4, 0], 0 * qb[: n_envs // 4, 0], 1 + 0 * qb[: n_envs // 4, 0]]).T
    vb = np.zeros((n_envs, 6))
    if kick:
        vb[:, :3] = rng.uniform(-0.5, 0.5, (n_envs, 3))
        vb[:, 3:] = rng.uniform(-3, 3, (n_envs, 3))
    simu.set_free_joint_qpos("box_joint", qb)
    simu.set_free_joint_qvel("box_joint", vb)
    for e, o in enumerate(osims):
        o.box_qpos, o.box_qvel = qb[e], vb[e]
    rep = {"max_abs_pos": 0.0, "max_abs_quat": 0.0, "max_abs_vel": 0.0, "max_abs_robot_qpos": 0.0, "max_ncon": 0, "zones": set(),
           "max_newton": 0, "max_noslip": 0, "final_speed": 0.0}
    home = np.tile(FR3_Q_HOME, (n_envs, 1))
    for _ in range(n_calls):
        robot.set_joint_position(home)
        simu.step(k)
        qk, vk, qr = simu.free_joint_qpos("box_joint"), simu.free_joint_qvel("box_joint"), simu.qpos
        for e, o in enumerate(osims):
            o.set_joint_position(home[e])
            o.step(k)
            bd = o.s.d.box
            rep["max_ncon"] = max(rep["max_ncon"], int(bd.ncon))
            rep["zones"].update(int(z) for z in bd.zone[: bd.ncon])
            rep["max_newton"] = max(rep["max_newton"], int(bd.newton_iter))
            rep["max_noslip"] = max(rep["max_noslip"], int(bd.noslip_iter))
            rep["max_abs_pos"] = max(rep["max_abs_pos"], float(np.abs(qk[e, :3] - o.box_qpos[:3]).max()))
            rep["max_abs_quat"] = max(rep["max_abs_quat"], float(np.abs(qk[e, 3:] - o.box_qpos[3:]).max()))
            rep["max_abs_vel"] = max(rep["max_abs_vel"], float(np.abs(vk[e] - o.box_qvel).max()))
            rep["max_abs_robot_qpos"] = max(rep["max_abs_robot_qpos"], float(np.abs(qr[e] - o.qpos).max()))
    rep["final_speed"] = float(np.abs(vk).max())
    rep["final_z"] = qk[:, 2].copy()
    simu.close()
    return rep


def run_depth_render_parity(n_envs=6, width=64, height=48, seed=0, cameras=("wrist_0", "bird_eye_cam"), n_calls=3, k=17, double_precision=False):
    """Depth images of the pick-up scene (floor, cube, robot hulls) from the wrist and the bird's-eye camera: ray-casting
    kernel vs the numpy restatement on the oracle's frames, after random joint moves and cube placements."""
    from rcs_amd import sim as S
    from rcs_amd.camera import SimCameraConfig, SimCameraSet
    from rcs_amd.envs import default_sim_gripper_cfg, default_sim_robot_cfg
    from rcs_amd.mjcf import compile_mjcf
    from rcs_amd import render
    import rcs_oracle as O
    import rcs_render_oracle as RO
    from rcs_env_oracle import FR3_Q_HOME

    cfg = default_sim_robot_cfg("fr3_simple_pick_up")
    simu = S.Sim(cfg.mjcf_scene_path, S.SimConfig(), n_envs=n_envs)
    robot = S.SimRobot(simu, None, cfg)
    S.SimGripper(simu, default_sim_gripper_cfg())
    cams = {c: SimCameraConfig(identifier=c, frame_rate=0, resolution_width=width, resolution_height=height) for c in cameras}
    cs = SimCameraSet(simu, cams, physical_units=True, render_on_demand=True)
    cs.set_double_precision(double_precision)
    cm = compile_mjcf(PICKUP_SCENE)
    arm = [f"fr3_joint{i}_0" for i in range(1, 8)]
    osims = [O.Sim(cm, arm, arm, "attachment_site_0", "base_0", FR3_Q_HOME, None, "finger_joint1_0", "actuator8_0") for _ in range(n_envs)]
    rng = np.random.default_rng(seed)
    qb = np.tile(np.array([0.5, 0.0, 0.0288, 1, 0, 0, 0.0]), (n_envs, 1))
    qb[:, 0] += rng.uniform(-0.1, 0.1, n_envs)
    qb[:, 1] += rng.uniform(-0.1, 0.1, n_envs)
    qb[:, 6] = rng.uniform(-1, 1, n_envs)
    simu.set_free_joint_qpos("box_joint", qb)
    for e, o in enumerate(osims):
        o.box_qpos = qb[e]
    capsule_shapes = [g for g in range(len(cs._scene.shape)) if cs._scene.shape[g] == render.SHAPE_CAPSULE]
    rep = {"pixels": 0, "mismatched_mm": 0, "off_by_more_than_1mm": 0, "max_mm_diff": 0, "max_abs_depth_gl": 0.0, "max_abs_extrinsics": 0.0, "robot_pixels": 0,
           "cube_pixels": 0, "floor_pixels": 0, "background_pixels": 0, "fused_mismatch": 0, "rgb_mismatched_pixels": 0, "rgb_max_level_diff": 0,
           "rgb_off_by_more_than_one": 0, "green_pixels": 0, "white_pixels": 0, "colours_seen": set()}
    for _ in range(n_calls):
        tgt = FR3_Q_HOME + rng.uniform(-0.4, 0.4, (n_envs, 7))
        robot.set_joint_position(tgt)
        simu.step(k)
        for e, o in enumerate(osims):
            o.set_joint_position(tgt[e])
            o.step(k)
        frames = cs.get_latest_frames()
        for name in cameras:
            raw, xmat, xpos = cs.render_raw(name)
            fused = cs.render_depth_mm(name)
            data = frames.frames[name].camera.depth.data
            rep["fused_mismatch"] += int((fused != data[..., 0]).sum())
            link, pos, rot, fovy = render.camera_in_link(cm, name)
            for e, o in enumerate(osims):
                dgl, mm, cR, cp, orgb = RO.render_depth(cs._scene, (link, pos, rot, fovy, width, height), RO.oracle_frames(o, cm), colour=True)
                krgb = frames.frames[name].camera.color.data[e]  # [H, W, 3] uint8, rows top-down
                cd = np.abs(krgb.astype(np.int64) - orgb[::-1].astype(np.int64)).max(axis=-1)
                same_surface = data[e, ..., 0] == mm  # (a silhouette pixel that fell on the other side shows another shape)
                rep["rgb_mismatched_pixels"] += int((cd != 0).sum())
                rep["rgb_off_by_more_than_one"] += int(((cd > 1) & same_surface).sum())
                rep["rgb_max_level_diff"] = max(rep["rgb_max_level_diff"], int(cd[same_surface].max()))
                rep["green_pixels"] += int(((krgb[..., 1] > 150) & (krgb[..., 0] < 60)).sum())   # the cube (rgba 0 0.984 0.373)
                rep["white_pixels"] += int((krgb.min(axis=-1) > 150).sum())                       # the robot's hulls
                diff = np.abs(data[e, ..., 0].astype(np.int64) - mm.astype(np.int64))
                rep["pixels"] += diff.size
                rep["mismatched_mm"] += int((diff != 0).sum())
                rep["off_by_more_than_1mm"] += int((diff > 1).sum())
                rep["max_mm_diff"] = max(rep["max_mm_diff"], int(diff.max()))
                rep["max_abs_depth_gl"] = max(rep["max_abs_depth_gl"], float(np.abs(raw[e] - dgl).max()))
                ext = np.linalg.inv(np.block([[cR @ np.diag([1.0, -1.0, -1.0]), cp[:, None]], [np.zeros((1, 3)), np.ones((1, 1))]]))
                rep["max_abs_extrinsics"] = max(rep["max_abs_extrinsics"], float(np.abs(frames.frames[name].camera.depth.extrinsics[e] - ext).max()))
                rep["background_pixels"] += int((dgl == 1.0).sum())
                if capsule_shapes:  # pixels whose ray enters a capsule first (the wrist camera's body, drawn as its collision capsule)
                    rep["capsule_pixels"] = rep.get("capsule_pixels", 0) + int(np.isin(RO.last_hit_shape, capsule_shapes).sum())
                if name == "bird_eye_cam":
                    rep["robot_pixels"] += int((mm < 1900).sum())  # nearer than the floor: robot and cube seen from above
                else:
                    rep["wrist_min_mm"] = min(rep.get("wrist_min_mm", 65535), int(mm.min()))
    rep["sample"] = data[0, ::8, ::8, 0]
    simu.close()
    return rep


def run_pick_task_parity(n_envs=16, n_steps=6, seed=0, episodes=2, async_control=True):
    """FR3SimplePickUpSimEnvCreator()(...) (30 Hz async control, relative TRPY actions, RandomCubePos, PickCubeSuccessWrapper)
    against the oracle's restatement of that wrapper stack on the same actions and the same cube placements."""
    from rcs_amd.envs import FR3SimplePickUpSimEnvCreator
    from rcs_amd.mjcf import compile_mjcf
    import rcs_oracle as O
    from rcs_env_oracle import OraclePickCubeEnv

    cm = compile_mjcf(PICKUP_SCENE)
    tcp = O.Pose(translation=[0.0, 0.0, 0.1034], rotation=np.array([[0.707, 0.707, 0], [-0.707, 0.707, 0], [0, 0, 1]]))
    if async_control:
        venv = FR3SimplePickUpSimEnvCreator()(n_envs=n_envs)
    else:  # SimTaskEnvCreator with the reference's default SimConfig: step_until_convergence
        from rcs_amd import common, sim
        from rcs_amd.envs import SimTaskEnvCreator, default_sim_robot_cfg

        rc = default_sim_robot_cfg(scene="fr3_simple_pick_up")
        rc.tcp_offset = common.Pose(translation=np.array([0.0, 0.0, 0.1034]), rotation=np.array([[0.707, 0.707, 0], [-0.707, 0.707, 0], [0, 0, 1]]))
        venv = SimTaskEnvCreator()(rc, sim_cfg=sim.SimConfig(), n_envs=n_envs)
    oenvs = [OraclePickCubeEnv(cm, tcp_offset=tcp, async_control=async_control) for _ in range(n_envs)]
    rng = np.random.default_rng(seed)
    rep = {"max_abs_obs": 0.0, "max_abs_box": 0.0, "max_abs_reward": 0.0, "flag_mismatches": 0, "min_reward": 9.0, "max_reward": -9.0,
           "grasped_seen": 0}
    for _ in range(episodes):
        np.random.seed(int(rng.integers(1 << 30)))
        box = venv.draw_box_qpos()
        obs, info = venv.reset(options={"box_qpos": box})
        for e, oe in enumerate(oenvs):
            oo, oi = oe.reset(box_qpos=box[e])
            rep["max_abs_obs"] = max(rep["max_abs_obs"], float(np.abs(obs["tquat"][e] - oo["tquat"]).max()), float(np.abs(obs["joints"][e] - oo["joints"]).max()))
            rep["flag_mismatches"] += int(bool(info["is_grasped"][e]) != bool(oi["is_grasped"]))
        kb = venv.sim.free_joint_qpos("box_joint")
        for e, oe in enumerate(oenvs):
            rep["max_abs_box"] = max(rep["max_abs_box"], float(np.abs(kb[e] - oe.sim.box_qpos).max()))
        for t in range(n_steps):
            # reach down towards the cube while closing / opening the gripper at random
            a = np.concatenate([rng.uniform(-0.05, 0.05, (n_envs, 2)), rng.uniform(-0.08, 0.01, (n_envs, 1)), rng.uniform(-0.1, 0.1, (n_envs, 3))], axis=1)
            g = rng.uniform(0, 1, n_envs).astype(np.float32)
            obs, reward, term, trunc, info = venv.step({"xyzrpy": a, "gripper": g})
            for e, oe in enumerate(oenvs):
                oo, orw, oterm, otrunc, oi = oe.step({"xyzrpy": a[e], "gripper": g[e]})
                rep["max_abs_obs"] = max(rep["max_abs_obs"], float(np.abs(obs["tquat"][e] - oo["tquat"]).max()), float(np.abs(obs["joints"][e] - oo["joints"]).max()))
                rep["max_abs_box"] = max(rep["max_abs_box"], float(np.abs(info["box_qpos"][e] - oe.sim.box_qpos).max()))
                rep["max_abs_reward"] = max(rep["max_abs_reward"], abs(float(reward[e]) - float(orw)))
                rep["flag_mismatches"] += int(bool(term[e]) != bool(oterm)) + int(bool(trunc[e]) != bool(otrunc)) + int(bool(info["success"][e]) != bool(oi["success"]))
                rep["flag_mismatches"] += int(bool(info["is_grasped"][e]) != bool(oi["is_grasped"])) + int(float(obs["gripper"][e]) != float(oo["gripper"]))
                rep["grasped_seen"] += int(bool(oi["is_grasped"]))
                if not async_control:
                    rep["flag_mismatches"] += int(int(info["substeps"][e]) != int(oe.sim.s.convergence_steps)) + int(bool(info["is_sim_converged"][e]) != bool(oi["is_sim_converged"]))
            rep["min_reward"], rep["max_reward"] = min(rep["min_reward"], float(reward.min())), max(rep["max_reward"], float(reward.max()))
    venv.close()
    return rep


def xarm7_frictionless_scene() -> str:
    """The xArm7 scene with frictionloss = 0, written to a temporary file: a 7-dof arm without gripper and WITHOUT dry friction
    rows -- the `Topo<7, false>` kernels without the friction variant, which no shipped scene selects."""
    import tempfile

    path = os.path.join(tempfile.gettempdir(), "rcs_amd_xarm7_frictionless", "scene.xml")
    if not os.path.exists(path):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        xml = open(XARM7_SCENE).read()
        assert 'frictionloss="1"' in xml
        open(path, "w").write(xml.replace('frictionloss="1"', 'frictionloss="0"'))
    return path


def fr3_arm_only_scene() -> str:
    """The FR3 scene without its hand (hand, camera body, fingers, gripper actuator, tendon and equality removed), written to a
    temporary file: 7 dofs and NO constraint row away from the joint limits, so one substep is the smooth system alone --
    what tests/golden/fr3_arm_dynamics.json (tools/derive_fr3_dynamics.py) predicts from first principles."""
    import tempfile
    import xml.etree.ElementTree as ET

    path = os.path.join(tempfile.gettempdir(), "rcs_amd_fr3_arm_only", "scene.xml")
    if not os.path.exists(path):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        tree = ET.parse(SCENE)
        root = tree.getroot()
        for parent in root.iter("body"):
            for child in list(parent):
                if child.tag == "body" and child.get("name") == "hand_0":
                    parent.remove(child)
        for tag in ("tendon", "equality"):
            for node in root.findall(tag):
                root.remove(node)
        for act in root.findall("actuator"):
            for a in list(act):
                if a.get("tendon") is not None:
                    act.remove(a)
        tree.write(path)
        for extra in ("collision_vertices.npz", "render_hulls.npz"):
            if os.path.exists(os.path.join(os.path.dirname(SCENE), extra)):
                import shutil

                shutil.copy(os.path.join(os.path.dirname(SCENE), extra), os.path.dirname(path))
    return path


def scene_with_joint_friction(robot: str) -> str:
    """The FR3 / arm6 scene with dry joint friction on every joint (frictionloss = 0.5), written to a temporary file: the
    friction-row kernels for the archetypes whose shipped scenes have none -- FR3 + hand (friction rows next to the coupling
    equality, the finger limits and the tendon actuator) and the 6-dof arm."""
    import tempfile

    src, old, new = {"fr3_fric": (SCENE, '<joint armature="0.1" damping="1"/>', '<joint armature="0.1" damping="1" frictionloss="0.5"/>'),
                     "arm6_fric": (ARM6_SCENE, 'range="-6.28319 6.28319" damping="2"/>', 'range="-6.28319 6.28319" damping="2" frictionloss="0.5"/>')}[robot]
    path = os.path.join(tempfile.gettempdir(), f"rcs_amd_{robot}", "scene.xml")
    if not os.path.exists(path):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        xml = open(src).read()
        assert old in xml
        open(path, "w").write(xml.replace(old, new))
        for extra in ("collision_vertices.npz",):
            if os.path.exists(os.path.join(os.path.dirname(src), extra)):
                import shutil

                shutil.copy(os.path.join(os.path.dirname(src), extra), os.path.dirname(path))
    return path


def make_vec_env(n_envs: int, async_control: bool, gripper: bool = True, relative: bool = True, control_mode=None, device: int = 0,
                 max_relative_movement=None, robot: str = "fr3", relative_to: str = "last_step", frequency: int = 30,
                 max_convergence_steps: int = 500, resolve_robot_contacts=None):
    """rcs_amd.envs.make_vec_env plus the test-only scene variants (`*_fric`, `xarm7_nofric`) and the kernel pin."""
    from rcs_amd import envs

    rcfg = None
    if robot in ("fr3_fric", "arm6_fric"):
        rcfg = envs.default_sim_robot_cfg("fr3_empty_world") if robot == "fr3_fric" else envs.arm6_sim_robot_cfg()
        rcfg.mjcf_scene_path = rcfg.kinematic_model_path = scene_with_joint_friction(robot)
    if robot == "xarm7_nofric":
        rcfg = envs.xarm7_sim_robot_cfg()
        rcfg.mjcf_scene_path = rcfg.kinematic_model_path = xarm7_frictionless_scene()
    venv = envs.make_vec_env(n_envs, async_control, gripper=gripper, relative=relative, control_mode=control_mode, device=device,
                             max_relative_movement=max_relative_movement, robot=robot if rcfg is None else robot.split("_")[0],
                             relative_to=relative_to, frequency=frequency, max_convergence_steps=max_convergence_steps, robot_cfg=rcfg,
                             resolve_robot_contacts=resolve_robot_contacts)
    if KERNEL != "auto":  # "auto" leaves the handle's default (batch-size rule, or the RCSH_KERNEL environment variable)
        venv.sim.set_kernel(KERNEL)
    return venv


def make_oracle_envs(n_envs: int, async_control: bool, gripper: bool = True, relative: bool = True, mode: str = "joints",
                     max_relative_movement=None, robot: str = "fr3", relative_to: str = "last_step", frequency: int = 30,
                     max_convergence_steps: int = 500):
    from rcs_amd.mjcf import compile_mjcf
    from rcs_env_oracle import ARM6, SO101, UR5E, XARM7, OracleEnv

    cm = compile_mjcf(xarm7_frictionless_scene() if robot == "xarm7_nofric" else scene_with_joint_friction(robot) if robot.endswith("_fric")
                      else {"xarm7": XARM7_SCENE, "arm6": ARM6_SCENE, "ur5e": UR5E_SCENE, "so101": SO101_SCENE}.get(robot, SCENE))
    if relative and max_relative_movement is None:
        max_relative_movement = MAX_JOINT_MOV
    return [OracleEnv(cm, control_mode=mode, gripper=gripper and robot in GRIPPER_ROBOTS, max_relative_movement=max_relative_movement if relative else None,
                      async_control=async_control, robot={"xarm7": XARM7, "xarm7_nofric": XARM7, "arm6": ARM6, "arm6_fric": ARM6, "ur5e": UR5E, "so101": SO101}.get(robot), relative_to=relative_to,
                      frequency=frequency, max_convergence_steps=max_convergence_steps) for _ in range(n_envs)]


def run_joint_rollout_parity(n_envs: int = 64, n_steps: int = 3, async_control: bool = True, seed: int = 0, gripper: bool = True,
                             episodes: int = 1, robot: str = "fr3", relative_to: str = "last_step", frequency: int = 30,
                             max_convergence_steps: int = 500):
    """Fused HIP env-step vs the oracle on the same seeded actions; returns max abs differences + flag mismatches."""
    gripper = gripper and robot in GRIPPER_ROBOTS
    venv = make_vec_env(n_envs, async_control, gripper=gripper, robot=robot, relative_to=relative_to, frequency=frequency,
                        max_convergence_steps=max_convergence_steps)
    oenvs = make_oracle_envs(n_envs, async_control, gripper=gripper, robot=robot, relative_to=relative_to, frequency=frequency,
                             max_convergence_steps=max_convergence_steps)
    dof = robot_dof(robot)
    joints, grip = synthetic_actions(n_envs, n_steps * episodes, seed, dof=dof)
    rep = {"max_abs_obs": 0.0, "max_abs_qpos": 0.0, "max_abs_qvel": 0.0, "max_abs_finger": 0.0, "max_abs_gripper_width": 0.0,
           "flag_mismatches": 0, "substep_mismatches": 0, "steps": 0}

    def compare(obs, info, oracle_results, substeps=None):
        for e, (oo, oi) in enumerate(oracle_results):
            for k in ("tquat", "joints"):
                rep["max_abs_obs"] = max(rep["max_abs_obs"], float(np.abs(obs[k][e] - oo[k]).max()))
            # xyzrpy: the reference's Euler extraction (yaw in [0, pi], roll near +-pi at a downward-pointing TCP)
            # is discontinuous exactly where the arm lives, so 1e-16 of noise flips the triple; compare the rotation
            rep["max_abs_obs"] = max(rep["max_abs_obs"], float(np.abs(obs["xyzrpy"][e][:3] - oo["xyzrpy"][:3]).max()),
                                     rpy_error(obs["xyzrpy"][e][3:], oo["xyzrpy"][3:]))
            if gripper:
                rep["flag_mismatches"] += int(float(obs["gripper"][e]) != float(oo["gripper"]))
                rep["max_abs_gripper_width"] = max(rep["max_abs_gripper_width"], abs(float(info["gripper_width"][e]) - oi["gripper_width"]))
            for k in ("collision", "ik_success", "is_sim_converged", "is_grasped"):
                if k in oi and k in info:
                    rep["flag_mismatches"] += int(bool(info[k][e]) != bool(oi[k]))
        q, v = venv.sim.qpos, venv.sim.qvel
        for e, oe in enumerate(oenvs):
            # arm joints and finger slides are reported separately: the fingers rest exactly on a joint limit with zero
            # actuator force (the reference model's gripper equilibrium IS the limit), where the sign of round-off decides
            # whether a limit row exists in a substep, so their trajectories are only reproducible to ~1e-5 m
            rep["max_abs_qpos"] = max(rep["max_abs_qpos"], float(np.abs(q[e][:dof] - oe.sim.qpos[:dof]).max()))
            rep["max_abs_qvel"] = max(rep["max_abs_qvel"], float(np.abs(v[e][:dof] - oe.sim.qvel[:dof]).max()))
            if q.shape[1] > dof:
                rep["max_abs_finger"] = max(rep["max_abs_finger"], float(np.abs(q[e][dof:] - oe.sim.qpos[dof : q.shape[1]]).max()))
            if substeps is not None and not async_control:
                rep["substep_mismatches"] += int(int(substeps[e]) != int(oe.sim.s.convergence_steps))

    t = 0
    for _ in range(episodes):
        obs, info = venv.reset()
        compare(obs, info, [oe.reset() for oe in oenvs])
        for _ in range(n_steps):
            act = {"joints": joints[t]}
            if gripper:
                act["gripper"] = grip[t]
            obs, _, _, trunc, info = venv.step(act)
            ores = []
            for e, oe in enumerate(oenvs):
                a = {"joints": joints[t, e]}
                if gripper:
                    a["gripper"] = grip[t, e]
                oo, _, _, otr, oi = oe.step(a)
                rep["flag_mismatches"] += int(bool(trunc[e]) != bool(otr))
                ores.append((oo, oi))
            compare(obs, info, ores, info["substeps"])
            t += 1
            rep["steps"] += 1
    venv.close()
    return rep


def cartesian_actions(n_envs: int, n_steps: int, seed: int = 0):
    """xyz ~ U(+-0.05)^3 m, rpy ~ U(+-0.1)^3 rad, gripper toggled every 10 steps."""
    act = np.zeros((n_steps, n_envs, 6))
    grip = np.zeros((n_steps, n_envs), dtype=np.float32)
    for e in range(n_envs):
        rng = np.random.default_rng(seed + e)
        act[:, e, :3] = rng.uniform(-0.05, 0.05, size=(n_steps, 3))
        act[:, e, 3:] = rng.uniform(-0.1, 0.1, size=(n_steps, 3))
        grip[:, e] = ((np.arange(n_steps) // 10) % 2).astype(np.float32)
    return act, grip


def run_cartesian_rollout_parity(n_envs=32, n_steps=4, async_control=True, seed=0, mode="xyzrpy", relative=True, gripper=True,
                                 relative_to: str = "last_step", robot: str = "fr3"):
    """Cartesian control (relative TRPY / TQuat actions -> CLIK -> joint targets): HIP path vs the oracle."""
    from rcs_amd.envs import ControlMode

    cm = ControlMode.CARTESIAN_TRPY if mode == "xyzrpy" else ControlMode.CARTESIAN_TQuat
    mm = (0.2, float(np.deg2rad(45)))
    gripper = gripper and robot in ("fr3", "so101")
    venv = make_vec_env(n_envs, async_control, gripper=gripper, relative=relative, control_mode=cm, max_relative_movement=mm,
                        relative_to=relative_to, robot=robot)
    oenvs = make_oracle_envs(n_envs, async_control, gripper=gripper, relative=relative, mode=mode, max_relative_movement=mm,
                             relative_to=relative_to, robot=robot)
    act6, grip = cartesian_actions(n_envs, n_steps, seed)
    rep = {"max_abs_qpos": 0.0, "max_abs_tquat": 0.0, "max_abs_target": 0.0, "flag_mismatches": 0, "steps": 0, "ik_fail": 0}
    obs, info = venv.reset()
    ores = [oe.reset() for oe in oenvs]
    for t in range(n_steps):
        if mode == "xyzrpy":
            a_all = act6[t]
        else:  # relative tquat action: translation + quaternion of the small rpy rotation
            from rcs_amd.common import Pose

            a_all = np.stack([np.concatenate([act6[t, e, :3], Pose(rpy_vector=act6[t, e, 3:]).rotation_q()]) for e in range(n_envs)])
        if not relative:  # absolute target = current pose shifted
            base = obs["xyzrpy"] if mode == "xyzrpy" else obs["tquat"]
            a_all = base.copy()
            a_all[:, :3] += act6[t, :, :3]
        action = {mode: a_all}
        if gripper:
            action["gripper"] = grip[t]
        obs, _, _, trunc, info = venv.step(action)
        st = venv.robot.get_state()
        q = venv.sim.qpos
        for e, oe in enumerate(oenvs):
            a = {mode: a_all[e]}
            if gripper:
                a["gripper"] = grip[t, e]
            oo, _, _, otr, oi = oe.step(a)
            rep["flag_mismatches"] += int(bool(trunc[e]) != bool(otr))
            for k in ("collision", "ik_success", "is_sim_converged"):
                rep["flag_mismatches"] += int(bool(info[k][e]) != bool(oi[k]))
            rep["ik_fail"] += int(not oi["ik_success"])
            rep["max_abs_qpos"] = max(rep["max_abs_qpos"], float(np.abs(q[e][:venv.dof] - oe.sim.qpos[:venv.dof]).max()))
            rep["max_abs_tquat"] = max(rep["max_abs_tquat"], float(np.abs(obs["tquat"][e] - oo["tquat"]).max()))
            rep["max_abs_xyzrpy"] = max(rep.get("max_abs_xyzrpy", 0.0), float(np.abs(obs["xyzrpy"][e][:3] - oo["xyzrpy"][:3]).max()),
                                        rpy_error(obs["xyzrpy"][e][3:], oo["xyzrpy"][3:]))
            near = any(min(abs(abs(x) - s_) for s_ in (0.0, np.pi)) < 1e-6 for x in oo["xyzrpy"][3:])
            rep["rpy_componentwise"] = rep.get("rpy_componentwise", 0) + int(not near)
            rep["max_abs_target"] = max(rep["max_abs_target"], float(np.abs(st.target_angles[e] - np.array(oe.sim.s.target_angles[:venv.dof])).max()))
        rep["steps"] += 1
    venv.close()
    return rep


def run_physics_parity_mid_stroke(n_envs=64, n_calls=6, k=17, seed=0):
    """Sim.step(k) through the fine-grained API with the fingers held mid-stroke (no limit row ever switches):
    every joint must then agree with the oracle to round-off."""
    from rcs_amd import sim as S
    from rcs_amd.envs import default_sim_gripper_cfg, default_sim_robot_cfg
    from rcs_amd.mjcf import compile_mjcf
    import rcs_oracle as O

    cfg = default_sim_robot_cfg("fr3_empty_world")
    simu = S.Sim(cfg.mjcf_scene_path, S.SimConfig(), n_envs=n_envs)
    if KERNEL != "auto":
        simu.set_kernel(KERNEL)
    robot = S.SimRobot(simu, None, cfg)
    grip = S.SimGripper(simu, default_sim_gripper_cfg())
    cm = compile_mjcf(SCENE)
    arm = [f"fr3_joint{i}_0" for i in range(1, 8)]
    from rcs_env_oracle import FR3_Q_HOME

    osims = [O.Sim(cm, arm, arm, "attachment_site_0", "base_0", FR3_Q_HOME, None, "finger_joint1_0", "actuator8_0") for _ in range(n_envs)]
    rng = np.random.default_rng(seed)
    q0 = np.tile(np.concatenate([FR3_Q_HOME, [0.02, 0.02]]), (n_envs, 1))
    q0[:, :7] += rng.uniform(-0.2, 0.2, size=(n_envs, 7))
    simu.set_qpos(q0)
    grip.set_normalized_width(np.full(n_envs, 0.5))
    for e, o in enumerate(osims):
        for i in range(9):
            o.s.d.qpos[i] = q0[e, i]
        o.gripper_set_normalized_width(0.5)
    rep = {"max_abs_qpos": 0.0, "max_abs_qvel": 0.0, "max_abs_cart": 0.0, "flag_mismatches": 0}
    for _ in range(n_calls):
        tgt = q0[:, :7] + rng.uniform(-0.1, 0.1, size=(n_envs, 7))
        robot.set_joint_position(tgt)
        simu.step(k)
        q, v, cart, st = simu.qpos, simu.qvel, robot.get_cartesian_position(), robot.get_state()
        for e, o in enumerate(osims):
            o.set_joint_position(tgt[e])
            o.step(k)
            rep["max_abs_qpos"] = max(rep["max_abs_qpos"], float(np.abs(q[e] - o.qpos).max()))
            rep["max_abs_qvel"] = max(rep["max_abs_qvel"], float(np.abs(v[e] - o.qvel).max()))
            p = o.get_cartesian_position()
            rep["max_abs_cart"] = max(rep["max_abs_cart"], float(np.abs(cart[e] - np.concatenate([p.translation(), p.rotation_q()])).max()))
            rep["flag_mismatches"] += int(bool(st.is_moving[e]) != bool(o.s.is_moving)) + int(bool(st.is_arrived[e]) != bool(o.s.is_arrived))
    simu.close()
    return rep


def run_physics_parity_at_joint_limits(n_envs=32, n_calls=8, k=17, seed=0, n_over=6):
    """Sim.step(k) with `n_over` arm joints started BEYOND their range (penetrating limit rows, all active at once) and
    commanded to the range end: exercises the many-row path of the constraint solve (team kernel: more than three limit
    rows -> Newton with line search instead of the active-set vote).  Fingers mid-stroke, as in the mid-stroke test."""
    from rcs_amd import sim as S
    from rcs_amd.envs import default_sim_gripper_cfg, default_sim_robot_cfg
    from rcs_amd.mjcf import compile_mjcf
    import rcs_oracle as O
    from rcs_env_oracle import FR3_Q_HOME

    cfg = default_sim_robot_cfg("fr3_empty_world")
    # (contacts detected only, on both sides: with two joints at the end of their ranges the teleported arm starts 20 cm BELOW the floor
    # -- this test is about the limit rows)
    simu = S.Sim(cfg.mjcf_scene_path, S.SimConfig(), n_envs=n_envs, resolve_robot_contacts=False)
    if KERNEL != "auto":
        simu.set_kernel(KERNEL)
    robot = S.SimRobot(simu, None, cfg)
    grip = S.SimGripper(simu, default_sim_gripper_cfg())
    cm = compile_mjcf(SCENE)
    arm = [f"fr3_joint{i}_0" for i in range(1, 8)]
    osims = [O.Sim(cm, arm, arm, "attachment_site_0", "base_0", FR3_Q_HOME, None, "finger_joint1_0", "actuator8_0", resolve_contacts=False) for _ in range(n_envs)]
    rng = np.random.default_rng(seed)
    hi = np.asarray(cm.jnt_range)[:7, 1]
    q0 = np.tile(np.concatenate([FR3_Q_HOME, [0.02, 0.02]]), (n_envs, 1))
    q0[:, :n_over] = hi[:n_over] + rng.uniform(0.002, 0.02, size=(n_envs, n_over))
    simu.set_qpos(q0)
    grip.set_normalized_width(np.full(n_envs, 0.5))
    for e, o in enumerate(osims):
        for i in range(9):
            o.s.d.qpos[i] = q0[e, i]
        o.gripper_set_normalized_width(0.5)
    rep = {"max_abs_qpos": 0.0, "max_abs_qvel": 0.0, "max_rows": 0}
    tgt = np.tile(np.concatenate([hi[:n_over], FR3_Q_HOME[n_over:]]), (n_envs, 1))
    for _ in range(n_calls):
        robot.set_joint_position(tgt)
        simu.step(k)
        q, v = simu.qpos, simu.qvel
        for e, o in enumerate(osims):
            rep["max_rows"] = max(rep["max_rows"], int((np.asarray(o.qpos)[:7] > hi).sum()))
            o.set_joint_position(tgt[e])
            o.step(k)
            rep["max_abs_qpos"] = max(rep["max_abs_qpos"], float(np.abs(q[e] - o.qpos).max()))
            rep["max_abs_qvel"] = max(rep["max_abs_qvel"], float(np.abs(v[e] - o.qvel).max()))
    simu.close()
    return rep


def run_xarm7_at_joint_limits_parity(n_envs=48, n_calls=8, k=17, seed=0, n_over=3):
    """xArm7 (a dry-friction row on every joint) with `n_over` of its bounded joints started BEYOND their range and commanded
    to the range end, the others to random targets: limit rows and friction rows change zones in the same solves -- the
    candidate rounds of the factorisation slot carry limit rows in their zone sets, and the serial routine's team-wide line
    search takes the limit crossings in its second pass.  Kernel vs oracle through the fine-grained API."""
    from rcs_amd import sim as S
    from rcs_amd.envs import xarm7_sim_robot_cfg
    from rcs_amd.mjcf import compile_mjcf
    import rcs_oracle as O
    from rcs_env_oracle import XARM7

    cfg = xarm7_sim_robot_cfg()
    simu = S.Sim(cfg.mjcf_scene_path, S.SimConfig(), n_envs=n_envs)
    if KERNEL != "auto":
        simu.set_kernel(KERNEL)
    robot = S.SimRobot(simu, None, cfg)
    cm = compile_mjcf(XARM7_SCENE)
    osims = [O.Sim(cm, XARM7["joints"], XARM7["actuators"], XARM7["site"], XARM7["base"], XARM7["q_home"], None, arm_collision_geoms=[]) for _ in range(n_envs)]
    rng = np.random.default_rng(seed)
    hi = np.asarray(cm.jnt_range)[:7, 1]
    bounded = [1, 3, 5][:n_over]  # (the xArm7's joints 2, 4, 6 have ranges narrower than a turn)
    q0 = np.tile(np.asarray(XARM7["q_home"]), (n_envs, 1)) + rng.uniform(-0.1, 0.1, (n_envs, 7))
    q0[:, bounded] = hi[bounded] + rng.uniform(0.002, 0.03, size=(n_envs, len(bounded)))
    simu.reset(); robot.reset()
    simu.set_qpos(q0)
    for e, o in enumerate(osims):
        o.reset(); o.robot_reset()
        for i in range(7):
            o.s.d.qpos[i] = q0[e, i]
    rep = {"max_abs_qpos": 0.0, "max_abs_qvel": 0.0, "max_rows": 0}
    for _ in range(n_calls):
        tgt = np.asarray(XARM7["q_home"]) + rng.uniform(-0.3, 0.3, (n_envs, 7))
        tgt[:, bounded] = hi[bounded]
        robot.set_joint_position(tgt)
        simu.step(k)
        q, v = simu.qpos, simu.qvel
        for e, o in enumerate(osims):
            rep["max_rows"] = max(rep["max_rows"], int((np.asarray(o.qpos)[:7] > hi - 1e-3).sum()))
            o.set_joint_position(tgt[e])
            o.step(k)
            rep["max_abs_qpos"] = max(rep["max_abs_qpos"], float(np.abs(q[e] - np.asarray(o.qpos)[:7]).max()))
            rep["max_abs_qvel"] = max(rep["max_abs_qvel"], float(np.abs(v[e] - np.asarray(o.qvel)[:7]).max()))
    simu.close()
    return rep


def _pinch_placements(n_envs: int, seed: int):
    """Cube poses a few millimetres / degrees off the gripper's closing axis (the scene's own pose first)."""
    rng = np.random.default_rng(seed)
    qb = np.tile(np.array([0.44, 0.1, 0.0288, 0, 0, 0, 1.0]), (n_envs, 1))
    qb[1:, 0] += rng.uniform(-0.004, 0.004, n_envs - 1)
    qb[1:, 1] += rng.uniform(-0.004, 0.004, n_envs - 1)
    yaw = np.zeros(n_envs)
    yaw[1:] = rng.uniform(-0.1, 0.1, n_envs - 1)
    qb[:, 3], qb[:, 6] = np.cos((np.pi + yaw) / 2), np.sin((np.pi + yaw) / 2)
    return qb


def run_grasp_parity(n_envs=4, seed=0, swing_up=True):
    """A scripted pinch through the 1:1 Sim / SimRobot / SimGripper API, kernel vs oracle: move over the cube, descend,
    close the fingers on it (finger pads against the cube: box-box contacts, friction 2), lift, swing the arm up until
    the cube is above PickCubeSuccessWrapper's success height (1.002 m), release.  Every stage compares joint and cube
    state and the collision flags; returns the worst differences and what happened to the cube."""
    import dataclasses

    from rcs_amd import common
    from rcs_amd import sim as S
    from rcs_amd.envs import default_sim_gripper_cfg, default_sim_robot_cfg
    from rcs_amd.mjcf import compile_mjcf
    import rcs_oracle as O
    from rcs_env_oracle import FR3_Q_HOME

    cfg = dataclasses.replace(default_sim_robot_cfg("fr3_simple_pick_up"), tcp_offset=common.Pose(common.FrankaHandTCPOffset()))
    simu = S.Sim(cfg.mjcf_scene_path, S.SimConfig(), n_envs=n_envs)
    if KERNEL != "auto":
        simu.set_kernel(KERNEL)
    robot = S.SimRobot(simu, None, cfg)
    grip = S.SimGripper(simu, default_sim_gripper_cfg())
    cm = compile_mjcf(PICKUP_SCENE)
    arm = [f"fr3_joint{i}_0" for i in range(1, 8)]
    osims = [O.Sim(cm, arm, arm, "attachment_site_0", "base_0", FR3_Q_HOME, O.franka_hand_tcp_offset(), "finger_joint1_0", "actuator8_0") for _ in range(n_envs)]
    assert osims[0].model.resolve_contacts == 1
    qb = _pinch_placements(n_envs, seed)
    simu.reset(); robot.reset(); grip.reset()
    for o in osims:
        o.reset(); o.robot_reset(); o.gripper_reset()
    simu.set_free_joint_qpos("box_joint", qb)
    for e, o in enumerate(osims):
        o.box_qpos = qb[e]
    rep = {"max_abs_qpos": 0.0, "max_abs_qvel": 0.0, "max_abs_box": 0.0, "max_abs_box_vel": 0.0, "flag_mismatches": 0, "max_ncon": 0,
           "coupled_substeps": 0, "max_newton": 0, "max_noslip": 0, "stages": {}}

    def advance(tag, k, conv=False):
        if conv:
            simu.step_until_convergence()
            [o.step_until_convergence() for o in osims]
        else:
            simu.step(k)
        q, v, bq, bv, st, gs = simu.qpos, simu.qvel, simu.free_joint_qpos("box_joint"), simu.free_joint_qvel("box_joint"), robot.get_state(), grip.get_state()
        for e, o in enumerate(osims):
            if not conv:
                for _ in range(k):
                    o.step(1)
                    d = o.s.d
                    rep["max_ncon"] = max(rep["max_ncon"], int(d.ncon))
                    if d.coupled:
                        rep["coupled_substeps"] += 1
                        rep["max_newton"], rep["max_noslip"] = max(rep["max_newton"], int(d.solver_niter)), max(rep["max_noslip"], int(d.noslip_niter))
            rep["max_abs_qpos"] = max(rep["max_abs_qpos"], float(np.abs(q[e] - np.asarray(o.qpos)).max()))
            rep["max_abs_qvel"] = max(rep["max_abs_qvel"], float(np.abs(v[e] - np.asarray(o.qvel)).max()))
            rep["max_abs_box"] = max(rep["max_abs_box"], float(np.abs(bq[e] - o.box_qpos).max()))
            rep["max_abs_box_vel"] = max(rep["max_abs_box_vel"], float(np.abs(bv[e] - o.box_qvel).max()))
            rep["flag_mismatches"] += int(bool(st.collision[e]) != bool(o.s.robot_collision)) + int(bool(gs.collision[e]) != bool(o.s.grp_collision))
            rep["flag_mismatches"] += int(bool(grip.is_grasped()[e]) != o.gripper_is_grasped())
            if conv:
                rep["flag_mismatches"] += int(int(simu.convergence_steps()[e]) != int(o.s.convergence_steps))
        rep["stages"][tag] = {"box_z": bq[:, 2].copy(), "width": grip.get_normalized_width().copy()}

    simu.step(1); [o.step(1) for o in osims]
    home = osims[0].get_cartesian_position()  # (the site frame is that of the last position stage: valid after a step)

    def move(xyz):
        robot.set_cartesian_position(np.tile(np.concatenate([xyz, home.rotation_q()]), (n_envs, 1)))
        for o in osims:
            o.set_cartesian_position(O.Pose(translation=np.array(xyz), quaternion=home.rotation_q()))

    grip.open(); [o.gripper_open() for o in osims]
    move([0.44, 0.1, 0.20]); advance("above", 400)
    move([0.44, 0.1, 0.035]); advance("down", 600)
    grip.shut(); [o.gripper_grasp() for o in osims]
    advance("closed", 200)
    move([0.44, 0.1, 0.30]); advance("lifted", 500)
    if swing_up:
        qg = robot.get_joint_position()
        qup = np.array([0, 0, 0, -0.2, 0, 2.0, 0.785])
        for k in range(1, 9):
            tgt = qg + (qup - qg) * k / 8
            robot.set_joint_position(tgt)
            for e, o in enumerate(osims):
                o.set_joint_position(tgt[e])
            advance(f"swing{k}", 150)
        advance("held", 200)
        advance("converged", 0, conv=True)
    grip.open(); [o.gripper_open() for o in osims]
    advance("released", 300)
    simu.close()
    return rep


XARM7_PICK_SCENE = os.path.join(ROOT, "robot-control-stack_amd", "rcs_amd", "scenes", "xarm7_pick_world", "scene.xml")


def run_xarm7_links_on_the_floor_parity(n_envs=3, seed=0, substeps=900, width=48, height=36):
    """scenes/xarm7_pick_world: the arm's OWN link hulls (the reference's link<i>_convex collision meshes, xarm7.xml:103-156)
    against the floor.  The shoulder joint is sent far enough forward that the forearm links come down on the floor plane
    (hull-plane contacts on arm links, resolved in the coupled solve together with the arm's dry-friction rows), then the
    fixed camera's frame is compared with the numpy ray caster on the oracle's frames -- the arm is in the picture."""
    from rcs_amd import render
    from rcs_amd import sim as S
    from rcs_amd.camera import SimCameraConfig, SimCameraSet
    from rcs_amd.envs import xarm7_pick_sim_gripper_cfg, xarm7_pick_sim_robot_cfg
    from rcs_amd.mjcf import compile_mjcf
    import rcs_oracle as O
    import rcs_render_oracle as RO
    from rcs_env_oracle import XARM7_PICK

    cfg = xarm7_pick_sim_robot_cfg()
    simu = S.Sim(cfg.mjcf_scene_path, S.SimConfig(), n_envs=n_envs)
    if KERNEL != "auto":
        simu.set_kernel(KERNEL)
    robot = S.SimRobot(simu, None, cfg)
    S.SimGripper(simu, xarm7_pick_sim_gripper_cfg())
    cs = SimCameraSet(simu, {"side": SimCameraConfig(identifier="side_cam", frame_rate=0, resolution_width=width, resolution_height=height)},
                      physical_units=True, render_on_demand=True)
    cs.set_double_precision(True)  # (this comparison is about WHAT is drawn and when: exact pixels against the float64 restatement)
    cm = compile_mjcf(XARM7_PICK_SCENE)
    R = XARM7_PICK
    osims = [O.Sim(cm, R["joints"], R["actuators"], R["site"], R["base"], R["q_home"], O.Pose(translation=np.array([0.0, 0.0, 0.1034])),
                   R["gripper_joint"], R["gripper_actuator"], arm_collision_geoms=[], gripper_cfg=R["gripper_cfg"]) for _ in range(n_envs)]
    assert osims[0].model.resolve_contacts == 1 and simu.resolve_robot_contacts
    rng = np.random.default_rng(seed)
    simu.reset(); robot.reset()
    for o in osims:
        o.reset(); o.robot_reset()
    # shoulder forward and down (joint2 range: -2.059 .. 2.0944), elbow bent so that the forearm points at the floor, the wrist folded
    # back so that the hand stays clear: links 5 and 6 are what comes down; turned away from the cube
    tgt = np.tile(np.array([0.0, 2.05, 0.0, 1.2, 0.0, 3.0, 0.0]), (n_envs, 1))
    tgt[:, 0] = rng.uniform(0.9, 1.3, n_envs)
    tgt[:, 3] += rng.uniform(-0.1, 0.1, n_envs)
    tgt[:, 5] += rng.uniform(-0.1, 0.1, n_envs)
    robot.set_joint_position(tgt)
    for e, o in enumerate(osims):
        o.set_joint_position(tgt[e])
    rep = {"max_abs_qpos": 0.0, "max_abs_qvel": 0.0, "max_abs_box": 0.0, "max_ncon": 0, "coupled_substeps": 0, "arm_link_contacts": 0,
           "max_links_in_contact": 0, "pixels": 0, "mismatched_mm": 0, "arm_pixels": 0}
    gb = np.asarray(cm.arrays["geom_bodyid"])
    hand_body = cm.name2id("body", "hand")
    done = 0
    while done < substeps:
        k = min(150, substeps - done)
        simu.step(k)
        done += k
        q, v, bq = simu.qpos, simu.qvel, simu.free_joint_qpos("box_joint")
        for e, o in enumerate(osims):
            for _ in range(k):
                o.step(1)
                d = o.s.d
                rep["max_ncon"] = max(rep["max_ncon"], int(d.ncon))
                rep["coupled_substeps"] += int(bool(d.coupled))
                # contacts whose robot geom rides on an arm link (a body before the hand in the tree)
                bodies = {int(gb[g]) for c in range(int(d.ncon)) for g in d.contact_geom[c] if g < cm.ngeom} - {0}  # (ids past the model's: the free cube)
                rep["arm_link_contacts"] += sum(1 for b in bodies if b < hand_body)
                rep["max_links_in_contact"] = max(rep["max_links_in_contact"], len(bodies))
            rep["max_abs_qpos"] = max(rep["max_abs_qpos"], float(np.abs(q[e] - np.asarray(o.qpos)).max()))
            rep["max_abs_qvel"] = max(rep["max_abs_qvel"], float(np.abs(v[e] - np.asarray(o.qvel)).max()))
            rep["max_abs_box"] = max(rep["max_abs_box"], float(np.abs(bq[e] - o.box_qpos).max()))
    rep["qpos"] = simu.qpos[:, :7].copy()
    rep["target"] = tgt
    frames = cs.get_latest_frames()
    data = frames.frames["side"].camera.depth.data
    link, pos, rot, fovy = render.camera_in_link(cm, "side_cam")
    for e, o in enumerate(osims):
        dgl, mm, cR, cp, orgb = RO.render_depth(cs._scene, (link, pos, rot, fovy, width, height), RO.oracle_frames(o, cm), colour=True)
        diff = np.abs(data[e, ..., 0].astype(np.int64) - mm.astype(np.int64))
        rep["pixels"] += diff.size
        rep["mismatched_mm"] += int((diff != 0).sum())
        rep["arm_pixels"] += int((orgb.min(axis=-1) > 150).sum())  # the white hulls
    simu.close()
    return rep


def run_xarm7_pick_parity(n_envs=4, seed=0, stages=("above", "down", "closed", "lifted", "held", "released")):
    """BASELINE configs[3] as written: the xArm7 (dry joint friction on every arm joint) with a two-finger gripper picks the
    cube up -- move over it, descend, close the fingers (pads against the cube: box-box contacts; the friction-dof rows are
    rows of the same coupled problem), lift 20 cm, hold, release.  Kernel vs oracle through the 1:1 Sim / SimRobot / SimGripper
    API, cube placements a few millimetres / degrees apart per environment."""
    from rcs_amd import sim as S
    from rcs_amd.envs import xarm7_pick_sim_gripper_cfg, xarm7_pick_sim_robot_cfg
    from rcs_amd.mjcf import compile_mjcf
    import rcs_oracle as O
    from rcs_env_oracle import XARM7_PICK

    cfg = xarm7_pick_sim_robot_cfg()
    simu = S.Sim(cfg.mjcf_scene_path, S.SimConfig(), n_envs=n_envs)
    if KERNEL != "auto":
        simu.set_kernel(KERNEL)
    robot = S.SimRobot(simu, None, cfg)
    grip = S.SimGripper(simu, xarm7_pick_sim_gripper_cfg())
    cm = compile_mjcf(XARM7_PICK_SCENE)
    R = XARM7_PICK
    tcp = O.Pose(translation=np.array([0.0, 0.0, 0.1034]))
    osims = [O.Sim(cm, R["joints"], R["actuators"], R["site"], R["base"], R["q_home"], tcp, R["gripper_joint"], R["gripper_actuator"],
                   arm_collision_geoms=[], gripper_cfg=R["gripper_cfg"]) for _ in range(n_envs)]
    assert osims[0].model.resolve_contacts == 1 and simu.resolve_robot_contacts
    rng = np.random.default_rng(seed)
    qb = np.tile(np.array([0.40, 0.0, 0.0288, 0, 0, 0, 1.0]), (n_envs, 1))
    qb[1:, 0] += rng.uniform(-0.004, 0.004, n_envs - 1)
    qb[1:, 1] += rng.uniform(-0.004, 0.004, n_envs - 1)
    yaw = np.zeros(n_envs)
    yaw[1:] = rng.uniform(-0.1, 0.1, n_envs - 1)
    qb[:, 3], qb[:, 6] = np.cos((np.pi + yaw) / 2), np.sin((np.pi + yaw) / 2)
    simu.reset(); robot.reset(); grip.reset()
    for o in osims:
        o.reset(); o.robot_reset(); o.gripper_reset()
    simu.set_free_joint_qpos("box_joint", qb)
    for e, o in enumerate(osims):
        o.box_qpos = qb[e]
    rep = {"max_abs_qpos": 0.0, "max_abs_qvel": 0.0, "max_abs_box": 0.0, "max_abs_box_vel": 0.0, "flag_mismatches": 0, "max_ncon": 0,
           "coupled_substeps": 0, "max_newton": 0, "ik_failures": 0, "stages": {}}

    def advance(tag, k):
        simu.step(k)
        q, v, bq, bv, st, gs = simu.qpos, simu.qvel, simu.free_joint_qpos("box_joint"), simu.free_joint_qvel("box_joint"), robot.get_state(), grip.get_state()
        for e, o in enumerate(osims):
            for _ in range(k):
                o.step(1)
                d = o.s.d
                rep["max_ncon"] = max(rep["max_ncon"], int(d.ncon))
                if d.coupled:
                    rep["coupled_substeps"] += 1
                    rep["max_newton"] = max(rep["max_newton"], int(d.solver_niter))
            rep["max_abs_qpos"] = max(rep["max_abs_qpos"], float(np.abs(q[e] - np.asarray(o.qpos)).max()))
            rep["max_abs_qvel"] = max(rep["max_abs_qvel"], float(np.abs(v[e] - np.asarray(o.qvel)).max()))
            rep["max_abs_box"] = max(rep["max_abs_box"], float(np.abs(bq[e] - o.box_qpos).max()))
            rep["max_abs_box_vel"] = max(rep["max_abs_box_vel"], float(np.abs(bv[e] - o.box_qvel).max()))
            rep["flag_mismatches"] += int(bool(st.collision[e]) != bool(o.s.robot_collision)) + int(bool(gs.collision[e]) != bool(o.s.grp_collision))
            rep["flag_mismatches"] += int(bool(grip.is_grasped()[e]) != o.gripper_is_grasped()) + int(bool(st.ik_success[e]) != bool(o.s.ik_success))
            rep["ik_failures"] += int(not o.s.ik_success)
        rep["stages"][tag] = {"box_z": bq[:, 2].copy(), "width": grip.get_normalized_width().copy()}

    simu.step(1); [o.step(1) for o in osims]
    down = O.Pose(rotation=np.diag([1.0, -1.0, -1.0])).rotation_q()  # tool axis pointing at the floor, fingers closing along y
    base_z = 0.12  # the robot frame: the xArm7's base body sits 0.12 m above the floor

    def move(xyz):
        t = np.array([xyz[0], xyz[1], xyz[2] - base_z])
        robot.set_cartesian_position(np.tile(np.concatenate([t, down]), (n_envs, 1)))
        for o in osims:
            o.set_cartesian_position(O.Pose(translation=t, quaternion=down))

    script = {"above": (lambda: (grip.open(), [o.gripper_open() for o in osims], move([0.40, 0.0, 0.20])), 500),
              "down": (lambda: move([0.40, 0.0, 0.035]), 700),
              "closed": (lambda: (grip.shut(), [o.gripper_grasp() for o in osims]), 250),
              "lifted": (lambda: move([0.40, 0.0, 0.30]), 600),
              "held": (lambda: None, 300),
              "released": (lambda: (grip.open(), [o.gripper_open() for o in osims]), 300)}
    for tag in stages:
        script[tag][0]()
        advance(tag, script[tag][1])
    simu.close()
    return rep


def run_hard_pinch_parity(chunks=(1, 20, 60)):
    """The cube placement of tests/test_contacts_cpu.py::HARD_PINCH_PLACEMENT (found at batch scale: the coupled Newton solve
    of its closing pads is the hardest of 4096 random placements) among ordinary ones, pinched with the closing stage cut into
    launches of different lengths.  The robot's warm start inside a launch is not the one a fresh launch has (DESIGN design
    point 12), so the iterates differ between the splits: the RESULT must not.  Every split against the oracle."""
    import dataclasses

    from rcs_amd import common
    from rcs_amd import sim as S
    from rcs_amd.envs import default_sim_gripper_cfg, default_sim_robot_cfg
    from rcs_amd.mjcf import compile_mjcf
    import rcs_oracle as O
    from rcs_env_oracle import FR3_Q_HOME
    from test_contacts_cpu import HARD_PINCH_PLACEMENT

    n = 4
    qb = _pinch_placements(n, 3)
    qb[0] = HARD_PINCH_PLACEMENT
    cfg = dataclasses.replace(default_sim_robot_cfg("fr3_simple_pick_up"), tcp_offset=common.Pose(common.FrankaHandTCPOffset()))
    cm = compile_mjcf(PICKUP_SCENE)
    arm = [f"fr3_joint{i}_0" for i in range(1, 8)]
    osims = [O.Sim(cm, arm, arm, "attachment_site_0", "base_0", FR3_Q_HOME, O.franka_hand_tcp_offset(), "finger_joint1_0", "actuator8_0") for _ in range(n)]
    for e, o in enumerate(osims):
        o.reset(); o.robot_reset(); o.gripper_reset()
        o.box_qpos = qb[e]
        o.step(1)
    home = osims[0].get_cartesian_position()
    rep = {"max_newton": 0, "splits": {}}
    for o in osims:
        o.gripper_open()
        for xyz, k in (([0.44, 0.1, 0.20], 400), ([0.44, 0.1, 0.035], 600)):
            o.set_cartesian_position(O.Pose(translation=np.array(xyz), quaternion=home.rotation_q()))
            o.step(k)
        o.gripper_grasp()
        for _ in range(120):
            o.step(1)
            if o.s.d.coupled:
                rep["max_newton"] = max(rep["max_newton"], int(o.s.d.solver_niter))
    for ch in chunks:
        simu = S.Sim(cfg.mjcf_scene_path, S.SimConfig(), n_envs=n)
        if KERNEL != "auto":
            simu.set_kernel(KERNEL)
        robot = S.SimRobot(simu, None, cfg)
        grip = S.SimGripper(simu, default_sim_gripper_cfg())
        simu.reset(); robot.reset(); grip.reset()
        simu.set_free_joint_qpos("box_joint", qb)
        simu.step(1)
        grip.open()
        for xyz, k in (([0.44, 0.1, 0.20], 400), ([0.44, 0.1, 0.035], 600)):
            robot.set_cartesian_position(np.tile(np.concatenate([xyz, home.rotation_q()]), (n, 1)))
            simu.step(k)
        grip.shut()
        for i in range(0, 120, ch):
            simu.step(min(ch, 120 - i))
        q, v, bq, bv = simu.qpos, simu.qvel, simu.free_joint_qpos("box_joint"), simu.free_joint_qvel("box_joint")
        rep["splits"][ch] = {
            "qpos": max(float(np.abs(q[e] - np.asarray(o.qpos)).max()) for e, o in enumerate(osims)),
            "qvel": max(float(np.abs(v[e] - np.asarray(o.qvel)).max()) for e, o in enumerate(osims)),
            "box": max(float(np.abs(bq[e] - o.box_qpos).max()) for e, o in enumerate(osims)),
            "box_vel": max(float(np.abs(bv[e] - o.box_qvel).max()) for e, o in enumerate(osims)),
            "box_z": bq[:, 2].copy(),
        }
        simu.close()
    return rep


def run_pick_success_parity(n_envs=3, seed=0):
    """The registered pick-up task driven to SUCCESS: SimTaskEnvCreator with absolute joint actions (30 Hz async control), a
    scripted pinch-lift-swing; reward, `success` / `terminated`, is_grasped and the cube pose against the oracle's wrapper
    stack on the same actions.  Joint targets come from the oracle's IK, evaluated once on environment 0's trajectory."""
    from rcs_amd import common, sim
    from rcs_amd.envs import ControlMode, SimTaskEnvCreator, default_sim_robot_cfg
    from rcs_amd.mjcf import compile_mjcf
    import rcs_oracle as O
    from rcs_env_oracle import JOINTS, OraclePickCubeEnv

    cm = compile_mjcf(PICKUP_SCENE)
    tcp = O.Pose(translation=[0.0, 0.0, 0.1034], rotation=np.array([[0.707, 0.707, 0], [-0.707, 0.707, 0], [0, 0, 1]]))
    rc = default_sim_robot_cfg(scene="fr3_simple_pick_up")
    rc.tcp_offset = common.Pose(translation=np.array([0.0, 0.0, 0.1034]), rotation=np.array([[0.707, 0.707, 0], [-0.707, 0.707, 0], [0, 0, 1]]))
    venv = SimTaskEnvCreator()(rc, control_mode=ControlMode.JOINTS, delta_actions=False,
                               sim_cfg=sim.SimConfig(async_control=True, realtime=False, frequency=30), n_envs=n_envs)
    if KERNEL != "auto":
        venv.sim.set_kernel(KERNEL)
    oenvs = [OraclePickCubeEnv(cm, control_mode=JOINTS, delta_actions=False, tcp_offset=tcp, async_control=True) for _ in range(n_envs)]
    box = _pinch_placements(n_envs, seed)
    obs, info = venv.reset(options={"box_qpos": box})
    for e, oe in enumerate(oenvs):
        oe.reset(box_qpos=box[e])
    rep = {"max_abs_obs": 0.0, "max_abs_box": 0.0, "max_abs_reward": 0.0, "flag_mismatches": 0, "success_steps": 0, "grasped_steps": 0,
           "max_box_z": 0.0, "steps": 0, "truncated": 0}
    home = oenvs[0].sim.get_cartesian_position()
    q = np.asarray(oenvs[0].sim.qpos[:7]).copy()
    plan = []  # (joint target, gripper command, env-steps)
    for xyz, g, k in (([0.44, 0.1, 0.20], 1.0, 14), ([0.44, 0.1, 0.035], 1.0, 20), ([0.44, 0.1, 0.035], 0.0, 8), ([0.44, 0.1, 0.30], 0.0, 16)):
        sol, _ = oenvs[0].sim.ik_inverse(O.Pose(translation=np.array(xyz), quaternion=home.rotation_q()), q, tcp)
        assert sol is not None
        q = np.asarray(sol[:7]).copy()
        plan.append((q.copy(), g, k))
    qup = np.array([0, 0, 0, -0.2, 0, 2.0, 0.785])
    for k in range(1, 9):
        plan.append((q + (qup - q) * k / 8, 0.0, 5))
    plan.append((qup, 0.0, 12))
    for tgt, g, k in plan:
        for _ in range(k):
            a = {"joints": np.tile(tgt, (n_envs, 1)), "gripper": np.full(n_envs, g, dtype=np.float32)}
            obs, reward, term, trunc, info = venv.step(a)
            for e, oe in enumerate(oenvs):
                oo, orw, oterm, otrunc, oi = oe.step({"joints": tgt, "gripper": np.float32(g)})
                rep["max_abs_obs"] = max(rep["max_abs_obs"], float(np.abs(obs["tquat"][e] - oo["tquat"]).max()), float(np.abs(obs["joints"][e] - oo["joints"]).max()))
                rep["max_abs_box"] = max(rep["max_abs_box"], float(np.abs(info["box_qpos"][e] - oe.sim.box_qpos).max()))
                rep["max_abs_reward"] = max(rep["max_abs_reward"], abs(float(reward[e]) - float(orw)))
                rep["flag_mismatches"] += int(bool(term[e]) != bool(oterm)) + int(bool(trunc[e]) != bool(otrunc)) + int(bool(info["success"][e]) != bool(oi["success"]))
                rep["flag_mismatches"] += int(bool(info["is_grasped"][e]) != bool(oi["is_grasped"])) + int(float(obs["gripper"][e]) != float(oo["gripper"]))
                rep["success_steps"] += int(bool(oi["success"]) and bool(term[e]))
                rep["grasped_steps"] += int(bool(oi["is_grasped"]))
                rep["truncated"] += int(bool(otrunc))
            rep["max_box_z"] = max(rep["max_box_z"], float(info["box_qpos"][:, 2].min()))
            rep["contact_overflows"] = rep.get("contact_overflows", 0) + int(np.asarray(info["contact_overflow"]).sum())
            rep["steps"] += 1
    venv.close()
    return rep


def run_floor_contact_parity(n_envs=8, seed=0):
    """fr3_empty_world with robot contacts RESOLVED (opt-in: Sim(resolve_robot_contacts=True)): every environment drives its arm
    into the floor (hand and forearm come down on it), kernel vs oracle; the arm must stop ON the floor instead of sinking
    through it, the collision flags must fire in both."""
    from rcs_amd import sim as S
    from rcs_amd.envs import default_sim_gripper_cfg, default_sim_robot_cfg
    from rcs_amd.mjcf import compile_mjcf
    import rcs_oracle as O
    from rcs_env_oracle import FR3_Q_HOME

    cfg = default_sim_robot_cfg("fr3_empty_world")
    simu = S.Sim(cfg.mjcf_scene_path, S.SimConfig(), n_envs=n_envs, resolve_robot_contacts=True)
    assert simu.resolve_robot_contacts
    robot = S.SimRobot(simu, None, cfg)
    grip = S.SimGripper(simu, default_sim_gripper_cfg())
    cm = compile_mjcf(SCENE)
    arm = [f"fr3_joint{i}_0" for i in range(1, 8)]
    osims = [O.Sim(cm, arm, arm, "attachment_site_0", "base_0", FR3_Q_HOME, None, "finger_joint1_0", "actuator8_0", resolve_contacts=True) for _ in range(n_envs)]
    rng = np.random.default_rng(seed)
    simu.reset(); robot.reset(); grip.reset()
    for o in osims:
        o.reset(); o.robot_reset(); o.gripper_reset()
    # the reference's own collision configuration (python/tests/test_sim_envs.py:347-360) and variations of it
    tgt = np.tile([0, 1.78, 0, -1.45, 0, 0, 0], (n_envs, 1)) + rng.uniform(-0.15, 0.15, (n_envs, 7)) * (np.arange(n_envs) > 0)[:, None]
    tgt = np.clip(tgt, [-2.7, -1.78, -2.9, -3.04, -2.8, 0.55, -3.0], [2.7, 1.78, 2.9, -0.16, 2.8, 4.5, 3.0])
    robot.set_joint_position(tgt)
    for e, o in enumerate(osims):
        o.set_joint_position(tgt[e])
    rep = {"max_abs_qpos": 0.0, "max_abs_qvel": 0.0, "flag_mismatches": 0, "coupled_substeps": 0, "max_ncon": 0, "collisions": 0, "min_z": 9.0}
    for call in range(8):
        simu.step(100)
        q, v = simu.qpos, simu.qvel
        for e, o in enumerate(osims):
            for _ in range(100):
                o.step(1)
                rep["coupled_substeps"] += int(o.s.d.coupled)
                rep["max_ncon"] = max(rep["max_ncon"], int(o.s.d.ncon))
            rep["max_abs_qpos"] = max(rep["max_abs_qpos"], float(np.abs(q[e] - np.asarray(o.qpos)).max()))
            rep["max_abs_qvel"] = max(rep["max_abs_qvel"], float(np.abs(v[e] - np.asarray(o.qvel)).max()))
    simu.step_until_convergence()
    st, gs = robot.get_state(), grip.get_state()
    for e, o in enumerate(osims):
        o.step_until_convergence()
        rep["flag_mismatches"] += int(bool(st.collision[e]) != bool(o.s.robot_collision)) + int(bool(gs.collision[e]) != bool(o.s.grp_collision))
        rep["flag_mismatches"] += int(int(simu.convergence_steps()[e]) != int(o.s.convergence_steps))
        rep["collisions"] += int(bool(o.s.robot_collision) or bool(o.s.grp_collision))
        rep["min_z"] = min(rep["min_z"], float(o.get_cartesian_position().translation()[2]))
    rep["tracking_error"] = float(np.abs(simu.qpos[:, :7] - tgt).max())
    simu.close()
    return rep


def run_self_collision_parity(n_envs=40, seed=1, scene="fr3_empty_world", resolve=None):
    """step_until_convergence towards folded-arm targets inside the MODEL's joint ranges but outside the RobotEnv's limits
    (fine-grained API: SimRobot.set_joint_position does not clip): in about a third of them the fingers / the hand run into
    links 1 and 2 with nothing touching the floor.  Compared with the oracle per environment: both collision flags, the
    substep count at which the loop ended, the joint positions there."""
    from rcs_amd import sim as S
    from rcs_amd.envs import default_sim_gripper_cfg, default_sim_robot_cfg
    from rcs_amd.mjcf import compile_mjcf
    import rcs_oracle as O
    from rcs_env_oracle import FR3_Q_HOME

    cfg = default_sim_robot_cfg(scene)
    simu = S.Sim(cfg.mjcf_scene_path, S.SimConfig(), n_envs=n_envs, resolve_robot_contacts=resolve)
    robot = S.SimRobot(simu, None, cfg)
    grip = S.SimGripper(simu, default_sim_gripper_cfg())
    cm = compile_mjcf(cfg.mjcf_scene_path)
    arm = [f"fr3_joint{i}_0" for i in range(1, 8)]
    osims = [O.Sim(cm, arm, arm, "attachment_site_0", "base_0", FR3_Q_HOME, None, "finger_joint1_0", "actuator8_0",
                   resolve_contacts=simu.resolve_robot_contacts) for _ in range(n_envs)]
    rng = np.random.default_rng(seed)
    q = np.tile(FR3_Q_HOME, (n_envs, 1))
    q[:, 0] = rng.uniform(-1, 1, n_envs)
    q[:, 1] = rng.uniform(-1.78, 0.2, n_envs)
    q[:, 3] = rng.uniform(-3.04, -2.6, n_envs)
    q[:, 4] = rng.uniform(-0.5, 0.5, n_envs)
    q[:, 5] = rng.uniform(0.55, 1.6, n_envs)
    simu.step(1)
    robot.set_joint_position(q)
    simu.step_until_convergence()
    st, gs = robot.get_state(), grip.get_state()
    steps = simu.convergence_steps()
    qk = simu.qpos
    rep = {"flag_mismatches": 0, "substep_mismatches": 0, "max_abs_qpos": 0.0, "robot_hits": 0, "gripper_hits": 0, "self_only": 0, "floor": 0}
    for e, o in enumerate(osims):
        o.reset(); o.robot_reset(); o.gripper_reset(); o.step(1)
        o.set_joint_position(q[e])
        o.step_until_convergence()
        rep["flag_mismatches"] += int(bool(st.collision[e]) != bool(o.s.robot_collision)) + int(bool(gs.collision[e]) != bool(o.s.grp_collision))
        rep["substep_mismatches"] += int(int(steps[e]) != int(o.s.convergence_steps))
        rep["max_abs_qpos"] = max(rep["max_abs_qpos"], float(np.abs(qk[e] - o.qpos[: qk.shape[1]]).max()))
        rep["robot_hits"] += int(o.s.robot_collision)
        rep["gripper_hits"] += int(o.s.grp_collision)
        # (contacts of a robot body with floor / cube; where self contacts are resolved -- the default of round 5 -- d->contact also lists
        # those between two bodies of the robot)
        robot_other = sum(1 for c in range(o.s.d.ncon) if (o.s.d.contact[c].body[0] > 0) != (o.s.d.contact[c].body[1] > 0))
        rep["self_only"] += int((o.s.robot_collision or o.s.grp_collision) and robot_other == 0 and o.s.d.nself > 0)
        rep["floor"] += int(o.s.d.ncon > 0)
    simu.close()
    return rep


def run_rate_driven_camera_parity(n_envs=4, width=32, height=24, seed=3):
    """SimCameraSet(render_on_demand=False): frames at the cameras' frame rates from inside Sim.step / step_until_convergence
    (Sim::invoke_rendering_callbacks, sim.cpp:63-81,108-115).  The kernel records, the host renders after the launch; the
    restatement steps the oracle one substep at a time, applies the reference's rule (due when time - last > 1 / rate, clocks
    at -1 / rate after construction and after Sim::reset) and renders with the numpy ray-caster.  Compared per environment:
    the sequence of (timestamp, cameras) -- exactly -- and the pixels."""
    from rcs_amd import render
    from rcs_amd import sim as S
    from rcs_amd.camera import SimCameraConfig, SimCameraSet
    from rcs_amd.envs import default_sim_gripper_cfg, default_sim_robot_cfg
    from rcs_amd.mjcf import compile_mjcf
    import rcs_oracle as O
    import rcs_render_oracle as RO
    from rcs_env_oracle import FR3_Q_HOME

    cfg = default_sim_robot_cfg("fr3_simple_pick_up")
    simu = S.Sim(cfg.mjcf_scene_path, S.SimConfig(), n_envs=n_envs)
    robot = S.SimRobot(simu, None, cfg)
    S.SimGripper(simu, default_sim_gripper_cfg())
    rates = {"wrist_0": 30, "bird_eye_cam": 10}
    cams = {c: SimCameraConfig(identifier=c, frame_rate=r, resolution_width=width, resolution_height=height) for c, r in rates.items()}
    cs = SimCameraSet(simu, cams, physical_units=True, render_on_demand=False, max_framesets=10000)
    cs.set_double_precision(True)  # (this comparison is about WHAT is drawn and when: exact pixels against the float64 restatement)
    cm = compile_mjcf(PICKUP_SCENE)
    arm = [f"fr3_joint{i}_0" for i in range(1, 8)]
    osims = [O.Sim(cm, arm, arm, "attachment_site_0", "base_0", FR3_Q_HOME, None, "finger_joint1_0", "actuator8_0") for _ in range(n_envs)]
    campose = {c: render.camera_in_link(cm, c) for c in rates}
    last = [{c: -1.0 / r for c, r in rates.items()} for _ in range(n_envs)]
    expect = [[] for _ in range(n_envs)]  # per environment: (timestamp, {camera: (mm, rgb)})

    def oracle_substep(e, o):
        o.step(1)
        t = float(o.s.d.time)
        due = [c for c, r in rates.items() if t - last[e][c] > 1.0 / r]
        if due:
            fr = RO.oracle_frames(o, cm)
            imgs = {}
            for c in due:
                last[e][c] = t
                link, pos, rot, fovy = campose[c]
                _, mm, _, _, rgb = RO.render_depth(cs._scene, (link, pos, rot, fovy, width, height), fr, colour=True)
                imgs[c] = (mm, rgb[::-1])
            expect[e].append((t, imgs))

    rng = np.random.default_rng(seed)
    # 1. a plain Sim.step: the first substep renders both cameras, then each at its own rate
    tgt = FR3_Q_HOME + rng.uniform(-0.3, 0.3, (n_envs, 7))
    robot.set_joint_position(tgt)
    simu.step(60)
    for e, o in enumerate(osims):
        o.set_joint_position(tgt[e])
        for _ in range(60):
            oracle_substep(e, o)
    # 2. Sim.reset of half the batch: their camera clocks restart, the others' go on -- the batch is out of step from here
    mask = np.arange(n_envs) % 2 == 0
    simu.reset(mask)
    for e, o in enumerate(osims):
        if mask[e]:
            o.reset()
            last[e] = {c: -1.0 / r for c, r in rates.items()}
    # 3. step_until_convergence: hundreds of substeps, ~15 frames of the wrist camera per environment in one launch
    tgt = FR3_Q_HOME + rng.uniform(-0.2, 0.2, (n_envs, 7))
    robot.set_joint_position(tgt)
    simu.step_until_convergence()
    steps = simu.convergence_steps()
    for e, o in enumerate(osims):
        o.set_joint_position(tgt[e])
        o.s.convergence_steps = 0
        # (Sim::step_until_convergence = step(1) in a loop: the same substeps, the rendering callback after each)
        for _ in range(int(steps[e])):
            oracle_substep(e, o)
    rep = {"events": 0, "timestamp_mismatches": 0, "camera_set_mismatches": 0, "pixels": 0, "mismatched_mm": 0, "rgb_off_by_more_than_one": 0,
           "max_abs_qpos": 0.0, "events_per_env": []}
    qk = simu.qpos
    for e, o in enumerate(osims):
        rep["max_abs_qpos"] = max(rep["max_abs_qpos"], float(np.abs(qk[e] - o.qpos[: qk.shape[1]]).max()))
        got = [(float(ev["timestamp"][e]), {c: ev for c, have in ev["have"].items() if have[e]}) for ev in cs._buffer if not np.isnan(ev["timestamp"][e])]
        rep["events_per_env"].append((len(got), len(expect[e])))
        if len(got) != len(expect[e]):
            rep.setdefault("debug", []).append(([g[0] for g in got], [x[0] for x in expect[e]], int(steps[e])))
        rep["events"] += len(expect[e])
        if len(got) != len(expect[e]):
            rep["timestamp_mismatches"] += abs(len(got) - len(expect[e]))
            continue
        for (tg, cg), (to, co) in zip(got, expect[e]):
            rep["timestamp_mismatches"] += int(tg != to)
            rep["camera_set_mismatches"] += int(set(cg) != set(co))
            for c in set(cg) & set(co):
                ev = cg[c]
                near, far = cs._scene.znear, cs._scene.zfar
                z = near / (1 - ev["depth"][c][e][::-1] * (1 - near / far))
                mm = (z * 1000).astype(np.uint16)
                diff = mm.astype(np.int64) != co[c][0].astype(np.int64)
                rep["pixels"] += diff.size
                rep["mismatched_mm"] += int(diff.sum())
                cd = np.abs(ev["color"][c][e][::-1].astype(np.int64) - co[c][1].astype(np.int64)).max(axis=-1)
                rep["rgb_off_by_more_than_one"] += int(((cd > 1) & ~diff).sum())
    latest = cs.get_latest_frames()
    rep["latest_timestamp_ok"] = bool(all(float(latest.avg_timestamp[e]) == expect[e][-1][0] for e in range(n_envs)))
    rep["obs_keys"] = sorted(latest.frames)
    simu.close()
    return rep


def run_cube_against_base_parity(n_envs=12, n_calls=6, k=25, seed=5):
    """The free cube thrown at / dropped onto the robot's base: link 0's collision hull is welded to the world, so MuJoCo
    filters it against the floor but not against the cube (mjc_Convex, box first).  Kernel vs oracle: cube pose, how many
    environments saw a (link 0, cube) contact."""
    from rcs_amd import sim as S
    from rcs_amd.envs import default_sim_gripper_cfg, default_sim_robot_cfg
    from rcs_amd.mjcf import compile_mjcf
    import rcs_oracle as O
    from rcs_env_oracle import FR3_Q_HOME

    cfg = default_sim_robot_cfg("fr3_simple_pick_up")
    simu = S.Sim(cfg.mjcf_scene_path, S.SimConfig(), n_envs=n_envs)
    robot = S.SimRobot(simu, None, cfg)
    S.SimGripper(simu, default_sim_gripper_cfg())
    cm = compile_mjcf(PICKUP_SCENE)
    arm = [f"fr3_joint{i}_0" for i in range(1, 8)]
    osims = [O.Sim(cm, arm, arm, "attachment_site_0", "base_0", FR3_Q_HOME, None, "finger_joint1_0", "actuator8_0") for _ in range(n_envs)]
    g0 = cm.name2id("geom", "fr3_link0_collision_0")
    rng = np.random.default_rng(seed)
    ang = rng.uniform(-np.pi, np.pi, n_envs)
    rad = rng.uniform(0.16, 0.22, n_envs)
    qb = np.zeros((n_envs, 7))
    qb[:, 0], qb[:, 1], qb[:, 2] = rad * np.cos(ang), rad * np.sin(ang), rng.uniform(0.03, 0.12, n_envs)
    qb[:, 3:] = rng.normal(size=(n_envs, 4))
    vb = np.zeros((n_envs, 6))
    vb[:, 0], vb[:, 1] = -1.2 * np.cos(ang), -1.2 * np.sin(ang)  # towards the base
    vb[:, 3:] = rng.uniform(-2, 2, (n_envs, 3))
    simu.set_free_joint_qpos("box_joint", qb)
    simu.set_free_joint_qvel("box_joint", vb)
    for e, o in enumerate(osims):
        o.box_qpos, o.box_qvel = qb[e], vb[e]
    rep = {"max_abs_pos": 0.0, "max_abs_quat": 0.0, "max_abs_robot_qpos": 0.0, "base_contact_envs": set(), "max_ncon": 0,
           "env_pos_err": np.zeros(n_envs)}
    home = np.tile(FR3_Q_HOME, (n_envs, 1))
    for _ in range(n_calls):
        robot.set_joint_position(home)
        simu.step(k)
        qk, qr = simu.free_joint_qpos("box_joint"), simu.qpos
        for e, o in enumerate(osims):
            o.set_joint_position(home[e])
            for _ in range(k):
                o.step(1)
                d = o.s.d
                rep["max_ncon"] = max(rep["max_ncon"], int(d.ncon))
                if any(g0 in (d.contact[c].geom[0], d.contact[c].geom[1]) for c in range(d.ncon)):
                    rep["base_contact_envs"].add(e)
            rep["env_pos_err"][e] = max(rep["env_pos_err"][e], float(np.abs(qk[e, :3] - o.box_qpos[:3]).max()))
            rep["max_abs_pos"] = max(rep["max_abs_pos"], float(np.abs(qk[e, :3] - o.box_qpos[:3]).max()))
            rep["max_abs_quat"] = max(rep["max_abs_quat"], float(np.abs(qk[e, 3:] - o.box_qpos[3:]).max()))
            rep["max_abs_robot_qpos"] = max(rep["max_abs_robot_qpos"], float(np.abs(qr[e] - o.qpos).max()))
    rep["base_contact_envs"] = len(rep["base_contact_envs"])
    rep["final_radius"] = np.hypot(qk[:, 0], qk[:, 1])
    simu.close()
    return rep


def batch_pinch(n, chunk, seed=0, spread=0.004, yaw=0.1):
    """One scripted pinch / lift / release of `n` environments at once, stepping in launches of `chunk` substeps; the state
    after each stage."""
    import dataclasses

    from rcs_amd import common
    from rcs_amd import sim as S
    from rcs_amd.envs import default_sim_gripper_cfg, default_sim_robot_cfg

    cfg = dataclasses.replace(default_sim_robot_cfg("fr3_simple_pick_up"), tcp_offset=common.Pose(common.FrankaHandTCPOffset()))
    simu = S.Sim(cfg.mjcf_scene_path, S.SimConfig(), n_envs=n)
    robot = S.SimRobot(simu, None, cfg)
    grip = S.SimGripper(simu, default_sim_gripper_cfg())
    rng = np.random.default_rng(seed)
    qb = np.tile(np.array([0.44, 0.1, 0.0288, 0, 0, 0, 1.0]), (n, 1))
    qb[1:, 0] += rng.uniform(-spread, spread, n - 1)
    qb[1:, 1] += rng.uniform(-spread, spread, n - 1)
    y = np.zeros(n); y[1:] = rng.uniform(-yaw, yaw, n - 1)
    qb[:, 3], qb[:, 6] = np.cos((np.pi + y) / 2), np.sin((np.pi + y) / 2)
    simu.reset(); robot.reset(); grip.reset()
    simu.set_free_joint_qpos("box_joint", qb)
    simu.step(1)
    home = np.asarray(robot.get_cartesian_position())[0, 3:]

    def run(k):
        for i in range(0, k, chunk):
            simu.step(min(chunk, k - i))

    out = {}
    robot.set_cartesian_position(np.tile(np.concatenate([[0.44, 0.1, 0.2], home]), (n, 1))); run(400)
    robot.set_cartesian_position(np.tile(np.concatenate([[0.44, 0.1, 0.035], home]), (n, 1))); run(600)
    grip.shut(); run(200)
    out["closed"] = (simu.qpos.copy(), simu.free_joint_qpos("box_joint").copy())
    robot.set_cartesian_position(np.tile(np.concatenate([[0.44, 0.1, 0.3], home]), (n, 1))); run(500)
    out["lifted"] = (simu.qpos.copy(), simu.free_joint_qpos("box_joint").copy())
    grip.open(); run(300)
    out["released"] = (simu.qpos.copy(), simu.free_joint_qpos("box_joint").copy())
    simu.close()
    return out


def run_split_consistency(n=4096, ca=17, cb=100, **kw):
    """The same pinch at batch scale with the stepping cut into launches of different lengths, kernel against kernel.  The
    coupled solve's starting point inside a launch differs from a fresh launch's, so equal results for every environment say
    that every solve of every environment reached its minimiser (no stalled or capped Newton solve anywhere)."""
    a, b = batch_pinch(n, ca, **kw), batch_pinch(n, cb, **kw)
    rep = {}
    for tag in a:
        dq = np.abs(a[tag][0] - b[tag][0]).max(axis=1)
        db = np.abs(a[tag][1] - b[tag][1]).max(axis=1)
        rep[tag] = {"max_dq": float(dq.max()), "max_dbox": float(db.max()), "envs_over_1e-8": int(((dq > 1e-8) | (db > 1e-8)).sum()),
                    "worst_env": int(np.argmax(np.maximum(dq, db))), "box_z": (float(a[tag][1][:, 2].min()), float(a[tag][1][:, 2].max()))}
    return rep


def oracle_contacts_at_current_qpos(osim, touch=1e-9):
    """(ncon, nself) MuJoCo's collision pass would report at the oracle environment's CURRENT qpos -- what the next mj_step1 will
    see, the position the kernels' end-of-launch check (csrc/check_team.h) tests.  Computed on a copy of the oracle's data, so
    that neither the frames get_cartesian_position reads (those of the last mj_step1) nor the collision caches are disturbed."""
    import ctypes as C

    import rcs_oracle as O

    d = O.OrcData.from_buffer_copy(osim.s.d)
    L = O.lib()
    L.orc_kinematics(C.byref(osim.model), C.byref(d))
    L.orc_collide(C.byref(osim.model), C.byref(d))
    # "in contact": penetrating by more than a nanometre (csrc/check_team.h: kCheckTouch -- a pair that touches exactly, like the
    # fingertip pads at finger qpos 0 after every reset, has no reproducible sign)
    ncon = sum(1 for i in range(d.ncon) if d.contact[i].dist < -touch)
    nself = sum(1 for i in range(d.nself) if d.self_depth[i] > touch)
    return ncon, nself


def run_headline_contact_check(n_envs=64, n_steps=1000, seed=0, chunk=50):
    """The headline workload (fr3_empty_world, JOINTS, relative +-5 deg LAST_STEP actions, async 17 substeps, NO resets) run for
    BASELINE.md's rollout length against TWO oracle instances per environment:

    * one with contacts RESOLVED, as MuJoCo resolves them: its collision passes -- every substep's -- say in which env-step the
      environment first touches anything (`first_event`); until then the kernel must match it to round-off, afterwards the two differ by
      construction (the oracle stops on the floor, the lean kernel does not);
    * one that, like the lean kernel, resolves nothing: it follows the kernel's trajectory for the whole rollout, and its collision
      pass on the position an env-step ENDS on is exactly what the kernels' end-of-launch check evaluates: the kernel's sticky
      info["contact_unresolved"] must come on in exactly the env-step in which that pass first reports a contact (`first_boundary`).

    What the end-of-launch check cannot see is a contact that begins and ends inside one launch (a graze of a few substeps, sub-millimetre
    in this workload): `first_event < first_boundary` in those environments, counted in `transient_before_flag`."""
    import rcs_oracle as O

    venv = make_vec_env(n_envs, True, resolve_robot_contacts=False)  # (the lean kernels + the flag: the round-4 configuration)
    venv.on_unresolved_contact = "flag"
    saved = O.DEFAULT_RESOLVE_CONTACTS
    try:
        O.DEFAULT_RESOLVE_CONTACTS = 3
        oenvs = make_oracle_envs(n_envs, True)
        O.DEFAULT_RESOLVE_CONTACTS = False
        lean = make_oracle_envs(n_envs, True)
    finally:
        O.DEFAULT_RESOLVE_CONTACTS = saved
    joints, grip = synthetic_actions(n_envs, n_steps, seed)
    venv.reset()
    for oe in oenvs + lean:
        oe.reset()
    first_event = np.full(n_envs, -1)
    first_boundary = np.full(n_envs, -1)
    first_kernel = np.full(n_envs, -1)
    rep = {"max_abs_qpos_unflagged": 0.0, "max_abs_qvel_unflagged": 0.0, "max_abs_qpos_lean": 0.0, "flag_mismatch_steps": 0, "kinds": {}}
    for t in range(n_steps):
        _, _, _, _, info = venv.step({"joints": joints[t], "gripper": grip[t]})
        flagged = np.asarray(info["contact_unresolved"], dtype=bool)
        q, v = venv.sim.qpos, venv.sim.qvel
        for e, oe in enumerate(oenvs):
            le = lean[e]
            le.step({"joints": joints[t, e], "gripper": grip[t, e]})
            rep["max_abs_qpos_lean"] = max(rep["max_abs_qpos_lean"], float(np.abs(q[e] - le.sim.qpos[: q.shape[1]]).max()))
            if first_boundary[e] < 0:
                ncon, nself = oracle_contacts_at_current_qpos(le.sim)
                if ncon + nself > 0:
                    first_boundary[e] = t
                    rep["kinds"][e] = (ncon, nself)
            if first_event[e] >= 0:
                continue  # (the resolving oracle's trajectory may have parted from the kernel's)
            oe.sim.s.d.pen_seen = 0.0
            oe.step({"joints": joints[t, e], "gripper": grip[t, e]})
            ncon, nself = oracle_contacts_at_current_qpos(oe.sim)
            if oe.sim.s.d.pen_seen > 1e-9 or ncon + nself > 0:
                first_event[e] = t
            else:
                rep["max_abs_qpos_unflagged"] = max(rep["max_abs_qpos_unflagged"], float(np.abs(q[e] - oe.sim.qpos[: q.shape[1]]).max()))
                rep["max_abs_qvel_unflagged"] = max(rep["max_abs_qvel_unflagged"], float(np.abs(v[e] - oe.sim.qvel[: v.shape[1]]).max()))
        newly = flagged & (first_kernel < 0)
        first_kernel[newly] = t
        rep["flag_mismatch_steps"] += int(((first_kernel >= 0) != (first_boundary >= 0)).sum())
    rep["first_oracle"] = first_boundary
    rep["first_event"] = first_event
    rep["first_kernel"] = first_kernel
    rep["flagged_oracle"] = int((first_boundary >= 0).sum())
    rep["flagged_kernel"] = int((first_kernel >= 0).sum())
    ev = np.where(first_event >= 0, first_event, n_steps + 1)
    fk = np.where(first_kernel >= 0, first_kernel, n_steps + 1)
    rep["flag_before_any_contact"] = int((fk < ev).sum())          # false positives: must be 0
    rep["transient_before_flag"] = int((ev < fk).sum())           # contacts that began and ended inside a launch before the flag came on
    rep["sticky_accessor_equal"] = bool(np.array_equal(venv.sim.contact_unresolved(), first_kernel >= 0))
    venv.close()
    return rep


def run_self_contact_parity(n_envs=24, seed=1, launches=40, substeps=17, mode=7):
    """Round 5: contacts between two geoms of the robot are RESOLVED (reference src/sim/sim.cpp:108-115: mj_step2 resolves every entry
    of mjData.contact).  Folded-arm targets through the fine-grained API, Sim.step(k) launches, kernel (resolve_robot_contacts `mode`:
    7 = environment by environment, 3 = the whole batch on the contact-resolving kernel) against the oracle with self-contact rows."""
    from rcs_amd import sim as S
    from rcs_amd.envs import default_sim_gripper_cfg, default_sim_robot_cfg
    from rcs_amd.mjcf import compile_mjcf
    import rcs_oracle as O
    from rcs_env_oracle import FR3_Q_HOME

    cfg = default_sim_robot_cfg("fr3_empty_world")
    simu = S.Sim(cfg.mjcf_scene_path, S.SimConfig(async_control=True), n_envs=n_envs, resolve_robot_contacts=mode)
    robot = S.SimRobot(simu, None, cfg)
    S.SimGripper(simu, default_sim_gripper_cfg())
    cm = compile_mjcf(cfg.mjcf_scene_path)
    arm = [f"fr3_joint{i}_0" for i in range(1, 8)]
    rng = np.random.default_rng(seed)
    q = np.tile(FR3_Q_HOME, (n_envs, 1))
    q[:, 0] = rng.uniform(-1, 1, n_envs)
    q[:, 1] = rng.uniform(-1.78, 0.2, n_envs)
    q[:, 3] = rng.uniform(-3.04, -2.6, n_envs)
    q[:, 4] = rng.uniform(-0.5, 0.5, n_envs)
    q[:, 5] = rng.uniform(0.55, 1.6, n_envs)
    osims = []
    for e in range(n_envs):
        o = O.Sim(cm, arm, arm, "attachment_site_0", "base_0", FR3_Q_HOME, None, "finger_joint1_0", "actuator8_0", resolve_contacts=3)
        o.s.async_control = 1
        o.reset(); o.robot_reset(); o.gripper_reset(); o.step(1)
        o.set_joint_position(q[e])
        osims.append(o)
    simu.step(1)
    robot.set_joint_position(q)
    rep = {"max_abs_qpos": 0.0, "max_abs_qvel": 0.0, "max_contacts": 0, "self_contact_substeps": 0}
    touched = np.zeros(n_envs, dtype=bool)
    for _ in range(launches):
        simu.step(substeps)
        qk, vk = simu.qpos, simu.qvel
        for e, o in enumerate(osims):
            for _ in range(substeps):
                o.step(1)
                d = o.s.d
                nself = sum(1 for c in range(d.ncon) if d.contact[c].body[0] > 0 and d.contact[c].body[1] > 0)
                rep["self_contact_substeps"] += int(nself > 0)
                rep["max_contacts"] = max(rep["max_contacts"], int(d.ncon))
                touched[e] |= d.ncon > 0
            rep["max_abs_qpos"] = max(rep["max_abs_qpos"], float(np.abs(qk[e] - o.qpos[: qk.shape[1]]).max()))
            rep["max_abs_qvel"] = max(rep["max_abs_qvel"], float(np.abs(vk[e] - o.qvel[: vk.shape[1]]).max()))
    now, ever = simu.contact_escalated()
    rep["touched"] = touched
    rep["escalated_now"] = now
    rep["resolved_ever"] = ever
    rep["in_contact_at_end"] = np.array([o.s.d.ncon > 0 for o in osims])
    rep["tracking_error"] = np.abs(simu.qpos[:, :7] - q).max(axis=1)
    rep["overflow"] = int(np.asarray(simu.contact_overflow()).sum()) if hasattr(simu, "contact_overflow") else 0
    simu.close()
    return rep


def run_headline_resolved_parity(n_envs=64, n_steps=1000, seed=0):
    """The headline workload (fr3_empty_world, JOINTS, relative +-5 deg LAST_STEP random actions, async 17 substeps, NO resets) for
    BASELINE.md's rollout length with every contact the robot runs into RESOLVED, environment by environment, against the oracle that
    resolves them too (floor and self contact rows, NO bound on the contact list): every environment, every step -- nobody is excluded.

    Reported next to the errors (not masks -- counts the caller may assert on):
    * `graze_steps`: env-steps in which the oracle saw a penetration in some substep and none on the position the step ends on (a contact
      that begins AND ends inside one launch: what round 5's check of the final position alone could not see; the certifying check,
      csrc/check_team.h, sends such a launch to the contact-resolving kernel).
    * `twin_split`: the env-step at which the ORACLE parts from itself (-1: never): every environment has a second oracle instance
      whose joint 4 is nudged by 1e-13 rad after the reset.  The two stay 1e-13 apart through smooth motion and through ordinary
      contacts -- and part by 1e-7 .. 1e-5 rad in the step in which two SHUT fingers' pads, which touch face to face with a gap of
      exactly 0.0, are pressed into each other: whether each of the 5 x 5 pad pairs then reports 0, 4 or 8 points is decided by the
      last bit (tools/oracle_sensitivity.py: 26 contacts in one run, 19 in its twin).  No implementation -- MuJoCo on another CPU
      included -- reproduces such a step to 1e-9, and from it on two runs of the SAME code are different rollouts (contact dynamics
      amplify: 1e-2 rad within a hundred steps).  The bars: every environment, every step BEFORE its twins part (distance <= 1e-10):
      the plain 1e-9 / 1e-8 and flags bit-equal (`excess_env`, `vexcess_env`, `flag_env`); the step they part in: 100 x the twins'
      distance; afterwards the error is reported (`post_split_err`), not held to anything.
    * `overflow_envs`: a contact phase ran out of its contact slots (info["contact_overflow"])."""
    import rcs_oracle as O

    venv = make_vec_env(n_envs, True)
    assert venv.sim.resolve_robot_contacts == 7
    oenvs = make_oracle_envs(n_envs, True)
    assert oenvs[0].sim.model.resolve_contacts == 3
    joints, grip = synthetic_actions(n_envs, n_steps, seed)
    venv.reset()
    twins = make_oracle_envs(n_envs, True)
    for oe, tw in zip(oenvs, twins):
        oe.reset()
        tw.reset()
        tw.sim.s.d.qpos[3] += 1e-13
    rep = {"max_abs_qpos": 0.0, "max_abs_qvel": 0.0, "max_abs_obs": 0.0, "flag_mismatches": 0, "worst_env": -1, "worst_step": -1}
    twin_err = np.zeros(n_envs)
    twin_verr = np.zeros(n_envs)
    excess = np.zeros(n_envs)    # the largest error before the environment's twins part (at the parting step: less 100 x their distance)
    vexcess = np.zeros(n_envs)
    twin_split = np.full(n_envs, -1)
    post_split_err = np.zeros(n_envs)
    first_contact = np.full(n_envs, -1)
    contact_steps = np.zeros(n_envs, dtype=int)
    err_env = np.zeros(n_envs)
    verr_env = np.zeros(n_envs)
    flag_env = np.zeros(n_envs, dtype=int)
    graze_steps = np.zeros(n_envs, dtype=int)    # env-steps with a contact that began and ended inside the step (see the docstring)
    overflow = np.zeros(n_envs, dtype=bool)      # a contact phase of the environment ran out of contact slots
    max_ncon = np.zeros(n_envs, dtype=int)
    first_bad = np.full(n_envs, -1)
    esc_count = np.zeros(n_steps, dtype=int)
    for t in range(n_steps):
        obs, _, _, trunc, info = venv.step({"joints": joints[t], "gripper": grip[t]})
        q, v = venv.sim.qpos, venv.sim.qvel
        overflow |= np.asarray(info["contact_overflow"], dtype=bool)
        esc_count[t] = int(venv.sim.contact_escalated()[0].sum())
        for e, oe in enumerate(oenvs):
            oe.sim.s.d.pen_seen = 0.0
            oo, _, _, otrunc, oi = oe.step({"joints": joints[t, e], "gripper": grip[t, e]})
            max_ncon[e] = max(max_ncon[e], int(oe.sim.s.d.ncon))
            if oe.sim.s.d.pen_seen > 1e-9:
                contact_steps[e] += 1
                if first_contact[e] < 0:
                    first_contact[e] = t
                if sum(oracle_contacts_at_current_qpos(oe.sim)) == 0:
                    graze_steps[e] += 1
            tw = twins[e]
            tw.step({"joints": joints[t, e], "gripper": grip[t, e]})
            twin_err[e] = max(twin_err[e], float(np.abs(np.asarray(tw.sim.qpos) - np.asarray(oe.sim.qpos)).max()))
            twin_verr[e] = max(twin_verr[e], float(np.abs(np.asarray(tw.sim.qvel) - np.asarray(oe.sim.qvel)).max()))
            dq = float(np.abs(q[e] - oe.sim.qpos[: q.shape[1]]).max())
            dv = float(np.abs(v[e] - oe.sim.qvel[: v.shape[1]]).max())
            if os.environ.get("RCS_TWIN_TRACE"):  # dev: "lo:hi" -- the steps to print the kernel's and the twins' distances for
                lo_, hi_ = (int(x) for x in os.environ["RCS_TWIN_TRACE"].split(":"))
                if lo_ <= t <= hi_:
                    print(f"step {t} env {e}: |dq| {dq:.2e} |dv| {dv:.2e} twins |dq| {twin_err[e]:.2e} |dv| {twin_verr[e]:.2e} contacts {int(oe.sim.s.d.ncon)}")
            if dq > 1e-9 and first_bad[e] < 0:
                first_bad[e] = t
            err_env[e] = max(err_env[e], dq)
            verr_env[e] = max(verr_env[e], dv)
            if twin_split[e] < 0 and twin_err[e] > 1e-10:
                twin_split[e] = t
                excess[e] = max(excess[e], dq - 100.0 * twin_err[e])
                vexcess[e] = max(vexcess[e], dv - 100.0 * twin_verr[e])
            elif twin_split[e] < 0:
                excess[e] = max(excess[e], dq)
                vexcess[e] = max(vexcess[e], dv)
            else:
                post_split_err[e] = max(post_split_err[e], dq)
            rep["max_abs_obs"] = max(rep["max_abs_obs"], float(np.abs(obs["joints"][e] - oo["joints"]).max()))
            if twin_split[e] < 0 or twin_split[e] == t:
                flag_env[e] += int(bool(info["collision"][e]) != bool(oi["collision"])) + int(bool(info["ik_success"][e]) != bool(oi["ik_success"]))
                flag_env[e] += int(bool(trunc[e]) != bool(otrunc)) + int(float(obs["gripper"][e]) != float(oo["gripper"]))
    now, ever = venv.sim.contact_escalated()
    rep["max_abs_qpos"] = float(err_env.max())
    rep["max_abs_qvel"] = float(verr_env.max())
    rep["flag_mismatches"] = int(flag_env.sum())
    rep["worst_env"] = int(np.argmax(err_env))
    rep["graze_steps"] = graze_steps
    rep["overflow_envs"] = overflow
    rep["twin_err_env"] = twin_err
    rep["excess_env"] = excess      # the caller's bar is 1e-9 on THIS
    rep["twin_split"] = twin_split
    rep["post_split_err"] = post_split_err
    rep["vexcess_env"] = vexcess
    rep["max_ncon"] = max_ncon
    rep["first_contact"] = first_contact
    rep["first_bad"] = first_bad
    rep["contact_steps"] = contact_steps
    rep["err_env"] = err_env
    rep["verr_env"] = verr_env
    rep["flag_env"] = flag_env
    rep["resolved_ever"] = ever
    rep["escalated_now"] = now
    rep["escalated_per_step"] = esc_count
    rep["unresolved"] = venv.sim.contact_unresolved()
    rep["overflow"] = int(np.asarray(info["contact_overflow"]).sum())
    venv.close()
    return rep
