"""dev tool: folded-arm targets through the fine-grained API, async stepping, contacts of the robot with itself RESOLVED: the
kernel (resolve_robot_contacts mode from argv: 3 whole batch, 7 per-environment escalation) against the oracle (mode 3).

    python tools/self_contact_probe.py [mode] [n_envs] [launches] [substeps per launch]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("robot-control-stack_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np  # noqa: E402


def main():
    mode = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    launches = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    k = int(sys.argv[4]) if len(sys.argv) > 4 else 17
    from rcs_amd import sim as S
    from rcs_amd.envs import default_sim_gripper_cfg, default_sim_robot_cfg
    from rcs_amd.mjcf import compile_mjcf
    import rcs_oracle as O
    from rcs_env_oracle import FR3_Q_HOME

    cfg = default_sim_robot_cfg("fr3_empty_world")
    simu = S.Sim(cfg.mjcf_scene_path, S.SimConfig(async_control=True), n_envs=n, resolve_robot_contacts=mode)
    robot = S.SimRobot(simu, None, cfg)
    S.SimGripper(simu, default_sim_gripper_cfg())
    cm = compile_mjcf(cfg.mjcf_scene_path)
    arm = [f"fr3_joint{i}_0" for i in range(1, 8)]
    rng = np.random.default_rng(1)
    q = np.tile(FR3_Q_HOME, (n, 1))
    q[:, 0] = rng.uniform(-1, 1, n)
    q[:, 1] = rng.uniform(-1.78, 0.2, n)
    q[:, 3] = rng.uniform(-3.04, -2.6, n)
    q[:, 4] = rng.uniform(-0.5, 0.5, n)
    q[:, 5] = rng.uniform(0.55, 1.6, n)
    osims = []
    for e in range(n):
        o = O.Sim(cm, arm, arm, "attachment_site_0", "base_0", FR3_Q_HOME, None, "finger_joint1_0", "actuator8_0", resolve_contacts=3)
        o.s.async_control = 1
        o.reset(); o.robot_reset(); o.gripper_reset(); o.step(1)
        o.set_joint_position(q[e])
        osims.append(o)
    simu.step(1)
    robot.set_joint_position(q)
    worst = np.zeros(n)
    first_bad = [None] * n
    incontact = np.zeros(n, dtype=bool)
    t0 = time.time()
    for it in range(launches):
        simu.step(k)
        qk, vk = simu.qpos, simu.qvel
        for e, o in enumerate(osims):
            o.step(k)
            incontact[e] |= o.s.d.ncon > 0
            err = float(np.abs(qk[e] - o.qpos[: qk.shape[1]]).max())
            worst[e] = max(worst[e], err)
            if err > 1e-8 and first_bad[e] is None:
                first_bad[e] = it
    print("mode", mode, "envs", n, "in contact", int(incontact.sum()), "time %.1fs" % (time.time() - t0))
    for e in range(n):
        o = osims[e]
        print(e, "contact" if incontact[e] else "free   ", "max|dq| %.2e" % worst[e], "first>1e-8 at launch", first_bad[e], "ncon", o.s.d.ncon, "niter", o.s.d.solver_niter)
    try:
        now, ever = simu.contact_escalated()
        print("escalated now", np.flatnonzero(now).tolist(), "ever resolved", np.flatnonzero(ever).tolist())
    except Exception as ex:  # noqa: BLE001
        print("contact_escalated:", ex)
    print("overflow", np.flatnonzero(simu.contact_overflow()).tolist() if hasattr(simu, "contact_overflow") else "?")
    print("WORST", worst.max())


if __name__ == "__main__":
    main()
