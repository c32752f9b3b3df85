"""Closed forms for the parts of the CPU oracle's physics that no mechanics derivation covers: MuJoCo's SOFT-CONSTRAINT model
(joint limit rows, joint-equality rows, dry-friction rows; CONTACT rows: a cube and a robot link at rest on the floor -- normal
rows, their regulariser's inverse weight -- and a pushed link creeping below the friction cone -- the tangential rows and
impratio) and the implicitfast integrator, on one- and two-dof models small enough for pencil and paper.  Each expected value below is worked out from the model MuJoCo documents ("Computation":
solref -> (K, B), solimp -> impedance d(r), R = (1 - d)/d . diagApprox, force = -D (J qacc - aref), D = 1/R) -- not from the
oracle -- and the oracle (oracle/rcs_physics.c, rcs_object.c, rcs_contact.c, which the HIP kernels are held to at 1e-9) has to
reproduce it; the resting cube is checked through the kernels as well (-m gpu).  What has no closed form and stays pinned by
the restatement alone: the cone's middle zone (sliding contacts), the noslip pass, the colliders' contact points.

The robot-scale counterpart is tests/test_dynamics_golden.py (mass matrix, bias, gravity compensation, actuation from a
Lagrangian derivation).  Together they leave of DESIGN.md section 5's "verify" list only what needs MuJoCo itself: the
numerical values of the defaults (solref 0.02 / 1, solimp 0.9 / 0.95 / 0.001 / 0.5 / 2 -- documented) and collision details.
"""
import ctypes as C
import os
import sys
import tempfile

import numpy as np
import pytest
from scipy.optimize import brentq

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(HERE, "..", "oracle"), os.path.join(HERE, "..", "robot-control-stack_amd"), HERE]

G = 9.81
H = 0.002
SOLREF, SOLIMP = (0.02, 1.0), (0.9, 0.95, 0.001, 0.5, 2.0)  # MuJoCo's documented defaults


def impedance(r, solimp=SOLIMP):
    d0, dmax, width, mid, power = solimp
    x = abs(r) / width
    if x >= 1:
        return dmax
    y = x**power / mid ** (power - 1) if x <= mid else 1 - (1 - x) ** power / (1 - mid) ** (power - 1)
    return d0 + y * (dmax - d0)


def stiffness_damping(solref=SOLREF, solimp=SOLIMP):
    tc, dr = max(solref[0], 2 * H), solref[1]
    return 1 / (solimp[1] ** 2 * tc**2 * dr**2), 2 / (solimp[1] * tc)


def toy(body_xml, extra=""):
    import rcs_oracle as O
    from rcs_amd.mjcf import compile_mjcf

    path = os.path.join(tempfile.mkdtemp(prefix="rcs_amd_toy"), "scene.xml")
    open(path, "w").write(f'<mujoco model="toy"><compiler angle="radian"/><option integrator="implicitfast"/>'
                          f"<worldbody>{body_xml}</worldbody>{extra}</mujoco>")
    m = O.make_model(compile_mjcf(path), False)
    d = O.OrcData()
    O.lib().orc_reset_data(C.byref(m), C.byref(d))
    return O, m, d


def step(O, m, d, n=1):
    for _ in range(n):
        O.lib().orc_step1(C.byref(m), C.byref(d))
        O.lib().orc_step2(C.byref(m), C.byref(d))


def test_joint_pressed_into_its_limit_settles_at_the_soft_penetration():
    """A 1.5 kg slider (armature 0.25) resting on its lower limit under gravity: the limit row's force D K d(r) |r| balances
    m g, with D = d / ((1 - d) diagApprox) and diagApprox = 1 / (m + armature)."""
    mass, arm = 1.5, 0.25
    O, m, d = toy(f'<body name="a" pos="0 0 1"><inertial mass="{mass}" pos="0 0 0" diaginertia="0.01 0.01 0.01"/>'
                  f'<joint name="ja" type="slide" axis="0 0 1" range="0 1" armature="{arm}" damping="3"/></body>')
    step(O, m, d, 6000)
    K, _ = stiffness_damping()
    r = brentq(lambda r: K * impedance(r) ** 2 / (1 - impedance(r)) * r * (mass + arm) - mass * G, 1e-9, 1e-2, xtol=1e-17)
    assert d.nefc == 1 and abs(d.qvel[0]) < 1e-13
    assert abs(d.qpos[0] + r) < 1e-12 and abs(d.efc_force[0] - mass * G) < 1e-10
    assert 2e-4 < r < 5e-4  # inside solimp's width: the impedance really varies with the penetration here


def test_implicitfast_on_a_damped_position_servo_is_the_analytic_update():
    """Hinge about the vertical (gravity does no work), inertia I + armature, joint damping c, position servo kp / kv:
    v+ = v + h (kp (u - q) - (kv + c) v) / (I + a + h (kv + c)),  q+ = q + h v+   -- implicit in the velocity-dependent forces."""
    inertia, arm, c, kp, kv = 0.3, 0.1, 1.5, 400.0, 40.0
    O, m, d = toy(f'<body name="a" pos="0 0 1"><inertial mass="2" pos="0 0 0" diaginertia="0.2 0.2 {inertia}"/>'
                  f'<joint name="ja" type="hinge" axis="0 0 1" armature="{arm}" damping="{c}"/></body>',
                  f'<actuator><position name="pa" joint="ja" kp="{kp}" kv="{kv}"/></actuator>')
    q, v, u = 0.2, -0.7, 0.5
    d.qpos[0], d.qvel[0], d.ctrl[0] = q, v, u
    for _ in range(50):
        v = v + H * (kp * (u - q) - (kv + c) * v) / (inertia + arm + H * (kv + c))
        q = q + H * v
        step(O, m, d)
        assert abs(d.qvel[0] - v) < 1e-13 and abs(d.qpos[0] - q) < 1e-14


def test_joint_equality_sags_by_the_soft_constraint_law():
    """Slider A (vertical, 0.8 kg, carries its weight) tied to slider B (horizontal, held at 0 by a position servo) by a joint
    equality with the FR3 fingers' solref / solimp: at rest the row's force equals A's weight, q_B = -m g / kp, and
    r = q_A - q_B solves  K d(r)^2 / ((1 - d(r)) (1/M_A + 1/M_B)) |r| = m g."""
    ma, mb, arm, kp = 0.8, 0.5, 0.1, 2000.0
    solref, solimp = (0.005, 1.0), (0.95, 0.99, 0.001, 0.5, 2.0)
    O, m, d = toy(f'<body name="a" pos="0 0 1"><inertial mass="{ma}" pos="0 0 0" diaginertia="0.01 0.01 0.01"/>'
                  f'<joint name="ja" type="slide" axis="0 0 1" armature="{arm}" damping="20"/></body>'
                  f'<body name="b" pos="1 0 1"><inertial mass="{mb}" pos="0 0 0" diaginertia="0.01 0.01 0.01"/>'
                  f'<joint name="jb" type="slide" axis="1 0 0" armature="{arm}" damping="20"/></body>',
                  f'<equality><joint joint1="ja" joint2="jb" solref="{solref[0]} {solref[1]}" solimp="{solimp[0]} {solimp[1]} {solimp[2]}"/></equality>'
                  f'<actuator><position name="pb" joint="jb" kp="{kp}"/></actuator>')
    step(O, m, d, 8000)
    K, _ = stiffness_damping(solref, solimp)
    diag = 1 / (ma + arm) + 1 / (mb + arm)
    r = brentq(lambda r: K * impedance(r, solimp) ** 2 / ((1 - impedance(r, solimp)) * diag) * r - ma * G, 1e-10, 1e-2, xtol=1e-18)
    assert d.nefc == 1 and max(abs(d.qvel[0]), abs(d.qvel[1])) < 1e-12
    assert abs(d.qpos[1] + ma * G / kp) < 1e-12
    assert abs((d.qpos[0] - d.qpos[1]) + r) < 1e-12, (d.qpos[0] - d.qpos[1], r)
    assert abs(abs(d.efc_force[0]) - ma * G) < 1e-9


@pytest.mark.parametrize("load", [0.5, 3.0])
def test_dry_friction_row_creeps_below_and_slides_above_its_limit(load):
    """A slider on a slope with frictionloss F = 1 (the xArm7 joints' value).  The friction row has no stiffness (aref = -B v)
    and impedance d(0) = solimp[0].  Load below F: the quadratic zone holds the load at the creep velocity
    v = load R B... i.e. load / (D B); load above F: the row saturates at F and the slider accelerates with (load - F) / M."""
    mass, arm, fl = 1.0, 0.2, 1.0
    sin_t = load / (mass * G)
    cos_t = float(np.sqrt(1 - sin_t * sin_t))
    O, m, d = toy(f'<body name="a" pos="0 0 1"><inertial mass="{mass}" pos="0 0 0" diaginertia="0.01 0.01 0.01"/>'
                  f'<joint name="ja" type="slide" axis="{cos_t!r} 0 {-sin_t!r}" armature="{arm}" frictionloss="{fl}"/></body>')
    _, B = stiffness_damping()
    d0 = SOLIMP[0]
    D = d0 / ((1 - d0) / (mass + arm))
    if load < fl:
        step(O, m, d, 3000)
        assert abs(d.qvel[0] - load / (D * B)) < 1e-13 * max(1, abs(d.qvel[0])) + 1e-15, (d.qvel[0], load / (D * B))
        assert abs(d.efc_force[0] + load) < 1e-11
    else:
        step(O, m, d, 5)
        v0 = d.qvel[0]
        step(O, m, d, 1)
        assert abs((d.qvel[0] - v0) / H - (load - fl) / (mass + arm)) < 1e-9
        assert abs(d.efc_force[0] + fl) < 1e-12


def _resting_depth(mass, ncon=4):
    """Penetration at which `ncon` equal contact normal rows carry m g: per row force = D K d(r) |r| with D = d / ((1 - d) / m)
    (diagApprox of a free body's translation against the world: 1 / m)."""
    K, _ = stiffness_damping()
    return brentq(lambda r: ncon * K * impedance(r) ** 2 / (1 - impedance(r)) * r * mass - mass * G, 1e-9, 1e-2, xtol=1e-18)


def test_cube_at_rest_on_the_floor_sinks_by_the_soft_contact_law():
    """The pick-up scene's cube (rcs_object.c, standalone: no robot) dropped flat onto the floor: plane-box gives four corner
    contacts of equal depth; at rest each normal row carries m g / 4, i.e. the cube's centre ends |r| below its half height
    with  4 K d(r)^2 / (1 - d(r)) |r| m = m g  -- MuJoCo's documented contact model (impedance from solimp, stiffness from
    solref, regulariser (1 - d) / d times the pair's inverse weight), no property of this code.  Friction rows, the elliptic
    cone and the noslip pass must leave that untouched."""
    import rcs_oracle as O

    lib = O.lib()
    b = O.OrcBox()
    half = (0.032, 0.016, 0.0288)
    mass = 50.0 * 8 * half[0] * half[1] * half[2]  # density 50 (assets/scenes/fr3_simple_pick_up/scene.xml:30-33)
    b.present, b.mass = 1, mass
    b.inertia[:] = [mass / 3 * (half[1] ** 2 + half[2] ** 2), mass / 3 * (half[0] ** 2 + half[2] ** 2), mass / 3 * (half[0] ** 2 + half[1] ** 2)]
    b.size[:] = half
    b.friction[:] = [1.0, 0.3, 0.1]
    b.geom_friction[:] = [1.0, 0.3, 0.1]
    b.solref[:], b.solimp[:] = SOLREF, SOLIMP
    b.plane_z, b.impratio, b.noslip_tolerance, b.noslip_iterations, b.nv_total = 0.0, 20.0, 1e-6, 5, 6
    b.meaninertia = (3 * mass + sum(b.inertia)) / 6
    b.qpos0[:] = [0.1, -0.2, half[2] + 0.002, np.cos(0.35), 0, 0, np.sin(0.35)]  # 2 mm above the floor, yawed
    d = O.OrcBoxData()
    lib.orc_box_reset(C.byref(b), C.byref(d))
    gravity = (C.c_double * 3)(0, 0, -G)
    for _ in range(4000):
        lib.orc_box_step1(C.byref(b), C.byref(d), C.c_double(H))
        lib.orc_box_step2(C.byref(b), C.byref(d), gravity, C.c_double(H), C.c_double(0.0))
    r = _resting_depth(mass)
    assert d.ncon == 4 and max(abs(v) for v in d.qvel) < 1e-12
    assert abs(d.qpos[2] - (half[2] - r)) < 1e-12, (d.qpos[2], half[2] - r)
    assert all(abs(d.force[3 * c] - mass * G / 4) < 1e-9 for c in range(4))  # normal rows; (friction rows: nothing to resist)
    assert all(abs(d.force[3 * c + k]) < 1e-9 for c in range(4) for k in (1, 2))
    assert abs(d.qpos[0] - 0.1) < 1e-12 and abs(d.qpos[1] + 0.2) < 1e-12  # it did not creep sideways
    assert 5e-5 < r < 3e-4  # inside solimp's width


@pytest.mark.gpu
def test_hip_cube_at_rest_on_the_floor_sinks_by_the_soft_contact_law():
    """The same law through the kernels: the pick-up scene's cube (csrc/box_team.h inside k_run_team) at rest beside the robot."""
    from rcs_amd import sim as S
    from rcs_amd.envs import default_sim_gripper_cfg, default_sim_robot_cfg

    cfg = default_sim_robot_cfg("fr3_simple_pick_up")
    simu = S.Sim(cfg.mjcf_scene_path, S.SimConfig(), n_envs=8)
    S.SimRobot(simu, None, cfg)
    S.SimGripper(simu, default_sim_gripper_cfg())
    half = (0.032, 0.016, 0.0288)
    mass = 50.0 * 8 * half[0] * half[1] * half[2]
    rng = np.random.default_rng(0)
    qb = np.zeros((8, 7))
    qb[:, 0], qb[:, 1], qb[:, 2] = rng.uniform(0.4, 0.6, 8), rng.uniform(-0.2, 0.2, 8), half[2] + rng.uniform(0.0, 0.003, 8)
    yaw = rng.uniform(-1, 1, 8)
    qb[:, 3], qb[:, 6] = np.cos(yaw / 2), np.sin(yaw / 2)
    simu.set_free_joint_qpos("box_joint", qb)
    simu.step(4000)
    z = simu.free_joint_qpos("box_joint")[:, 2]
    v = simu.free_joint_qvel("box_joint")
    r = _resting_depth(mass)
    assert np.abs(v).max() < 1e-11, np.abs(v).max()
    assert np.abs(z - (half[2] - r)).max() < 1e-11, (z, half[2] - r)
    simu.close()


def test_link_resting_on_the_floor_sinks_by_the_soft_contact_law_with_its_inverse_weight():
    """A robot LINK on the floor (rcs_contact.c: contact rows of a robot geom, the coupled solve): a vertical slider of mass m
    and armature a carrying a box geom, let down flat.  Four corner contacts; the regulariser's inverse weight is MuJoCo's
    body_invweight0 -- the translational mean (1/3) trace(J M^-1 J') = 1 / (3 (m + a)) for a body that can only move along z --
    so  4 . 3 (m + a) K d(r)^2 / (1 - d(r)) |r| = m g."""
    import rcs_oracle as O
    from rcs_amd.mjcf import compile_mjcf

    mass, arm, half = 0.8, 0.2, (0.05, 0.03, 0.02)
    path = os.path.join(tempfile.mkdtemp(prefix="rcs_amd_toy"), "scene.xml")
    open(path, "w").write(
        '<mujoco model="toy"><compiler angle="radian"/><option integrator="implicitfast" cone="elliptic" impratio="20"/><worldbody>'
        '<geom name="floor" type="plane" size="0 0 0.05"/>'
        f'<body name="a" pos="0 0 {half[2] + 0.001}"><inertial mass="{mass}" pos="0 0 0" diaginertia="0.01 0.01 0.01"/>'
        f'<joint name="ja" type="slide" axis="0 0 1" armature="{arm}" damping="2"/>'
        f'<geom name="foot" type="box" size="{half[0]} {half[1]} {half[2]}" mass="0"/></body></worldbody></mujoco>')
    m = O.make_model(compile_mjcf(path), True)
    d = O.OrcData()
    O.lib().orc_reset_data(C.byref(m), C.byref(d))
    step(O, m, d, 5000)
    K, _ = stiffness_damping()
    M = mass + arm
    r = brentq(lambda r: 4 * 3 * M * K * impedance(r) ** 2 / (1 - impedance(r)) * r - mass * G, 1e-9, 1e-2, xtol=1e-18)
    assert d.ncon == 4 and d.coupled == 1 and abs(d.qvel[0]) < 1e-12
    assert abs(d.qpos[0] - (-0.001 - r)) < 1e-12, (d.qpos[0], -0.001 - r)  # the joint's zero is 1 mm above touching
    assert 1e-5 < r < 3e-4


def test_pushed_link_on_the_floor_creeps_at_the_rate_of_the_regularised_friction_rows():
    """Below the friction cone MuJoCo's soft friction rows act as dampers: reference acceleration -B v (no position term),
    regulariser R1 = R0 / impratio.  A foot on the floor (x slider carrying a z slider carrying a box geom) pushed sideways by
    F < mu N therefore creeps at the steady rate where the four contacts' tangential rows balance the push:
    4 (1 / R1) B v = F  with  R1 = (1 - d(r)) / d(r) . invweight / impratio,  invweight = (1/Mx + 1/Mz) / 3  (the mean translational
    inverse inertia of the foot's body), r the resting depth of the normal rows.  No noslip pass in this scene."""
    import rcs_oracle as O
    from rcs_amd.mjcf import compile_mjcf

    m1, m2, ax, az, half, impratio, F = 0.5, 0.8, 0.1, 0.2, (0.05, 0.03, 0.02), 20.0, 1.5
    path = os.path.join(tempfile.mkdtemp(prefix="rcs_amd_toy"), "scene.xml")
    open(path, "w").write(
        f'<mujoco model="toy"><compiler angle="radian"/><option integrator="implicitfast" cone="elliptic" impratio="{impratio}"/><worldbody>'
        '<geom name="floor" type="plane" size="0 0 0.05"/>'
        f'<body name="cart" pos="0 0 {half[2] + 0.001}"><inertial mass="{m1}" pos="0 0 0" diaginertia="0.01 0.01 0.01"/>'
        f'<joint name="jx" type="slide" axis="1 0 0" armature="{ax}"/>'
        f'<body name="foot" pos="0 0 0"><inertial mass="{m2}" pos="0 0 0" diaginertia="0.01 0.01 0.01"/>'
        f'<joint name="jz" type="slide" axis="0 0 1" armature="{az}" damping="2"/>'
        f'<geom name="foot" type="box" size="{half[0]} {half[1]} {half[2]}" mass="0" friction="1 0.005 0.0001"/></body></body></worldbody>'
        '<actuator><general name="push" joint="jx" gainprm="1" biastype="none"/></actuator></mujoco>')
    m = O.make_model(compile_mjcf(path), True)
    d = O.OrcData()
    O.lib().orc_reset_data(C.byref(m), C.byref(d))
    step(O, m, d, 3000)  # settle on the floor
    d.ctrl[0] = F
    step(O, m, d, 3000)
    K, B = stiffness_damping()
    Mx, Mz = m1 + m2 + ax, m2 + az
    iw = (1 / Mx + 1 / Mz) / 3
    r = brentq(lambda r: 4 * K * impedance(r) ** 2 / (1 - impedance(r)) * r / iw - m2 * G, 1e-9, 1e-2, xtol=1e-18)
    R1 = (1 - impedance(r)) / impedance(r) * iw / impratio
    v = F * R1 / (4 * B)
    assert F < 1.0 * m2 * G  # inside the cone
    assert d.ncon == 4 and abs(d.qpos[1] - (-0.001 - r)) < 1e-11
    assert abs(d.qvel[0] - v) < 1e-12 * max(1.0, abs(v) / 1e-6), (d.qvel[0], v)
    assert 1e-6 < v < 1e-3
