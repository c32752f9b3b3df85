"""Closed forms for the parts of the CPU oracle's physics that no mechanics derivation covers: MuJoCo's SOFT-CONSTRAINT model
(joint limit rows, joint-equality rows, dry-friction rows) and the implicitfast integrator, on one- and two-dof models small
enough for pencil and paper.  Each expected value below is worked out from the model MuJoCo documents ("Computation":
solref -> (K, B), solimp -> impedance d(r), R = (1 - d)/d . diagApprox, force = -D (J qacc - aref), D = 1/R) -- not from the
oracle -- and the oracle (oracle/rcs_physics.c, which the HIP kernels are held to at 1e-9) has to reproduce it.

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
