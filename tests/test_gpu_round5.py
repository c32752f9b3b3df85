"""GPU tests added in round 5 (through the C-ABI; the oracle is the checker)."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_self_contact_rows_match_oracle():
    """Verdict r4, next 1a: robot <-> robot contacts carry constraint rows (J = G (S_B - S_A): Hessian terms over the joints between
    the two links, noslip cross blocks, five links in contact).  Folded arms whose fingers / hand come to rest ON link 1, stepped
    by Sim.step(17) launches: joint positions <= 1e-9, velocities <= 1e-8 against the oracle that resolves the same contacts --
    environment by environment (the lean kernel for the others) and with the whole batch on the contact-resolving kernel."""
    from parity_util import run_self_contact_parity

    rep = run_self_contact_parity(n_envs=24, seed=1, launches=40, mode=7)
    assert rep["max_abs_qpos"] < 1e-9 and rep["max_abs_qvel"] < 1e-8, rep
    assert rep["touched"].sum() >= 4 and rep["self_contact_substeps"] > 1000 and rep["max_contacts"] >= 4, rep
    assert np.array_equal(rep["resolved_ever"], rep["touched"]), rep           # exactly the environments the oracle saw contacts in
    assert (rep["escalated_now"] >= rep["in_contact_at_end"]).all(), rep       # whoever is in contact is on the contact-resolving kernel
    assert (rep["tracking_error"][rep["in_contact_at_end"]] > 0.02).all(), rep  # link 1 is in the way: the servo does not reach its target
    assert rep["overflow"] == 0, rep
    rep = run_self_contact_parity(n_envs=16, seed=1, launches=30, mode=3)
    assert rep["max_abs_qpos"] < 1e-9 and rep["max_abs_qvel"] < 1e-8 and rep["touched"].sum() >= 2, rep


def test_headline_rollout_matches_the_resolving_oracle_for_1000_steps():
    """Verdict r5, next 1: the headline workload for the WHOLE of BASELINE.md's rollout (1000 env-steps, no resets): 64 environments,
    EVERY one held to its own oracle instance that resolves floor AND self contacts (no bound on its contact list), every step: joints
    1e-9, velocities 1e-8, flags bit-equal.  Nobody is excluded: a contact that begins and ends inside one launch is caught by the
    certifying check (csrc/check_team.h) and the launch redone with its contacts resolved; nobody overflows a contact phase."""
    from parity_util import run_headline_resolved_parity

    rep = run_headline_resolved_parity(n_envs=64, n_steps=1000, seed=0)
    touched = rep["first_contact"] >= 0
    assert touched.sum() >= 8, rep  # (some environments do reach the floor / themselves)
    assert rep["overflow_envs"].sum() == 0 and not rep["unresolved"].any(), rep
    # the bars: 1e-9 / 1e-8 and flags bit-equal at every step before an environment's ORACLE stops reproducing itself (its twin, nudged by
    # 1e-13 rad, parts from it: shut fingers pressed into each other -- parity_util.run_headline_resolved_parity); 100 x the twins'
    # distance in that step; nearly all environments never get there
    assert rep["excess_env"].max() < 1e-9 and rep["vexcess_env"].max() < 1e-8 and rep["flag_env"].sum() == 0, rep
    assert (rep["twin_split"] < 0).sum() >= 56 and rep["err_env"][rep["twin_split"] < 0].max() < 1e-9, rep  # (the plain bars, start to end)
    assert np.array_equal(rep["resolved_ever"], rep["contact_steps"] > 0), rep  # resolved: exactly the environments the oracle saw contacts in
    # de-escalation: environments go back to the lean launch when they have moved clear (sticky until reset through round 5)
    assert (rep["resolved_ever"] & ~rep["escalated_now"]).sum() >= 1, rep


def test_split_contact_resolving_launch_gives_the_same_rollout(monkeypatch):
    """The contact-resolving launch in two parts (RunOp::esc_part; RCSH_ESC_SPLIT=1, off by default because it measures slower:
    profiles/r5_v2/split_ab.txt): already escalated environments on a stream of their own beside the lean launch, the newly flagged ones
    behind both, the form chosen per step from the hint the lean launch writes to host memory.  Same bars as the default form, over the
    700 steps in which the first environments of this seed run into a contact and stay on the contact-resolving launch."""
    from parity_util import run_headline_resolved_parity

    monkeypatch.setenv("RCSH_ESC_SPLIT", "1")
    monkeypatch.setenv("RCSH_ESC_SPLIT_MAX", "4096")
    rep = run_headline_resolved_parity(n_envs=64, n_steps=700, seed=0)
    touched = rep["first_contact"] >= 0
    assert touched.sum() >= 3, rep
    assert rep["max_abs_qpos"] < 1e-9 and rep["max_abs_qvel"] < 1e-8 and rep["flag_mismatches"] == 0, rep
    assert not rep["unresolved"].any(), rep


def test_until_convergence_false_alarms_of_the_certificate_stay_bounded():
    """`step_until_convergence` launches are up to 500 substeps long and move a joint by up to five degrees: the certificate's path-length
    margins are at their largest there.  A regression guard for what round 6 found (1900 of 4096 environments redone with contact phases
    WITHOUT ever meeting a contact: the fingertips' adjacent pads, and link 5's hull against the flange's charged the hand's lever):
    positions and substep counts still equal the resolving oracle's, and the environments on the contact-resolving launch that never met a
    contact stay under a fifth of the batch (a ninth, measured)."""
    from parity_util import make_oracle_envs, make_vec_env, synthetic_actions

    n, steps = 256, 5
    venv = make_vec_env(n, False)
    oenvs = make_oracle_envs(16, False)
    joints, grip = synthetic_actions(n, steps, 0)
    venv.reset()
    for oe in oenvs:
        oe.reset()
    held = []
    for t in range(steps):
        _, _, _, _, info = venv.step({"joints": joints[t], "gripper": grip[t]})
        now, ever = venv.sim.contact_escalated()
        held.append(float((np.asarray(now, bool) & ~np.asarray(ever, bool)).mean()))
        q = venv.sim.qpos
        for e, oe in enumerate(oenvs):
            oe.step({"joints": joints[t, e], "gripper": grip[t, e]})
            assert np.abs(q[e][:7] - oe.sim.qpos[:7]).max() < 1e-9, (t, e)
            assert int(info["substeps"][e]) == int(oe.sim.s.convergence_steps), (t, e)
    assert max(held[1:]) < 0.2, held


def test_until_convergence_in_pieces_equals_one_launch_bit_for_bit(monkeypatch):
    """`step_until_convergence` with contacts resolved per environment runs in pieces of `RCSH_CONV_CHUNK` substeps (csrc/rcs_hip.hip:
    launch_run -- each piece certified over its own travel, an environment that fails redone for that piece only): the pieces carry the
    substep count, the callbacks' verdicts and the cap over, so positions, velocities, substep counts and flags equal the one-launch
    form's to the last bit."""
    from parity_util import make_vec_env, synthetic_actions

    n, steps = 64, 4
    joints, grip = synthetic_actions(n, steps, 5)
    out = []
    for chunk in ("0", "48", "16"):
        monkeypatch.setenv("RCSH_CONV_CHUNK", chunk)
        venv = make_vec_env(n, False)
        venv.reset()
        rec = []
        for t in range(steps):
            obs, _, _, trunc, info = venv.step({"joints": joints[t], "gripper": grip[t]})
            rec.append((np.array(venv.sim.qpos), np.array(venv.sim.qvel), np.array(info["substeps"]), np.array(info["is_sim_converged"]), np.array(info["collision"]), np.array(obs["joints"])))
        out.append(rec)
        venv.close()
    for other in out[1:]:
        for a, b in zip(out[0], other):
            for x, y in zip(a, b):
                assert np.array_equal(x, y)
