"""dev tool (GPU): the end-of-launch contact check against the resolving oracle on more environments / seeds than the suite runs
(tests/test_gpu_round4.py::test_headline_no_contacts_is_a_checked_property_over_1000_steps): first-flag env-step equal in every environment."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "robot-control-stack_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np
import parity_util as pu
for n, seed in ((512, 1), (256, 2), (256, 3)):
    t0 = time.time()
    rep = pu.run_headline_contact_check(n_envs=n, n_steps=1000, seed=seed)
    same = bool(np.array_equal(rep["first_kernel"], rep["first_oracle"]))
    print(f"n {n} seed {seed}: flagged (oracle) {rep['flagged_oracle']}, first-flag steps equal {same}, flag mismatch steps {rep['flag_mismatch_steps']}, "
          f"unflagged max |dq| {rep['max_abs_qpos_unflagged']:.2e} |dv| {rep['max_abs_qvel_unflagged']:.2e}, vs the non-resolving oracle over the whole rollout {rep['max_abs_qpos_lean']:.2e}; "
          f"flags before any contact {rep['flag_before_any_contact']}, environments with a contact inside a launch before their flag {rep['transient_before_flag']}  ({time.time() - t0:.0f} s)", flush=True)
    if not same:
        bad = np.nonzero(np.asarray(rep["first_kernel"]) != np.asarray(rep["first_oracle"]))[0]
        print("   differing environments", bad[:10], np.asarray(rep["first_kernel"])[bad[:10]], np.asarray(rep["first_oracle"])[bad[:10]])
