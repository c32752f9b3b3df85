"""dev tool (GPU): the 1000-step headline rollout with contacts resolved environment by environment, every environment against its own
oracle (and the oracle's twin nudged by 1e-13 rad: the rollout's own conditioning) -- tests/parity_util.run_headline_resolved_parity,
per-environment report.
    python tools/headline_resolved_probe.py [n_envs] [n_steps] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("robot-control-stack_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np  # noqa: E402
import parity_util as PU  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rep = PU.run_headline_resolved_parity(n_envs=n, n_steps=steps, seed=seed)
touched = rep["first_contact"] >= 0
for e in range(n):
    if rep["err_env"][e] > 1e-9 or rep["flag_env"][e] or rep["overflow_envs"][e] or rep["graze_steps"][e]:
        print(f"env {e}: first contact step {rep['first_contact'][e]}, first step over 1e-9: {rep['first_bad'][e]}, max err {rep['err_env'][e]:.2e} "
              f"(twins part at step {rep['twin_split'][e]}, by {rep['twin_err_env'][e]:.2e} in the end; error before that {rep['excess_env'][e]:.2e}, after {rep['post_split_err'][e]:.2e}), most contacts {rep['max_ncon'][e]}, grazes {rep['graze_steps'][e]}, "
              f"flag mismatches {rep['flag_env'][e]}, overflow {bool(rep['overflow_envs'][e])}")
plain = rep["twin_split"] < 0
print(f"{n} environments x {steps} steps, seed {seed}: {int(touched.sum())} ran into a contact ({int((rep['graze_steps'] > 0).sum())} with a contact that began and "
      f"ended inside one env-step); within 1e-9 of the oracle at every step: {int((rep['err_env'] < 1e-9).sum())}; within 1e-9 at every step before their oracle stops reproducing itself (100 x the twins' distance in that step): "
      f"{int((rep['excess_env'] < 1e-9).sum())}; environments whose twins never part: {int(plain.sum())}, worst error among those {rep['err_env'][plain].max():.2e}; "
      f"flag mismatches {int(rep['flag_env'].sum())}; overflow {int(rep['overflow_envs'].sum())}; on the contact-resolving launch at the end {int(rep['escalated_now'].sum())} "
      f"(most at once {int(rep['escalated_per_step'].max())}); resolved == touched: {bool(np.array_equal(rep['resolved_ever'], rep['contact_steps'] > 0))}")
