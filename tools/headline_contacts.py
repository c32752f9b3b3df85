"""Which environments of the headline workload (fr3_empty_world, JOINTS, relative +-5 deg LAST_STEP random actions, no resets)
touch the floor or themselves, and when -- on the ORACLE with contacts resolved (what MuJoCo would do).  Test infrastructure.

    python tools/headline_contacts.py [n_envs] [n_steps] [seed]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("robot-control-stack_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np  # noqa: E402
import parity_util as PU  # noqa: E402
import rcs_oracle as O  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    O.DEFAULT_RESOLVE_CONTACTS = True
    oenvs = PU.make_oracle_envs(n, True)
    joints, grip = PU.synthetic_actions(n, steps, seed)
    t0 = time.time()
    first = {}
    kinds = {}
    for e, oe in enumerate(oenvs):
        oe.reset()
        cm = oe.sim.cm
        for t in range(steps):
            oe.step({"joints": joints[t, e], "gripper": grip[t, e]})
            d = oe.sim.s.d
            if d.ncon + d.nself > 0:
                if e not in first:
                    first[e] = t
                for i in range(d.ncon):
                    g = tuple(d.contact_geom[i][:])
                    kinds.setdefault(e, set()).add(("floor/box",) + tuple(cm.geom_names[x] for x in g))
                for i in range(d.nself):
                    g = (d.self_geom[2 * i], d.self_geom[2 * i + 1]) if not hasattr(d.self_geom[0], "__len__") else tuple(d.self_geom[i][:])
                    kinds.setdefault(e, set()).add(("self",) + tuple(cm.geom_names[x] for x in g))
    print(f"{len(first)} / {n} environments in contact within {steps} steps ({time.time() - t0:.1f} s)")
    for e in sorted(first):
        print(e, "first at step", first[e], sorted(kinds[e])[:6])


if __name__ == "__main__":
    main()
