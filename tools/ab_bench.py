"""A/B two builds of librcs_hip.so on the same GPU box: python tools/ab_bench.py <other.so> [bench.py args...]

Runs bench.py alternately with the in-tree library and with <other.so> (a build of another revision placed beside
it), three rounds, and prints ms/step of each.  Box-to-box clock differences are larger than most single kernel
changes; only same-box pairs are comparable."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
other = os.path.abspath(sys.argv[1])
args = sys.argv[2:] or ["--steps", "300", "--warmup", "30"]
code = ("import sys, os; sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'robot-control-stack_amd'));"
        "from rcs_amd import _lib; _lib.LIB_PATH = os.environ.get('AB_LIB', _lib.LIB_PATH);"
        "import bench; sys.argv = ['bench.py', '--no-cpu-baseline'] + %r; bench.main()") % (ROOT, ROOT, args)
for rnd in range(3):
    for name, path in (("tree", None), ("other", other)):
        env = dict(os.environ)
        if path:
            env["AB_LIB"] = path
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=ROOT)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(name, "FAILED", out.stderr[-500:])
            continue
        j = json.loads(line[-1])
        print(f"round {rnd} {name:5s} ms_per_step {j['ms_per_step']:.5f} kernel_ms {j['roofline']['kernel_ms_avg']:.5f} value {j['value']:.4g}")
