set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/p14
B="timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 3"
run() { name=$1; shift; env "$@" $B > gpurun_out/p14/$name.json 2>&1; python - $name <<'P'
import json,sys
g=sys.argv[1]
try:
    l=[x for x in open(f"gpurun_out/p14/{g}.json") if x.startswith("{")][-1]; d=json.loads(l)
    print(g, "%.2f M"%(d["value"]/1e6), "ms", round(d["ms_per_step"],4), "seen", d["config"].get("contacts_seen"))
except Exception as e: print(g, "ERR", e)
P
}
run base A=1
run base2 A=1
run nosnap RCSH_ESC_MEASURE_NO_SNAP=1
run nocheck RCSH_ESC_MEASURE_NO_CHECK=1
run neither RCSH_ESC_MEASURE_NO_SNAP=1 RCSH_ESC_MEASURE_NO_CHECK=1
