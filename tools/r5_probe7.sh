set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/p16
B="timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5"
run() { name=$1; shift; $B "$@" > gpurun_out/p16/$name.json 2>&1; python - $name <<'P'
import json,sys
g=sys.argv[1]
try:
    l=[x for x in open(f"gpurun_out/p16/{g}.json") if x.startswith("{")][-1]; d=json.loads(l)
    print(g, "%.2f M"%(d["value"]/1e6), "ms", round(d["ms_per_step"],4), "seen", d["config"].get("contacts_seen"))
except Exception as e: print(g, "ERR", e)
P
}
run resolve
run resolve2
run flag --contacts flag
run flag16 --contacts flag --contact-check-every 16
