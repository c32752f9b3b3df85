set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/p21
B="timeout 600 python bench.py --no-cpu-baseline"
run() { name=$1; shift; "$@" > gpurun_out/p21/$name.json 2>&1; python - $name <<'P'
import json,sys
g=sys.argv[1]
try:
    l=[x for x in open(f"gpurun_out/p21/{g}.json") if x.startswith("{")][-1]; d=json.loads(l)
    print(g, "%.2f M"%(d["value"]/1e6), "ms", round(d["ms_per_step"],4), "seen", d["config"].get("contacts_seen"))
except Exception as e: print(g, "ERR", e)
P
}
run drv_split $B --steps 20 --warmup 5
run drv_nosplit env RCSH_ESC_SPLIT=0 $B --steps 20 --warmup 5
run s300_split $B --steps 300 --warmup 30
run s300_nosplit env RCSH_ESC_SPLIT=0 $B --steps 300 --warmup 30
run s1000_split $B --steps 1000 --warmup 50
run s1000_nosplit env RCSH_ESC_SPLIT=0 $B --steps 1000 --warmup 50
