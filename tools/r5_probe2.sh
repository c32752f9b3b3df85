set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/p3
B="timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 3"
for g in 1024 256 64 8; do
  RCSH_ESC_GRID_PROBE=$g $B > gpurun_out/p3/b_g$g.json 2>&1
  python - $g <<'P'
import json,sys
g=sys.argv[1]
l=[x for x in open(f"gpurun_out/p3/b_g{g}.json") if x.startswith("{")][-1]; d=json.loads(l)
print("grid", g, "%.2f M"%(d["value"]/1e6), "ms", round(d["ms_per_step"],4), "seen", d["config"].get("contacts_seen"))
P
done
