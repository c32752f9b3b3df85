set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/p17
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/p17/gputest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/p17/gputest.txt
B="timeout 600 python bench.py --no-cpu-baseline"
$B --steps 20 --warmup 5 > gpurun_out/p17/b_drv.json 2>&1
$B --steps 300 --warmup 30 > gpurun_out/p17/b_300.json 2>&1
$B --steps 1000 --warmup 50 > gpurun_out/p17/b_1000.json 2>&1
tail -5 gpurun_out/p17/gputest.txt
for f in b_drv b_300 b_1000; do python - "$f" <<'P'
import json,sys
f=sys.argv[1]
try:
    l=[x for x in open(f"gpurun_out/p17/{f}.json") if x.startswith("{")][-1]; d=json.loads(l)
    print(f, "%.2f M"%(d["value"]/1e6), "ms", round(d["ms_per_step"],4), "seen", d["config"].get("contacts_seen"), "kernel_ms", d["roofline"].get("kernel_ms_avg"))
except Exception as e: print(f, "ERR", e)
P
done
