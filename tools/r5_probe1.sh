set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/p1
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/p1/gputest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/p1/gputest.txt
B="timeout 300 python bench.py --no-cpu-baseline"
$B --steps 20 --warmup 5 > gpurun_out/p1/b_drv.json 2> gpurun_out/p1/b_drv.err
HSA_SCRATCH_SINGLE_LIMIT=4000000000 $B --steps 20 --warmup 5 > gpurun_out/p1/b_drv_scr.json 2>&1
$B --steps 20 --warmup 5 --contacts flag > gpurun_out/p1/b_drv_flag.json 2>&1
$B --steps 20 --warmup 5 --contacts flag --contact-check-every 16 > gpurun_out/p1/b_drv_flag16.json 2>&1
$B --steps 300 --warmup 30 > gpurun_out/p1/b_300.json 2>&1
$B --steps 1000 --warmup 50 > gpurun_out/p1/b_1000.json 2>&1
timeout 300 python tools/grasp_bench.py 4096 > gpurun_out/p1/grasp_fr3.txt 2>&1
timeout 300 python tools/grasp_bench.py 4096 --xarm7 > gpurun_out/p1/grasp_xarm7.txt 2>&1
tail -3 gpurun_out/p1/gputest.txt
for f in b_drv b_drv_scr b_drv_flag b_drv_flag16 b_300 b_1000; do python - "$f" <<'P'
import json,sys
f=sys.argv[1]
try:
    l=[x for x in open(f"gpurun_out/p1/{f}.json") if x.startswith("{")][-1]; d=json.loads(l)
    print(f, "%.2f M"%(d["value"]/1e6), "ms", round(d["ms_per_step"],4), "seen", d["config"].get("contacts_seen"), "kernel_ms", d["roofline"].get("kernel_ms_avg"))
except Exception as e: print(f, "ERR", e)
P
done
tail -8 gpurun_out/p1/grasp_fr3.txt; tail -8 gpurun_out/p1/grasp_xarm7.txt
