set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/p28
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/p28/gputest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/p28/gputest.txt
tail -4 gpurun_out/p28/gputest.txt
B="timeout 600 python bench.py --no-cpu-baseline"
r() { name=$1; shift; $B "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name:', round(d['value']/1e6,2), 'M', round(d['ms_per_step'],4), 'ms seen', d['config'].get('contacts_seen'))"; }
r drv --steps 20 --warmup 5
r drv_flag --steps 20 --warmup 5 --contacts flag
r s300 --steps 300 --warmup 30
r s300_flag --steps 300 --warmup 30 --contacts flag
r s1000 --steps 1000 --warmup 50
r xarm7_pick --steps 200 --warmup 20 --robot xarm7_pick
r pick_up --steps 200 --warmup 20 --task pick_up
r conv30 --mode convergence --steps 30 --warmup 5
r conv30_flag --mode convergence --steps 30 --warmup 5 --contacts flag
r conv300_flag --mode convergence --steps 300 --warmup 5 --contacts flag
