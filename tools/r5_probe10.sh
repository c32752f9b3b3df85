set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/p33
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/p33/gputest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/p33/gputest.txt
tail -4 gpurun_out/p33/gputest.txt
B="timeout 600 python bench.py --no-cpu-baseline"
for cfg in "20 5" "300 30" "1000 50"; do set -- $cfg
  $B --steps $1 --warmup $2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('steps $1:', round(d['value']/1e6,2), 'M', round(d['ms_per_step'],4), 'ms seen', d['config']['contacts_seen'])"
done
$B --mode convergence --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('conv 30:', round(d['value']/1e6,3), 'M', round(d['ms_per_step'],3))"
timeout 600 python tools/headline_resolved_probe.py 512 1000 3 > gpurun_out/p33/soak.txt 2>&1; grep -c "first bad step -1" gpurun_out/p33/soak.txt; grep -v "first bad step -1" gpurun_out/p33/soak.txt | cut -c1-90
