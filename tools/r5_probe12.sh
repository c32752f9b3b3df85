set -u
cd "$GRAFT_REPO_ROOT"
for mode in 1 0; do
  echo "== RCSH_COPY_CARRIER_GRAPH=$mode"
  RCSH_COPY_CARRIER_GRAPH=$mode timeout -k 5 150 python -c "
import sys; sys.path[:0]=['tests','robot-control-stack_amd']
import test_gpu_sharding as t
t._run_exchange(2, carrier='copy', same_device=True, n_steps=12)
t._run_exchange(1, carrier='copy')
print('exchange tests ok')
" 2>&1 | grep -v "RCCL\|HIP version\|ROCm\|Hostname\|Librccl" | tail -4
  RCSH_COPY_CARRIER_GRAPH=$mode timeout -k 5 200 python bench.py --no-cpu-baseline --steps 40 --warmup 5 --gpus 2 --dist-backend sdma --envs 2048 2>/dev/null | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); c=d['config']; print('bench %.2f M'%(d['value']/1e6), round(d['ms_per_step'],4), 'exchange_ms', c.get('exchange_ms'), 'without', c.get('value_without_exchange'))
except Exception as e: print('bench failed', e)
"
done
