cd "$GRAFT_REPO_ROOT"
for s in 0 1 2 4; do echo "skip=$s: $(RCSH_CHECK_SKIP=$s python bench.py --no-cpu-baseline --steps 300 --warmup 30 --contact-check-every 1 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["ms_per_step"], d["value"]/1e6)')"; done
echo "off: $(python bench.py --no-cpu-baseline --steps 300 --warmup 30 --contact-check-every 0 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["ms_per_step"], d["value"]/1e6)')"
