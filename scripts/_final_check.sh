cd $GRAFT_REPO_ROOT
timeout 300 python scripts/gpu_plan_crossover.py 10,11,12,14,16 2>&1 | grep batch | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print(d['batch'], 'auto', d['auto'], 'lat2s', d['latency_two_streams'], 'thr_grp', d['throughput_grouped'])"
timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-profile --no-small-batch --no-e2e --no-c2 --no-split-bf16 --no-sustained 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value', d['value'], 'single_frame', (d.get('e2e_demo') or {}).get('single_frame'))"
