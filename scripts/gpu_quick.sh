#!/bin/bash
# Run on the GPU box (via gpurun): parity tests (optional) + the default bench line.
# usage: scripts/gpu_quick.sh <tag> [tests|notests] [pytest -k expression]
TAG=${1:-x}
MODE=${2:-tests}
KEXPR=${3:-}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
if [ "$MODE" = "tests" ]; then
  if [ -n "$KEXPR" ]; then
    (timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "$KEXPR" 2>&1 | tail -80) > $OUT/test_$TAG.log 2>&1
  else
    (timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=8 2>&1 | tail -80) > $OUT/test_$TAG.log 2>&1
  fi
  tail -30 $OUT/test_$TAG.log
fi
timeout 600 python bench.py --steps 20 --warmup 3 > $OUT/bench_${TAG}.json 2> $OUT/bench_${TAG}.err
cp $OUT/bench_profile.json $OUT/bench_${TAG}_layers.json 2>/dev/null
python - "$OUT/bench_${TAG}.json" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print('VALUE', d['value'], 'img/s', d['ms_per_step'],'ms | roof', d['roofline'] and (d['roofline']['achieved'], d['roofline']['frac']),
          '| sustained', d.get('sustained'), '| c2', d.get('c2') and (d['c2']['ms_per_step'], d['c2']['images_per_s']))
    print('CPU', d.get('cpu_baseline'))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
grep -A40 "per-stage" $OUT/bench_${TAG}.err | head -45
tail -5 $OUT/bench_${TAG}.err
