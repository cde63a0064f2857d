#!/bin/bash
# Run on the GPU box (via gpurun): parity tests, bench variants, rocprofv3 kernel stats.
# usage: scripts/gpu_check.sh <tag> [tests|notests]
TAG=${1:-x}
MODE=${2:-tests}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
if [ "$MODE" = "tests" ]; then
  (timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -60) > $OUT/test_$TAG.log 2>&1
  tail -4 $OUT/test_$TAG.log
fi
timeout 300 python bench.py --steps 10 --warmup 3 --no-overlap --no-graph --no-cpu-baseline > $OUT/bench_${TAG}_serial.json 2> $OUT/bench_${TAG}_serial.err
cp $OUT/bench_profile.json $OUT/bench_${TAG}_layers.json 2>/dev/null
timeout 300 python bench.py --steps 10 --warmup 3 > $OUT/bench_${TAG}.json 2> $OUT/bench_${TAG}.err
for v in 2 3; do
  timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-overlap --force-variant $v > $OUT/bench_${TAG}_v$v.json 2> $OUT/bench_${TAG}_v$v.err
done
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG -o stats -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-profile --no-overlap --no-graph > $OUT/rocprof_$TAG.log 2>&1
cd $R
for db in $(find $OUT/prof_$TAG -name "*.db"); do python scripts/rocprof_summary.py stats $db > $OUT/rocprof_${TAG}_kernel_stats.txt; done
rm -rf $OUT/prof_$TAG; head -8 $OUT/rocprof_${TAG}_kernel_stats.txt
for f in $OUT/bench_${TAG}_serial.json $OUT/bench_${TAG}.json $OUT/bench_${TAG}_v2.json $OUT/bench_${TAG}_v3.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], d['value'], 'img/s', d['ms_per_step'],'ms', 'roof', d['roofline'] and d['roofline']['achieved'], 'cpu', d.get('cpu_baseline'))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
tail -22 $OUT/bench_${TAG}_serial.err
