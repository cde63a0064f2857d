#!/bin/bash
# GPU box: latency-plan tests, knob sweep, per-layer small-batch profiles.  usage: scripts/gpu_latency.sh <tag> [sweep args]
TAG=${1:-x}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
(timeout 900 python -m pytest tests/test_gpu_latency.py tests/test_gpu_grouped.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -40) > $OUT/lat_test_$TAG.log 2>&1
tail -25 $OUT/lat_test_$TAG.log
timeout 900 python scripts/latency_sweep.py "$@" > $OUT/lat_sweep_$TAG.log 2>&1
cp $OUT/latency_sweep.jsonl $OUT/lat_sweep_$TAG.jsonl 2>/dev/null
grep "^{" $OUT/lat_sweep_$TAG.log
for b in 1 8; do timeout 300 python scripts/profile_small_batch.py --batch $b > $OUT/lat_prof_b${b}_$TAG.txt 2>&1; head -45 $OUT/lat_prof_b${b}_$TAG.txt; done
export TMPDIR=/tmp; cd /tmp
for b in 1 8; do
  timeout 300 rocprofv3 --kernel-trace -d $OUT/trace_${TAG}_b$b -o tr -- python $R/scripts/small_batch_trace.py --batch $b > $OUT/trace_${TAG}_b$b.log 2>&1
  for db in $(find $OUT/trace_${TAG}_b$b -name "*.db"); do python $R/scripts/rocprof_summary.py timeline $db 62 > $OUT/lat_timeline_b${b}_$TAG.txt 2>&1; done
  rm -rf $OUT/trace_${TAG}_b$b
  cat $OUT/lat_timeline_b${b}_$TAG.txt
done
