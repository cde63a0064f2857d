#!/bin/bash
# GPU box: per-layer profiles (HIP events) and rocprofv3 timelines of the graph-replayed step at batch 1 and 8 -> gpurun_out/
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
for b in 1 8; do timeout 300 python scripts/profile_small_batch.py --batch $b 2>/dev/null > $OUT/${TAG}_b${b}_profile.txt; done
export TMPDIR=/tmp; cd /tmp
for b in 1 8; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace_${TAG}_b$b -o tr -- python $R/scripts/small_batch_trace.py --batch $b --reps 50 > $OUT/trace_${TAG}_b$b.log 2>&1
  for db in $(find $OUT/trace_${TAG}_b$b -name "*.db"); do
    # one step = 58 dispatches at batch 1 (grouped launches), 108 at batch 8 (two trunks on two streams): show a whole step + the tail of the previous one
    n=62; [ "$b" = "8" ] && n=112
    python $R/scripts/rocprof_summary.py timeline $db $n > $OUT/${TAG}_b${b}_timeline_rocprof.txt 2>&1
    python $R/scripts/rocprof_summary.py stats $db > $OUT/${TAG}_b${b}_rocprof_kernel_stats.txt 2>&1
  done
  rm -rf $OUT/trace_${TAG}_b$b $OUT/trace_${TAG}_b$b.log
  tail -1 $OUT/${TAG}_b${b}_timeline_rocprof.txt; head -12 $OUT/${TAG}_b${b}_rocprof_kernel_stats.txt
done
