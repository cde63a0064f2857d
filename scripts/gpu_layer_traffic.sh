#!/bin/bash
# Run on the GPU box (via gpurun): FETCH_SIZE and WRITE_SIZE passes of the eager single-stream step, summarised PER LAYER
# against the algorithmic bytes of the per-launch table (scripts/rocprof_summary.py layer_traffic).
# usage: scripts/gpu_layer_traffic.sh <tag>
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sustained --no-c2 --no-e2e --no-small-batch --no-split-bf16 > /tmp/lt_bench.json 2>/dev/null
cp $OUT/bench_profile.json /tmp/lt_layers.json
cd /tmp
BENCH="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-overlap --no-graph --no-sustained --no-c2 --no-e2e --no-small-batch --no-split-bf16"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $C -d /tmp/lt_$C -o pmc -- $BENCH > /tmp/lt_$C.log 2>&1
done
cd $R
python scripts/rocprof_summary.py layer_traffic $(find /tmp/lt_FETCH_SIZE -name "*.db") $(find /tmp/lt_WRITE_SIZE -name "*.db") /tmp/lt_layers.json > $OUT/${TAG}_layer_traffic.txt 2>&1
tail -45 $OUT/${TAG}_layer_traffic.txt
