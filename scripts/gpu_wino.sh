#!/bin/bash
# Run on the GPU box (via gpurun): Winograd micro-benchmarks (tools/bin/wino_bench*).
# usage: scripts/gpu_wino.sh <tag>
TAG=${1:-x}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
L=$OUT/wino_$TAG.txt
: > $L
echo "== check" >> $L
timeout 120 tools/bin/wino_bench_a0 256 1 -1 0 1 0 >> $L 2>&1
echo "== check B=8" >> $L
timeout 120 tools/bin/wino_bench_a0 8 1 -1 0 1 0 >> $L 2>&1
echo "== late barrier" >> $L
timeout 120 tools/bin/wino_bench_late 256 0 -1 0 1 0 >> $L 2>&1
echo "== 1 WG per CU" >> $L
WINO_WG_PER_CU=1 timeout 120 tools/bin/wino_bench_a0 256 0 -1 0 1 0 >> $L 2>&1
echo "== again" >> $L
timeout 120 tools/bin/wino_bench_a0 256 0 -1 0 1 0 >> $L 2>&1
grep -v "^odd.*-1 " $L
