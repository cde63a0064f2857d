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
for A in "" _a1 _a2 _a3; do
for W in 1 2; do
echo "== bench$A  WG per CU $W" >> $L
WINO_WG_PER_CU=$W timeout 120 tools/bin/wino_bench$A 256 0 -1 0 1 >> $L 2>&1
done
done
grep -v "^odd" $L
