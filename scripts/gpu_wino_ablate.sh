#!/bin/bash
# Round 4: compile-time ablations of the Winograd kernel that bound what a restructured wave layout could buy
# (tools/bin/wino_bench_ab<bits>: 8 no A-fragment reads, 4 no input transform, 12 both, 13 + no U loads, 15 + no patch loads).
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
L=$OUT/wino_ablate_$TAG.txt; : > $L
for rep in 1 2; do
for A in "" _ab8 _ab4 _ab12 _ab13 _ab15; do
  echo "== rep $rep wino_bench$A (B=256, two workgroups per CU)" >> $L
  timeout 120 tools/bin/wino_bench$A 256 0 -1 0 1 2>&1 | grep -v "^odd\|epilogue split\|per-WG" >> $L
done
done
cat $L
