#!/bin/bash
# tools/bin/igemm_bench (SPECMI_TUNE build of conv_igemm.hip): per-layer time of the tile variants of the throughput kernel at a small batch
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT
for B in ${1:-8 16}; do for v in 0 2 11 10 12; do echo "=== B=$B variant=$v"; timeout 120 $R/tools/bin/igemm_bench $v $B 0 2>&1 | grep -E " ms |ms$|TF" | cut -c1-150; done; done > $OUT/igemm_variants.txt 2>&1
head -60 $OUT/igemm_variants.txt
