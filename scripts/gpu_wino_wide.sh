#!/bin/bash
# Run on the GPU box (via gpurun): A/B of the 16-byte (stage pair) patch loads against the 8-byte ones, same box, same minute.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
L=$OUT/wino_wide.txt
: > $L
for V in 0 8 16; do
echo "== correctness, variant $V (wide)" >> $L
timeout 120 tools/bin/wino_bench 8 1 -1 $V 1 >> $L 2>&1
done
for rep in 1 2; do
for A in _nw ""; do
echo "== bench$A rep $rep" >> $L
timeout 120 tools/bin/wino_bench$A 256 0 -1 0 1 >> $L 2>&1
done
done
for A in _nw ""; do
echo "== bench$A variant 16 on layer3" >> $L
timeout 120 tools/bin/wino_bench$A 256 0 6 16 1 >> $L 2>&1
echo "== bench$A variant 8 on layer4" >> $L
timeout 120 tools/bin/wino_bench$A 256 0 7 8 1 >> $L 2>&1
done
grep -v "epilogue split\|per-WG" $L
