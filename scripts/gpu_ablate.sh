#!/bin/bash
# Runtime ablations of the conv_igemm kernel on representative layers (wrong results, timing only): which part of the
# chunk costs what.  bits: 1 no A global loads in the loop, 2 no LDS restage, 4 no epilogue stores, 16 / 32 prologue parts.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=${1:-x}
: > $OUT/${TAG}_igemm_ablate.txt
for layer in 0 2 3 7 8 13 14 19 20; do
  for ab in 0 1 2 3 4 7; do
    echo "--- layer $layer ablate $ab" >> $OUT/${TAG}_igemm_ablate.txt
    timeout 60 $R/tools/bin/igemm_bench 0 256 0 $ab $layer 2>&1 | grep -E "ms|cycles" >> $OUT/${TAG}_igemm_ablate.txt
  done
done
python - "$OUT/${TAG}_igemm_ablate.txt" <<'PY'
import sys,re
cur=None; rows={}
for line in open(sys.argv[1]):
    m=re.match(r'--- layer (\d+) ablate (\d+)',line)
    if m: cur=(int(m.group(1)),int(m.group(2))); continue
    m=re.match(r'(.{34})\s+([\d.]+) ms',line)
    if m and cur: rows.setdefault(cur[0],{'name':m.group(1).strip()})[cur[1]]=float(m.group(2))
print('%-34s'%'layer'+''.join('%9s'%('abl%d'%a) for a in (0,1,2,3,4,7)))
for k,r in rows.items():
    print('%-34s'%r['name']+''.join('%9.3f'%r.get(a,float('nan')) for a in (0,1,2,3,4,7)))
PY
