#!/bin/bash
# Per-layer sweep of the conv_igemm tile variants on the GPU box (tools/igemm_bench, built with SPECMI_TUNE).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
TAG=${1:-x}
: > $OUT/${TAG}_igemm_sweep.txt
for v in 0 3 4 9 10 11 12 8 1 2; do
  echo "=== variant $v" >> $OUT/${TAG}_igemm_sweep.txt
  timeout 120 $R/tools/bin/igemm_bench $v 256 0 >> $OUT/${TAG}_igemm_sweep.txt 2>&1
done
python - "$OUT/${TAG}_igemm_sweep.txt" <<'PY'
import sys,re
rows={}; v=None; order=[]
for line in open(sys.argv[1]):
    m=re.match(r'=== variant (\d+)',line)
    if m: v=int(m.group(1)); continue
    m=re.match(r'(.{34})\s+([\d.]+) ms\s+([\d.]+) TF/s',line)
    if m:
        name=m.group(1).strip()
        if name not in rows: rows[name]={}; order.append(name)
        rows[name][v]=float(m.group(2))
vs=sorted({k for r in rows.values() for k in r})
print('%-34s'%'layer'+''.join('%8s'%('v%d'%k) for k in vs)+'   best')
for n in order:
    r=rows[n]; b=min(r,key=r.get)
    print('%-34s'%n+''.join('%8.3f'%r.get(k,float('nan')) for k in vs)+'   v%d (auto v0 %.3f)'%(b,r.get(0,float('nan'))))
PY
