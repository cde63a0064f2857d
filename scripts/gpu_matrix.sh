#!/bin/bash
# quick A/B matrix of bench.py flag sets on the GPU box: scripts/gpu_matrix.sh <tag> "<flags1>" "<flags2>" ...
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
i=0
for flags in "$@"; do
  i=$((i+1))
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline $flags > $OUT/mx_${TAG}_$i.json 2> $OUT/mx_${TAG}_$i.err
  python - "$OUT/mx_${TAG}_$i.json" "$flags" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d.get('roofline') or {}
    print(f"{sys.argv[2]:45s} {d['value']:9.1f} img/s {d['ms_per_step']:8.3f} ms  conv {r.get('achieved')} TF/s")
except Exception as e: print(sys.argv[2], 'ERR', e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
done
