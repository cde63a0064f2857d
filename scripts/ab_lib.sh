#!/bin/bash
# A/B of two builds of libspecmi.so on one box, alternating: scripts/ab_lib.sh <alternative .so> [reps]
ALT=$1; REPS=${2:-3}
cd ${GRAFT_REPO_ROOT:-$(pwd)}
S="--no-cpu-baseline --no-split-bf16 --no-c2 --no-small-batch --no-sustained --no-e2e --steps 20"
pe() { python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
st={r['stage']:r['ms'] for r in d.get('stages',[])}
print(sys.argv[2], 'ms_per_step', d['ms_per_step'], 'value', d['value'], {k:st.get(k) for k in ('layer4.conv3','layer4.conv3+downsample','layer4.conv1')})" $1 "$2"; }
for r in $(seq $REPS); do
  SPECMI_LIB=$(pwd)/$ALT python bench.py $S > /tmp/ab_a.json 2>/dev/null; pe /tmp/ab_a.json "alt  $r:"
  python bench.py $S > /tmp/ab_b.json 2>/dev/null; pe /tmp/ab_b.json "head $r:"
done
