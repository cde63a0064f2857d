#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel stats of the bench command + PMC passes (separate runs per counter
# group, --kernel-trace only) -> small text / json summaries under gpurun_out/ (copy the ones to keep into profiles/).
# usage: scripts/gpu_prof.sh <tag> [stats|pmc|all]
TAG=${1:-x}
WHAT=${2:-all}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-overlap --no-graph --no-sustained --no-c2 --no-e2e --no-small-batch --no-split-bf16"
if [ "$WHAT" = "stats" ] || [ "$WHAT" = "all" ]; then
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG -o stats -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-profile --no-overlap --no-graph --no-sustained --no-c2 --no-e2e --no-small-batch --no-split-bf16 > $OUT/rocprof_$TAG.log 2>&1
  for db in $(find $OUT/prof_$TAG -name "*.db"); do python $R/scripts/rocprof_summary.py stats $db > $OUT/${TAG}_rocprof_kernel_stats.txt; done
  rm -rf $OUT/prof_$TAG
  head -14 $OUT/${TAG}_rocprof_kernel_stats.txt
fi
if [ "$WHAT" = "pmc" ] || [ "$WHAT" = "all" ]; then
  rocprofv3 -L > $OUT/counters_avail.txt 2>&1
  pass() { # name, counters...
    local name=$1; shift
    local ok=""
    for c in "$@"; do if grep -qw "$c" $OUT/counters_avail.txt; then ok="$ok $c"; else echo "counter $c not available" >> $OUT/pmc_$TAG.log; fi; done
    [ -z "$ok" ] && return
    timeout 600 rocprofv3 --kernel-trace --pmc $ok -d $OUT/pmc_${TAG}_$name -o pmc -- $BENCH >> $OUT/pmc_$TAG.log 2>&1
  }
  pass sq GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES
  pass fetch FETCH_SIZE
  pass write WRITE_SIZE
  cd $R
  python scripts/rocprof_summary.py traffic $(find $OUT/pmc_${TAG}_fetch $OUT/pmc_${TAG}_write -name "*.db") > $OUT/${TAG}_pmc_traffic.json 2>&1
  # stamp with the hash of the kernel sources these counters were measured on (bench.py: roofline.traffic_stale)
  python - "$OUT/${TAG}_pmc_traffic.json" <<'PY'
import json, sys, time
sys.path.insert(0, '.')
from spec_amd import _lib
d = json.load(open(sys.argv[1]))
d['source_hash'] = _lib.source_hash()
d['command'] = 'scripts/gpu_prof.sh: bench.py --steps 3 --warmup 1 --no-overlap --no-graph under rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes)'
json.dump(d, open(sys.argv[1], 'w'), indent=1)
PY
  cp $OUT/${TAG}_pmc_traffic.json $OUT/pmc_traffic_latest.json
  python - "$OUT/${TAG}_pmc_traffic.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print('igemm traffic/launch', d.get('traffic_bytes_per_launch'))
for k,v in d.get('families',{}).items(): print(k, round(v['read_bytes_per_launch']/1e6,1),'MB read', round(v['write_bytes_per_launch']/1e6,1),'MB write per launch x', v['launches_fetch_pass'])
PY
  : > $OUT/${TAG}_rocprof_pmc_summary.txt
  for d in $OUT/pmc_${TAG}_*/; do
    for db in $(find $d -name "*.db"); do python scripts/rocprof_summary.py pmc $db >> $OUT/${TAG}_rocprof_pmc_summary.txt 2>&1; done
    rm -rf $d
  done
  tail -2 $OUT/pmc_$TAG.log | cut -c1-200
  rm -f $OUT/pmc_$TAG.log $OUT/counters_avail.txt
fi
