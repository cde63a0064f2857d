#!/bin/bash
# PMC passes on the GPU box (separate runs per counter group; no trace domains besides kernel-trace).
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > $OUT/counters_avail.txt 2>&1
BENCH="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-overlap --no-graph"
pass() { # name, counters...
  local name=$1; shift
  local ok=""
  for c in "$@"; do if grep -qw "$c" $OUT/counters_avail.txt; then ok="$ok $c"; else echo "counter $c not available" >> $OUT/pmc_$TAG.log; fi; done
  [ -z "$ok" ] && return
  timeout 600 rocprofv3 --kernel-trace --pmc $ok -d $OUT/pmc_${TAG}_$name -o pmc -- $BENCH >> $OUT/pmc_$TAG.log 2>&1
}
pass sq GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES
pass sq2 GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
cd $R
python scripts/rocprof_summary.py traffic $(find $OUT/pmc_${TAG}_fetch $OUT/pmc_${TAG}_write -name "*.db") > $OUT/pmc_${TAG}_traffic.json 2>&1
cat $OUT/pmc_${TAG}_traffic.json
: > $OUT/pmc_${TAG}_summary.txt
for d in $OUT/pmc_${TAG}_*/; do
  for db in $(find $d -name "*.db"); do python scripts/rocprof_summary.py pmc $db >> $OUT/pmc_${TAG}_summary.txt 2>&1; done
  rm -rf $d
done
grep -E "GUI_ACTIVE|MFMA|FETCH|WRITE|TCC|WAIT|conv_igemm|conv_wino|stem" $OUT/pmc_${TAG}_summary.txt | head -60
tail -2 $OUT/pmc_$TAG.log | cut -c1-200
rm -f $OUT/pmc_$TAG.log
head -c 200000 $OUT/counters_avail.txt > $OUT/counters_avail_head.txt; rm -f $OUT/counters_avail.txt
