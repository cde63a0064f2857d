#!/bin/bash
# Round-end evidence at HEAD (one gpurun call): GPU tests, parity report, bench line, rocprof stats + PMC traffic of the bench command,
# batch-1 / batch-8 timelines.  usage: scripts/gpu_final.sh <tag>
TAG=${1:-r06_z}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
(timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -12) > $OUT/${TAG}_gpu_tests.log 2>&1
tail -3 $OUT/${TAG}_gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/${TAG}_smoke.log 2>&1; tail -1 $OUT/${TAG}_smoke.log
timeout 600 python tests/parity_report.py > $OUT/${TAG}_parity_report.txt 2>&1; tail -3 $OUT/${TAG}_parity_report.txt
timeout 600 python scripts/gpu_pretrained_like_report.py > $OUT/${TAG}_parity_report_pretrained_like.txt 2>/dev/null; tail -8 $OUT/${TAG}_parity_report_pretrained_like.txt | cut -c1-200
bash scripts/gpu_prof.sh $TAG all > $OUT/${TAG}_prof.log 2>&1; tail -6 $OUT/${TAG}_prof.log
cd $R
timeout 600 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; tail -c 900 $OUT/${TAG}_bench.json; echo
cp $OUT/bench_profile.json $OUT/${TAG}_layers.json 2>/dev/null
bash scripts/gpu_small_batch_evidence.sh $TAG > $OUT/${TAG}_evidence.log 2>&1; grep "^# kernels" $OUT/${TAG}_evidence.log
