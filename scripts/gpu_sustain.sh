#!/bin/bash
# Sustained-load evidence on the GPU box: the bench step for >= 30 s while rocm-smi samples clocks / power / temperature.
TAG=${1:-x}
SECS=${2:-30}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
( while true; do echo "--- $(date +%s.%N)"; rocm-smi --showclocks --showpower --showtemp --showuse 2>/dev/null | grep -E "sclk|mclk|Power|Temperature|GPU use" ; sleep 2; done ) > $OUT/${TAG}_smi_samples.txt 2>&1 &
SMI=$!
timeout 900 python bench.py --steps 20 --warmup 3 --sustained-seconds $SECS --no-cpu-baseline --no-c2 --no-small-batch > $OUT/${TAG}_sustained_bench.json 2> $OUT/${TAG}_sustained_bench.err
kill $SMI 2>/dev/null
python - "$OUT/${TAG}_sustained_bench.json" "$OUT/${TAG}_smi_samples.txt" <<'PY'
import json,sys,re
d=json.load(open(sys.argv[1])); print('burst', d['value'], 'img/s;', 'sustained', d['sustained'])
txt=open(sys.argv[2]).read()
sclk=[int(x) for x in re.findall(r'sclk clock level:? \d+:? \((\d+)Mhz\)', txt)]
pw=[float(x) for x in re.findall(r'Power \(W\): ([\d.]+)', txt)]
tm=[float(x) for x in re.findall(r'Temperature \(Sensor (?:junction|edge)\) \(C\): ([\d.]+)', txt)]
use=[int(x) for x in re.findall(r'GPU use \(%\): (\d+)', txt)]
print('samples', txt.count('--- '), 'sclk MHz min/max', (min(sclk), max(sclk)) if sclk else None, 'power W max', max(pw) if pw else None, 'temp C max', max(tm) if tm else None, 'gpu use % max', max(use) if use else None)
PY
tail -30 $OUT/${TAG}_smi_samples.txt | head -40
