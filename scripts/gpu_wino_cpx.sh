cd $GRAFT_REPO_ROOT
echo "== correctness"; timeout 120 tools/bin/wino_bench 256 1 -1 0 1 | grep -v "epilogue split\|per-WG"
for rep in 1 2; do for A in _nc ""; do echo "== bench$A rep $rep"; timeout 120 tools/bin/wino_bench$A 256 0 -1 0 1 | grep -v "epilogue split\|per-WG\|odd"; done; done
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_grouped.py -q -k "winograd or pair" 2>&1 | tail -2
bash scripts/gpu_layer_traffic.sh r03_y 2>&1 | grep "conv2 \|by layer\|layer[234].conv2" | tail -12
