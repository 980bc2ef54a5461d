L='64,128,64,64;32,128,64,64'
for i in 1 2; do
echo "== x32"; timeout 100 python tools/bench_conv.py --layers "$L" 2>&1 | grep probe
echo "== 16x16 form"; IMM_CONV_DISABLE=halo2x timeout 100 python tools/bench_conv.py --layers "$L" 2>&1 | grep probe
done
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "conv_forward or dgrad" 2>&1 | grep -E "passed|failed" 
