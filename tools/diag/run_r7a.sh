mkdir -p gpurun_out/r7a
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "conv_forward or dgrad or vgg_head" 2>&1 | grep -E "passed|failed|Error|assert|FAILED" | head -20 > gpurun_out/r7a/tests.txt
cat gpurun_out/r7a/tests.txt
L='64,128,64,64;32,128,64,64;32,64,64,64'
for i in 1 2; do
echo "== x32 (default)"; timeout 300 python tools/bench_conv.py --layers "$L" 2>&1 | grep probe
echo "== IMM_CONV_DISABLE=halo2x"; IMM_CONV_DISABLE=halo2x timeout 300 python tools/bench_conv.py --layers "$L" 2>&1 | grep probe
done | tee gpurun_out/r7a/bench_conv_ab.txt
