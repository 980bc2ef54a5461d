mkdir -p gpurun_out/r6r
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "conv_forward or dgrad or tap" 2>&1 | grep -E "passed|failed|Error" > gpurun_out/r6r/tests.txt
cat gpurun_out/r6r/tests.txt
L='64,64,128,128;64,32,256,256;64,16,512,512;64,64,64,128;64,32,128,256;64,16,256,512;32,64,128,128;32,32,256,256'
for i in 1 2; do
echo "== roll (default)"; timeout 300 python tools/bench_conv.py --layers "$L" 2>&1 | grep probe
echo "== IMM_H6_ROLL=0"; IMM_HIP_LIB=$PWD/imm_amd/libimm_roll0.so timeout 300 python tools/bench_conv.py --layers "$L" 2>&1 | grep probe
done | tee gpurun_out/r6r/bench_conv_ab.txt
