mkdir -p gpurun_out/r6t
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "conv_forward or dgrad or tap or stats or batch_norm" 2>&1 | grep -E "passed|failed|Error" > gpurun_out/r6t/tests.txt
cat gpurun_out/r6t/tests.txt
L='32,16,256,256;32,32,128,128;32,16,288,256;32,32,256,128;32,64,128,64;32,16,512,512;64,16,512,512'
for i in 1 2; do
echo "== roll (default)"; timeout 300 python tools/bench_conv.py --bn --layers "$L" 2>&1 | grep probe
echo "== IMM_HD_ROLL=0"; IMM_HIP_LIB=$PWD/imm_amd/libimm_roll0.so timeout 300 python tools/bench_conv.py --bn --layers "$L" 2>&1 | grep probe
done | tee gpurun_out/r6t/bench_conv_ab.txt
printf -- "-\nIMM_HIP_LIB=$PWD/imm_amd/libimm_roll0.so\n-\nIMM_HIP_LIB=$PWD/imm_amd/libimm_roll0.so\n-\nIMM_HIP_LIB=$PWD/imm_amd/libimm_roll0.so\n-\nIMM_HIP_LIB=$PWD/imm_amd/libimm_roll0.so\n" > /tmp/ab.txt
bash tools/gpu_ab.sh /tmp/ab.txt gpurun_out/r6t --steps 50 --warmup 10 2>&1 | tee gpurun_out/r6t/ab.txt
