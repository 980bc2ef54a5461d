mkdir -p gpurun_out/r6s
timeout 1500 python -m pytest tests/test_golden.py tests/test_step_gpu.py tests/test_switches_gpu.py -q -x 2>&1 | grep -E "passed|failed|Error|assert" > gpurun_out/r6s/tests.txt
cat gpurun_out/r6s/tests.txt
printf -- "-\nIMM_HIP_LIB=$PWD/imm_amd/libimm_roll0.so\n-\nIMM_HIP_LIB=$PWD/imm_amd/libimm_roll0.so\n-\nIMM_HIP_LIB=$PWD/imm_amd/libimm_roll0.so\n" > /tmp/ab.txt
bash tools/gpu_ab.sh /tmp/ab.txt gpurun_out/r6s --steps 50 --warmup 10 2>&1 | tee gpurun_out/r6s/ab.txt
