mkdir -p gpurun_out/r6q
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "sse or perceptual" 2>&1 | grep -E "passed|failed|Error" > gpurun_out/r6q/tests.txt
timeout 1500 python -m pytest tests/test_switches_gpu.py tests/test_golden.py tests/test_step_gpu.py tests/test_witness_gpu.py -q -x 2>&1 | grep -E "passed|failed|Error|assert" >> gpurun_out/r6q/tests.txt
cat gpurun_out/r6q/tests.txt
printf -- "-\nIMM_SSE_ALL=0\n-\nIMM_SSE_ALL=0\n-\nIMM_SSE_ALL=0\n" > /tmp/ab.txt
bash tools/gpu_ab.sh /tmp/ab.txt gpurun_out/r6q --steps 50 --warmup 10 2>&1 | tee gpurun_out/r6q/ab.txt
