mkdir -p gpurun_out/r7d
timeout 600 python -m pytest tests/test_switches_gpu.py -q -k "halo2_on_32x32 or vgg_head" 2>&1 | grep -E "passed|failed|Error" 
IMM_HALO2X=1 timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "conv_forward or dgrad" 2>&1 | grep -E "passed|failed|Error"
timeout 200 tools/probes/bin/mrp > gpurun_out/r7d/mfma_rate.txt; timeout 100 tools/probes/bin/lpp > gpurun_out/r7d/lds_pattern.txt; cat gpurun_out/r7d/mfma_rate.txt | tail -3
