set -u
mkdir -p gpurun_out/r5a
L='64,64,64,128;64,64,128,128;64,32,128,256;64,32,256,256;64,16,256,512;64,16,512,512;32,16,512,512;32,64,128,128;32,32,256,256'
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv_forward or relu_stats_mask or test_conv_dgrad or tap_epilogue" > gpurun_out/r5a/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r5a/pytest.log
echo "== new"; timeout 300 python tools/bench_conv.py --layers "$L" 2>&1 | tee gpurun_out/r5a/conv_new.txt
echo "== old"; IMM_CONV_DISABLE=hdeep6 timeout 300 python tools/bench_conv.py --layers "$L" 2>&1 | tee gpurun_out/r5a/conv_old.txt
printf -- "-\nIMM_CONV_DISABLE=hdeep6\n-\nIMM_CONV_DISABLE=hdeep6\n" > /tmp/envs.txt
bash tools/gpu_ab.sh /tmp/envs.txt gpurun_out/r5a/ab --steps 100 --warmup 10
