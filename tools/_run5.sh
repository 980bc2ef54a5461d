set -u
mkdir -p gpurun_out/r5c
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "stride2_forward or conv_forward or stride2_halo" > gpurun_out/r5c/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/r5c/pytest.log
L='32,64,64,128,3,2;32,32,128,256,3,2'
echo "== s2f"; timeout 300 python tools/bench_conv.py --bn --layers "$L" 2>&1 | grep probe
echo "== im2col"; IMM_CONV_DISABLE=s2f timeout 300 python tools/bench_conv.py --bn --layers "$L" 2>&1 | grep probe
printf -- "-\nIMM_CONV_DISABLE=s2f\n-\nIMM_CONV_DISABLE=s2f\n" > /tmp/envs.txt
bash tools/gpu_ab.sh /tmp/envs.txt gpurun_out/r5c/ab --steps 100 --warmup 10
