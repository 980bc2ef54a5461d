set -u
mkdir -p gpurun_out/r5d
python tools/_det_s2f.py 2>&1 | grep DET
IMM_CONV_DISABLE=s2f timeout 900 python -m pytest tests/test_dp_gpu.py -x -q -k "two_ranks_on_one_gpu" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_dp_gpu.py -x -q -k "two_ranks_on_one_gpu" 2>&1 | tail -3
