set -u
mkdir -p gpurun_out/r5d
for i in 1 2 3 4 5 6; do
  timeout 300 python -m pytest tests/test_dp_gpu.py -x -q -k "two_ranks_on_one_gpu" 2>&1 | tail -1 | sed "s/^/NEW $i: /"
done
cd build_ab/old
for i in 1 2 3 4 5 6; do
  timeout 300 python -m pytest tests/test_dp_gpu.py -x -q -k "two_ranks_on_one_gpu" -p no:cacheprovider 2>&1 | tail -1 | sed "s/^/OLD $i: /"
done
