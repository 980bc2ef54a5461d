mkdir -p gpurun_out/r6m
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "pose_head or softargmax or gaussian" 2>&1 | grep -E "passed|failed" > gpurun_out/r6m/tests.txt
timeout 900 python -m pytest tests/test_golden.py tests/test_step_gpu.py -q -x 2>&1 | grep -E "passed|failed" >> gpurun_out/r6m/tests.txt
cat gpurun_out/r6m/tests.txt
for i in 1 2; do timeout 300 python bench.py --steps 50 --warmup 10 --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new', d['ms_per_step'], d['step']['windows_ms'])"; done | tee gpurun_out/r6m/bench.txt
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pt && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --windows 1 --spin-seconds 0 --no-cpu-baseline --no-pmc > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python tools/profile_summary.py stats /tmp/pt gpurun_out/r6m/kernel_stats.csv "r6m" && grep -i "finalize\|softargmax\|pose_head" gpurun_out/r6m/kernel_stats.csv | sed 's/(.*)"//' | cut -c1-100
