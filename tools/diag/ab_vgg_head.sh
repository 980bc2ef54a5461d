mkdir -p gpurun_out/r6j
timeout 900 python -m pytest tests/test_switches_gpu.py tests/test_step_gpu.py tests/test_golden.py tests/test_vgg_file_gpu.py -x -q 2>&1 | tail -8 > gpurun_out/r6j/tests.txt
cat gpurun_out/r6j/tests.txt
for i in 1 2; do
IMM_VGG_HEAD=0 timeout 300 python bench.py --steps 50 --warmup 10 --no-pmc 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('HEAD=0', d['ms_per_step'], d['step']['windows_ms'])"
IMM_VGG_HEAD=1 timeout 300 python bench.py --steps 50 --warmup 10 --no-pmc 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('HEAD=1', d['ms_per_step'], d['step']['windows_ms'])"
done 2>&1 | tee gpurun_out/r6j/ab_head.txt
