# same-box A/B of the current tree against the round-5 tree (git archive c7d2495 under build_ab/r05, built there)
mkdir -p gpurun_out/r6y
for i in 1 2 3 4; do
(cd build_ab/r05 && timeout 300 python bench.py --steps 50 --warmup 10 --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('round-5 tree  %.4f ms  windows %s' % (d['ms_per_step'], d['step']['windows_ms']))")
timeout 300 python bench.py --steps 50 --warmup 10 --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('round-6 tree  %.4f ms  windows %s' % (d['ms_per_step'], d['step']['windows_ms']))"
done | tee gpurun_out/r6y/ab_vs_r05.txt
