#!/bin/bash
# A/B of engine switches on one box: each line of the here-doc is an environment (possibly empty) for one bench run.
#   gpurun -- 'bash tools/sweep_env.sh <tag> < tools/sweeps/<file>'   or edit the default list below.  Output: gpurun_out/<tag>/sweep.txt
TAG=${1:-sweep}
OUT=gpurun_out/$TAG
mkdir -p $OUT
: > $OUT/sweep.txt
while IFS= read -r ENVLINE; do
  [ -z "${ENVLINE// }" ] && ENVLINE="BASELINE=1"
  RES=$(env $ENVLINE timeout 300 python bench.py --steps 40 --warmup 5 --windows 3 --no-cpu-baseline --no-pmc 2>$OUT/last.err | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f ms  windows %s  eager_sum %.3f' % (d['ms_per_step'], d['step']['windows_ms'], d['step']['sum_kernel_ms_eager']))
except Exception as e: print('FAILED', e)")
  echo "$ENVLINE => $RES" | tee -a $OUT/sweep.txt
done
