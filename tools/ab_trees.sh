#!/bin/bash
# Same-box A/B of two SOURCE TREES (for changes that touch the Python side or the ABI, where IMM_HIP_LIB of tools/gpu_ab.sh is not
# enough): the working tree against an exported older commit with its own built library.
#   prepare (here):  mkdir -p build_ab/old && git archive <commit> | tar -x -C build_ab/old && cp <its libimm_hip.so> build_ab/old/imm_amd/
#   run (GPU box):   gpurun -- 'bash tools/ab_trees.sh 3 --steps 300 --warmup 30'      (build_ab/ is git-ignored but travels with gpurun)
# Prints ms/step (median window), the windows, and the eager-timed filter-gradient / reduction / optimizer launches, alternating new / old.
n=$1; shift
for i in $(seq $n); do
  for t in new old; do
    if [ $t = new ]; then d=.; else d=build_ab/old; fi
    (cd $d && timeout 240 python bench.py --no-cpu-baseline --no-pmc "$@" 2>/dev/null | tail -1) | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('AB $t %.4f ms windows %s wgrad %.3f reduce %.3f adam %.3f' % (d['ms_per_step'], d['step']['windows_ms'], k.get('conv_wgrad',{}).get('ms',0), k.get('wgrad_reduce',{}).get('ms',0), k.get('clip_adam',{}).get('ms',0)))"
  done
done
