#!/bin/bash
# A/B bench runs on ONE box: each line of $1 is an environment ("VAR=val VAR2=val"; "-" = defaults); prints ms/step per line.
# usage: tools/gpu_ab.sh <file with env lines> <out dir> [bench args...]
set -u
envs="$1"; out="$2"; shift 2
mkdir -p "$out"
i=0
while IFS= read -r line; do
  i=$((i+1))
  [ "$line" = "-" ] && line=""
  env $line timeout 240 python bench.py --no-cpu-baseline --no-pmc "$@" > "$out/ab_$i.json" 2> "$out/ab_$i.err"
  python - "$out/ab_$i.json" "$line" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k = d['kernels']
    print('AB %-60s %.4f ms  windows %s  wgrad %.3f ms  reduce %.3f ms' % (sys.argv[2] or '(defaults)', d['ms_per_step'], d['step']['windows_ms'],
          k.get('conv_wgrad', {}).get('ms', 0), k.get('wgrad_reduce', {}).get('ms', 0)))
except Exception as e:
    print('AB %-60s FAILED %r' % (sys.argv[2], e))
PY
done < "$envs"
