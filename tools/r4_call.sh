#!/bin/bash
# One GPU-box call of round 4: every stage under its own timeout, logs under gpurun_out/<tag>/ (merged back by gpurun).
# usage: bash tools/r4_call.sh <tag> <stage> [<stage> ...]      stages: s2d step dp ab kern full bench prof
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
for st in "$@"; do
  t0=$(date +%s)
  case $st in
    s2d)  timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "stride2" > $out/s2d.log 2>&1 ;;
    newtests) timeout 600 python -m pytest tests/test_tf_checkpoint_gpu.py tests/test_step_gpu.py -q -k "loss_scale_state or follows_the_device" > $out/newtests.log 2>&1 ;;
    first) timeout 300 python -m pytest tests/test_kernels_gpu.py -q -k "first_conv" > $out/first.log 2>&1 ;;
    nol)  timeout 400 python -m pytest tests/test_kernels_gpu.py -q -k "nol or norm_on_load" > $out/nol.log 2>&1 ;;
    kern) timeout 600 python -m pytest tests/test_kernels_gpu.py -q > $out/kern.log 2>&1 ;;
    step) timeout 600 python -m pytest tests/test_step_gpu.py -q -x > $out/step.log 2>&1 ;;
    dp)   timeout 900 python -m pytest tests/test_dp_gpu.py -q > $out/dp.log 2>&1 ;;
    full) timeout 1200 python -m pytest tests -m gpu -q > $out/full.log 2>&1 ;;
    ab)   bash tools/ab_trees.sh 2 --steps 200 --warmup 20 > $out/ab.log 2>&1 ;;
    bench) timeout 400 python bench.py > $out/bench.json 2> $out/bench.err ;;
    dumpold) (cd build_ab/old && IMM_BENCH_DUMP=1 timeout 300 python bench.py --no-cpu-baseline --no-pmc --steps 50 > ../../$out/dumpold.json 2> ../../$out/dumpold.err) ;;
    dumpnonol) IMM_CONV_DISABLE=nol IMM_BENCH_DUMP=1 timeout 300 python bench.py --no-cpu-baseline --no-pmc --steps 50 > $out/dumpnonol.json 2> $out/dumpnonol.err ;;
    dump) IMM_BENCH_DUMP=1 timeout 300 python bench.py --no-cpu-baseline --no-pmc --steps 50 > $out/dump.json 2> $out/dump.err ;;
    prof) bash tools/make_profiles.sh ${tag} > $out/prof.log 2>&1 ;;
    ratios) bash tools/pmc_ratios.sh > $out/pmc_sq_ratios.txt 2> $out/ratios.err ;;
    configs) timeout 500 python tools/bench_configs.py --json $out/configs.json > /dev/null 2> $out/configs.log ;;
    timeline) IMM_DEBUG_STAMPS=marks timeout 300 python tools/graph_timeline.py > $out/phase_timeline.txt 2> $out/timeline.err ;;
    trace1) (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/trace1 -- python $PWD/bench.py --steps 10 --warmup 3 --windows 1 --spin-seconds 0 --no-cpu-baseline --no-pmc > /dev/null 2>&1); python tools/rocprof_csv.py /tmp/trace1 > $out/kernel_trace_one_step.txt 2>&1 ;;
    *) echo "unknown stage $st" ;;
  esac
  echo "STAGE $st rc=$? $(( $(date +%s) - t0 )) s" | tee -a $out/stages.log
done
tail -n 5 $out/*.log 2>/dev/null | tail -n 60
