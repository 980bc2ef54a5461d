#!/bin/bash
# Regenerates profiles/<tag>_{kernel_stats,pmc_hbm_bytes}.csv and <tag>_bench.json on an MI355X box:
#   gpurun -- 'bash tools/make_profiles.sh r02_v1 [stats|full|all]'   (writes under gpurun_out/profiles_<tag>/, copy into profiles/;
#   all = full + SQ ratios, phase timeline, one-step trace, non-headline configurations, LDS-DMA table)
# Counters are collected in their own passes with --kernel-trace only (no sys/hip/hsa tracing together with --pmc).
# Pass 2/3 profile `bench.py --pmc-pass` = 3 eager training steps of the bench workload (one dispatch per launch).
set -e
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_stats /tmp/prof_fetch /tmp/prof_write
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python $R/bench.py --steps 20 --warmup 5 --windows 1 --spin-seconds 0 --no-cpu-baseline --no-pmc > $OUT/stats_run.log 2>&1
python $R/tools/profile_summary.py stats /tmp/prof_stats $OUT/${TAG}_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --windows 1 --spin-seconds 0 --no-cpu-baseline --no-pmc (MI355X; 37 graph-replayed steps + 3 eager timing passes)"
if [ "${2:-full}" != "stats" ]; then   # full / all
  rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d /tmp/prof_fetch -- python $R/bench.py --pmc-pass --steps 3 > $OUT/fetch_run.log 2>&1
  rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d /tmp/prof_write -- python $R/bench.py --pmc-pass --steps 3 > $OUT/write_run.log 2>&1
  python $R/tools/profile_summary.py pmc /tmp/prof_fetch /tmp/prof_write $OUT/${TAG}_pmc_hbm_bytes.csv "rocprofv3 --kernel-trace --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2) -- python bench.py --pmc-pass --steps 3 (MI355X; 3 eager training steps)"
  cd $R && python bench.py > $OUT/${TAG}_bench.json 2> $OUT/bench_run.log
  tail -c 600 $OUT/${TAG}_bench.json
fi
if [ "${2:-full}" = "all" ]; then
  # everything profiles/<tag>_* holds: SQ counter ratios, the device-clock phase timeline of a graph replay, one step's dispatch trace,
  # the non-headline configurations, the VGG launches against the L2 -> LDS DMA rate
  cd $R
  bash tools/pmc_ratios.sh > $OUT/${TAG}_pmc_sq_ratios.txt 2> $OUT/ratios.err
  IMM_DEBUG_STAMPS=marks python tools/graph_timeline.py > $OUT/${TAG}_phase_timeline.txt 2> $OUT/timeline.err
  python tools/bench_configs.py --json $OUT/${TAG}_configs.json > /dev/null 2> $OUT/configs.log
  (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_trace -- python $R/bench.py --steps 10 --warmup 3 --windows 1 --spin-seconds 0 --no-cpu-baseline --no-pmc > /dev/null 2>&1)
  python tools/step_timeline.py /tmp/prof_trace > $OUT/${TAG}_kernel_trace_one_step.txt 2>&1
  IMM_BENCH_DUMP=1 python bench.py --no-cpu-baseline --no-pmc --steps 30 > /dev/null 2> $OUT/dump.err
  python tools/hdeep_dma_roof.py $OUT/dump.err > $OUT/${TAG}_hdeep_lds_dma_roof.txt 2>&1
  rm -rf /tmp/prof_trace
fi
rm -rf /tmp/prof_stats /tmp/prof_fetch /tmp/prof_write
ls -la $OUT
