#!/bin/bash
# Regenerates profiles/<tag>_{kernel_stats,pmc_hbm_bytes}.csv and <tag>_bench.json on an MI355X box:
#   gpurun -- 'bash tools/make_profiles.sh r02_v1'      (writes under gpurun_out/profiles_<tag>/, copy into profiles/)
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
if [ "${2:-full}" != "stats" ]; then
  rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d /tmp/prof_fetch -- python $R/bench.py --pmc-pass --steps 3 > $OUT/fetch_run.log 2>&1
  rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d /tmp/prof_write -- python $R/bench.py --pmc-pass --steps 3 > $OUT/write_run.log 2>&1
  python $R/tools/profile_summary.py pmc /tmp/prof_fetch /tmp/prof_write $OUT/${TAG}_pmc_hbm_bytes.csv "rocprofv3 --kernel-trace --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2) -- python bench.py --pmc-pass --steps 3 (MI355X; 3 eager training steps)"
  cd $R && python bench.py > $OUT/${TAG}_bench.json 2> $OUT/bench_run.log
  tail -c 600 $OUT/${TAG}_bench.json
fi
rm -rf /tmp/prof_stats /tmp/prof_fetch /tmp/prof_write
ls -la $OUT
