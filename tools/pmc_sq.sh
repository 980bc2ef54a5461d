#!/bin/bash
# SQ / TCC counters of ONE convolution layer of tools/bench_conv.py's table (separate rocprofv3 passes, --kernel-trace only
# besides --pmc): bash tools/pmc_sq.sh <layer tag substring> [ENV=VALUE ...]  -> stdout
TAGKEY=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
echo "== layer $TAGKEY  env: $* (tools/pmc_layer.py, 5 launches) =="
rm -rf /tmp/pmc_k; env "$@" rocprofv3 --kernel-trace --output-format csv -d /tmp/pmc_k -- python $R/tools/pmc_layer.py "$TAGKEY" 5 > /dev/null 2>&1
python $R/tools/rocprof_csv.py /tmp/pmc_k conv_
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  rm -rf /tmp/pmc_c; env "$@" rocprofv3 --kernel-trace --output-format csv --pmc $SET -d /tmp/pmc_c -- python $R/tools/pmc_layer.py "$TAGKEY" 5 > /dev/null 2>&1
  python $R/tools/rocprof_csv.py /tmp/pmc_c conv_ | grep -v " mean=.* us$"
done
rm -rf /tmp/pmc_k /tmp/pmc_c
