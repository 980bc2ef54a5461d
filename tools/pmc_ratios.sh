#!/bin/bash
# Per-kernel SQ counter ratios of one training step (matrix-pipe duty, s_waitcnt share, LDS bank conflicts, non-MFMA instructions per
# MFMA): gpurun -- "bash tools/pmc_ratios.sh > gpurun_out/pmc_ratios.txt".  Counters in their own passes, --kernel-trace only besides --pmc.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_a
for SET in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM"; do
  (cd $R && rocprofv3 --kernel-trace --output-format csv --pmc $SET -d /tmp/pmc_a/$RANDOM -- python bench.py --pmc-pass --steps 1) > /dev/null 2>&1
done
python $R/tools/pmc_ratios.py /tmp/pmc_a
