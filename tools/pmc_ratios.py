"""Summarise the counter_collection CSVs of tools/pmc_ratios.sh: one line per kernel (see profiles/r03_v3_pmc_sq_ratios.txt)."""
import csv, glob, sys, collections
root = sys.argv[1]
ctr = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for p in glob.glob(root + '/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(p)):
        n = row['Kernel_Name'].replace('void ', '')[:64]
        ctr[n][row['Counter_Name']] += float(row['Counter_Value'])
for n, c in sorted(ctr.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0)):
    wc = c.get('SQ_WAVE_CYCLES', 0)
    if wc < 1e6: continue
    mf = c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 16.0
    print('%-64s wave_cyc %7.1fM  mfma_duty %4.1f%%  wait_any %4.1f%%  lds_conf/active %4.2f  VALU/MFMA %5.2f  LDS/MFMA %4.2f  SALU/MFMA %4.2f' % (
        n, wc * 4 / 1e6, 100 * c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (wc * 4 + 1), 100 * c.get('SQ_WAIT_ANY', 0) / (wc + 1),
        c.get('SQ_LDS_BANK_CONFLICT', 0) / (c.get('SQ_LDS_IDX_ACTIVE', 0) + 1),
        (c.get('SQ_INSTS_VALU', 0) - mf) / (mf + 1), c.get('SQ_INSTS_LDS', 0) / (mf + 1), c.get('SQ_INSTS_SALU', 0) / (mf + 1)))
