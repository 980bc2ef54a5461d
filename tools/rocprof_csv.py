"""Summarise rocprofv3 CSV output (kernel_trace / counter_collection) under a directory: per kernel name, launches,
mean duration and mean of every counter.  Usage: python tools/rocprof_csv.py <dir> [name substring]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    key = sys.argv[2] if len(sys.argv) > 2 else ''
    dur = defaultdict(list)
    ctr = defaultdict(lambda: defaultdict(list))
    for p in glob.glob(os.path.join(root, '**', '*.csv'), recursive=True):
        with open(p, newline='') as f:
            rd = csv.DictReader(f)
            cols = rd.fieldnames or []
            for row in rd:
                name = row.get('Kernel_Name') or row.get('kernel_name') or ''
                if key not in name:
                    continue
                name = name[:70]
                if 'Start_Timestamp' in cols and 'End_Timestamp' in cols and 'Counter_Name' not in cols:
                    dur[name].append((int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3)
                if 'Counter_Name' in cols:
                    ctr[name][row['Counter_Name']].append(float(row['Counter_Value']))
    for name, d in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        print('%-70s n=%4d mean=%9.2f us' % (name, len(d), sum(d) / len(d)))
    for name, cs in ctr.items():
        print(name)
        for c, v in sorted(cs.items()):
            print('    %-34s n=%4d mean=%16.1f' % (c, len(v), sum(v) / len(v)))


if __name__ == '__main__':
    main()
