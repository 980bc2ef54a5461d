"""Build the committed profile summaries from rocprofv3 CSV output.
  python tools/profile_summary.py stats <trace_dir> <out.csv> "<header comment>"
  python tools/profile_summary.py pmc <fetch_dir> <write_dir> <out.csv> "<header comment>"
trace_dir: rocprofv3 --kernel-trace --output-format csv ; fetch_dir / write_dir: separate --pmc FETCH_SIZE / --pmc WRITE_SIZE
passes (MI355X_MICROARCH.md: the two counters do not fit one pass; FETCH_SIZE x2 on gfx950)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def rows_of(root, suffix):
    for p in glob.glob(os.path.join(root, '**', '*' + suffix), recursive=True):
        with open(p, newline='') as f:
            for r in csv.DictReader(f):
                yield r


def stats(trace_dir, out, header):
    d = defaultdict(list)
    for r in rows_of(trace_dir, 'kernel_trace.csv'):
        d[r['Kernel_Name']].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    total = sum(sum(v) for v in d.values())
    with open(out, 'w', newline='') as f:
        f.write('"# %s"\n' % header.replace('"', "'"))
        w = csv.writer(f)
        w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'MinNs', 'MaxNs', 'Percentage'])
        for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([k, len(v), sum(v), round(sum(v) / len(v), 1), min(v), max(v), round(100.0 * sum(v) / total, 3)])


def pmc(fetch_dir, write_dir, out, header):
    fetch, write, dur = defaultdict(float), defaultdict(float), defaultdict(list)
    n = defaultdict(int)
    for r in rows_of(fetch_dir, 'counter_collection.csv'):
        if r['Counter_Name'] == 'FETCH_SIZE':
            fetch[r['Kernel_Name']] += float(r['Counter_Value']); n[r['Kernel_Name']] += 1
    for r in rows_of(write_dir, 'counter_collection.csv'):
        if r['Counter_Name'] == 'WRITE_SIZE':
            write[r['Kernel_Name']] += float(r['Counter_Value'])
    for r in rows_of(fetch_dir, 'kernel_trace.csv'):
        dur[r['Kernel_Name']].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    with open(out, 'w', newline='') as f:
        f.write('# %s\n' % header)
        f.write('# FETCH_SIZE on gfx950 counts 128-B requests as 64 B for wide coalesced reads (MI355X_MICROARCH.md): '
                'fetch_MB_corrected = 2 x raw\n')
        w = csv.writer(f)
        w.writerow(['kernel', 'dispatches', 'fetch_KB_raw_sum', 'write_KB_sum', 'fetch_MB_corrected_per_dispatch',
                    'write_MB_per_dispatch', 'profiled_ns_per_dispatch'])
        for k in sorted(fetch, key=lambda k: -(2 * fetch[k] + write[k])):
            c = n[k]
            ns = int(sum(dur[k]) / len(dur[k])) if dur[k] else 0
            w.writerow([k, c, round(fetch[k], 1), round(write[k], 1), round(2 * fetch[k] / c / 1024, 3),
                        round(write[k] / c / 1024, 3), ns])


if __name__ == '__main__':
    if sys.argv[1] == 'stats':
        stats(sys.argv[2], sys.argv[3], sys.argv[4])
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5])
