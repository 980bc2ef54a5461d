"""Timeline analysis of one training step from a rocprofv3 --kernel-trace CSV: union busy time, idle gaps between
kernels, overlap, and per-kernel-name totals inside the LAST `n` complete steps (a step = the launches between two
consecutive clip/adam kernels).  Usage: python tools/trace_gaps.py <dir> [marker substring]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    marker = sys.argv[2] if len(sys.argv) > 2 else 'clip_adam'
    rows = []
    for p in glob.glob(os.path.join(root, '**', '*kernel_trace.csv'), recursive=True):
        with open(p, newline='') as f:
            for r in csv.DictReader(f):
                rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if marker in r[2]]
    # a step has several marker kernels in a row (norm / apply); take the last of each run
    ends = [i for k, i in enumerate(marks) if k + 1 == len(marks) or marks[k + 1] != i + 1]
    if len(ends) < 3:
        print('not enough steps', len(ends)); return
    lo, hi = ends[-3] + 1, ends[-2] + 1          # one complete steady-state step
    step = rows[lo:hi]
    t0, t1 = step[0][0], max(r[1] for r in step)
    busy, cur_s, cur_e = 0, step[0][0], step[0][1]
    gaps = []
    for s, e, n in step[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, n))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    total = sum(e - s for s, e, _ in step)
    print('kernels %d  wall %.1f us  union-busy %.1f us  idle %.1f us  sum-of-kernels %.1f us (overlap %.1f)' % (
        len(step), (t1 - t0) / 1e3, busy / 1e3, (t1 - t0 - busy) / 1e3, total / 1e3, (total - busy) / 1e3))
    gaps.sort(reverse=True)
    print('gaps: n=%d mean %.2f us; >3us: %d; top:' % (len(gaps), sum(g for g, _ in gaps) / max(1, len(gaps)) / 1e3,
                                                       sum(1 for g, _ in gaps if g > 3000)))
    for g, n in gaps[:12]:
        print('   %7.2f us before %s' % (g / 1e3, n[:90]))
    by = defaultdict(lambda: [0, 0])
    for s, e, n in step:
        k = n.split('(')[0][:60]
        by[k][0] += e - s; by[k][1] += 1
    for k, (t, c) in sorted(by.items(), key=lambda kv: -kv[1][0])[:32]:
        print('%9.1f us  n=%3d  %s' % (t / 1e3, c, k))


if __name__ == '__main__':
    main()
