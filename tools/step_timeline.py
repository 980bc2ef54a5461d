"""One steady-state training step of a rocprofv3 --kernel-trace CSV as a timeline: start offset, duration, queue, kernel
name, and per queue the busy / idle totals (a step = the launches between two consecutive clip/adam kernels).
Usage: python tools/step_timeline.py <dir> [marker substring] > timeline.txt"""
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
                rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', '?'), r.get('Stream_Id', '?'), r['Kernel_Name'],
                             r.get('Grid_Size_X', r.get('Grid_Size', '?')), r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?')), r.get('LDS_Block_Size', '?')))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if marker in r[4]]
    ends = [i for k, i in enumerate(marks) if k + 1 == len(marks) or marks[k + 1] != i + 1]
    if len(ends) < 3:
        print('not enough steps', len(ends)); return
    # the shortest complete step of the run = a steady-state graph replay (eager passes are longer)
    cands = []
    for a, b in zip(ends[:-1], ends[1:]):
        st = rows[a + 1:b + 1]
        cands.append((max(r[1] for r in st) - st[0][0], a + 1, b + 1))
    cands.sort()
    _w, lo, hi = cands[len(cands) // 4] if len(sys.argv) <= 3 else cands[int(sys.argv[3])]
    step = rows[lo:hi]
    t0 = step[0][0]
    busy = defaultdict(int)
    last_end = {}
    print('# %d kernels, wall %.1f us' % (len(step), (max(r[1] for r in step) - t0) / 1e3))
    print('# start_us  dur_us  gap_on_queue_us  queue/stream  grid  wg  lds  name')
    for s, e, q, st, n, g, wg, lds in step:
        key = (q, st)
        gap = (s - last_end[key]) / 1e3 if key in last_end else 0.0
        last_end[key] = e
        busy[key] += e - s
        short = n.replace('void ', '').replace('(anonymous namespace)::', '')
        print('%9.1f %7.1f %7.1f  q%s/s%s  %8s %5s %6s  %s' % ((s - t0) / 1e3, (e - s) / 1e3, gap, q, st, g, wg, lds, short[:110]))
    for k, v in sorted(busy.items()):
        print('# queue %s stream %s: busy %.1f us' % (k[0], k[1], v / 1e3))


if __name__ == '__main__':
    main()
