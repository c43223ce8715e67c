#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace CSV of a multi-stream bench run and prints, per kernel, the average duration under
concurrency, plus the busy fraction of the timeline (union of kernel intervals / wall) and the average number of
kernels in flight.   python tools/trace_overlap.py <dir with *_kernel_trace.csv>"""
import csv
import glob
import os
import sys
import collections

f = max(glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True), key=os.path.getmtime)
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))]
rows.sort()
# keep the steady-state second half
t0 = rows[len(rows) // 2][0]
rows = [r for r in rows if r[0] >= t0]
wall = rows[-1][1] - rows[0][0]
busy, cur_s, cur_e = 0, None, None
for s, e, _ in rows:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e, _ in rows)
print(f'wall {wall / 1e3:.1f} us, timeline busy {busy / wall:.3f}, kernel-time / wall (avg kernels in flight) {tot / wall:.2f}')
agg = collections.defaultdict(list)
for s, e, n in rows:
    agg[n[:60]].append(e - s)
for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f'{n:60s} n={len(v):5d} avg {sum(v) / len(v) / 1e3:8.1f} us  share of kernel-time {sum(v) / tot:.3f}')
