#!/usr/bin/env python3
"""One training step's kernel timeline from a rocprofv3 --kernel-trace CSV: every launch of the LAST complete step in order, with its
duration and the idle gap in front of it, and the totals (busy / idle).  A step starts at the subsample conv's im2col kernel.

    python tools/step_timeline.py <kernel_trace.csv> [first-kernel-substring]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
first = sys.argv[2] if len(sys.argv) > 2 else 'im2col_kernel'
rows.sort(key=lambda r: int(r['Start_Timestamp']))
starts = [i for i, r in enumerate(rows) if first in r['Kernel_Name']]
if len(starts) < 3:
    sys.exit('not enough steps in the trace')
a, b = starts[-2], starts[-1]
step = rows[a:b]
t0 = int(step[0]['Start_Timestamp'])
prev_end = t0
busy = 0
print(f'{len(step)} launches; step wall {(int(rows[b]["Start_Timestamp"]) - t0) / 1e3:.1f} us')
print('| # | kernel | start us | dur us | gap before us |\n|---|---|---|---|---|')
for i, r in enumerate(step):
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
    n = re.sub(r'^void ', '', n)
    n = re.sub(r'\(.*$', '', n)[:60]
    print(f'| {i} | `{n}` | {(s - t0) / 1e3:.1f} | {(e - s) / 1e3:.1f} | {(s - prev_end) / 1e3:.1f} |')
    busy += e - s
    prev_end = max(prev_end, e)
wall = int(rows[b]['Start_Timestamp']) - t0
print(f'\nbusy {busy / 1e3:.1f} us, idle {(wall - busy) / 1e3:.1f} us of {wall / 1e3:.1f} us')
