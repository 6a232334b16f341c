#!/usr/bin/env python3
"""Per-step kernel table from a rocprofv3 --kernel-trace CSV of bench.py: takes the training step in the middle of the
trace (delimited by the step's last kernel, the final adamw launch group) and prints a markdown table.

    python tools/profile_summary.py gpurun_out/prof10/runc/246_kernel_trace.csv [marker_kernel_substring]
"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r'^void ', '', name)
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    return re.sub(r'\(.*$', '', name)


def main():
    path = sys.argv[1]
    marker = sys.argv[2] if len(sys.argv) > 2 else 'clip_coef_kernel'
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if marker in r[2]]
    if len(marks) < 4:
        sys.exit(f'marker {marker!r} found {len(marks)} times')
    mid = len(marks) // 2
    step = rows[marks[mid] + 1: marks[mid + 1] + 1]        # kernels after one marker up to and including the next
    # rotate so that the window is one whole step regardless of where the marker sits inside it
    agg = defaultdict(lambda: [0, 0])
    for s, e, n in step:
        agg[n][0] += 1
        agg[n][1] += e - s
    total = sum(v[1] for v in agg.values())
    span = step[-1][1] - step[0][0]
    print(f'One training step from the middle of the trace: {len(step)} kernels, sum of durations {total/1e3:.0f} us, '
          f'first-start to last-end {span/1e3:.0f} us under the profiler.\n')
    print('| kernel | launches / step | total us | avg us | share |\n|---|---|---|---|---|')
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f'| `{n}` | {c} | {t/1e3:.1f} | {t/1e3/c:.1f} | {100*t/total:.1f}% |')


if __name__ == '__main__':
    main()
