"""Ordered kernel list of ONE GPT training step from a rocprofv3 kernel trace of tools/bench_gpt.py (the span between the last two
adamw_multi_kernel launches): start, duration, gap, grid.   python tools/gpt_step_timeline.py <kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adamw_multi' in r['Kernel_Name']]
step = rows[idx[-2] + 1:idx[-1] + 1]
t0 = int(step[0]['Start_Timestamp'])
prev_end = t0
print('| # | kernel | wgs | start us | dur us | gap us |')
print('|---|---|---|---|---|---|')
for i, r in enumerate(step):
    n = r['Kernel_Name'].replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '').split('(')[0]
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    wgs = (int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])) * (int(r['Grid_Size_Y']) // max(1, int(r['Workgroup_Size_Y']))) * (int(r['Grid_Size_Z']) // max(1, int(r['Workgroup_Size_Z'])))
    print(f'| {i} | `{n[:70]}` | {wgs} | {(s - t0) / 1e3:.1f} | {(e - s) / 1e3:.1f} | {(s - prev_end) / 1e3:.1f} |')
    prev_end = e
