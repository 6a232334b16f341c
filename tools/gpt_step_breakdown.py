"""Per-kernel breakdown of ONE GPT training step from a rocprofv3 kernel trace of tools/bench_gpt.py (the span between the last
two adamw_multi_kernel launches).   python tools/gpt_step_breakdown.py <kernel_trace.csv> [rows]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adamw_multi' in r['Kernel_Name']]
step = rows[idx[-2] + 1:idx[-1] + 1]
t0 = int(step[0]['Start_Timestamp'])
wall = (int(step[-1]['End_Timestamp']) - t0) / 1e6
agg = collections.OrderedDict()
for r in step:
    n = r['Kernel_Name'].replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '').split('(')[0]
    key = (n, int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']))
    e = agg.setdefault(key, [0, 0.0])
    e[0] += 1
    e[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
busy = sum(v[1] for v in agg.values()) / 1e3
print(f'step wall {wall:.2f} ms, busy {busy:.2f} ms, {len(step)} launches')
print('| kernel | workgroups | launches | total ms | avg us |')
print('|---|---|---|---|---|')
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f'| `{k[0][:60]}` | {k[1]} | {v[0]} | {v[1] / 1e3:.2f} | {v[1] / v[0]:.1f} |')
