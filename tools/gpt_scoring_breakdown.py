"""Per-kernel breakdown of ONE GPT scoring pass (forward_all) from a rocprofv3 kernel trace of tools/bench_gpt.py: the span between two
embed_kernel launches of the scoring loop.   python tools/gpt_scoring_breakdown.py <kernel_trace.csv> [rows]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rows.sort(key=lambda r: int(r['Start_Timestamp']))
emb = [i for i, r in enumerate(rows) if 'embed_kernel' in r['Kernel_Name']]
step = rows[emb[3]:emb[4]]
t0 = int(step[0]['Start_Timestamp'])
wall = (int(step[-1]['End_Timestamp']) - t0) / 1e6
agg = collections.OrderedDict()
for r in step:
    n = r['Kernel_Name'].replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '').split('(')[0]
    key = (n, int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']))
    e = agg.setdefault(key, [0, 0.0])
    e[0] += 1
    e[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
print(f'pass wall {wall:.2f} ms, {len(step)} launches')
print('| kernel | workgroups | launches | total ms | avg us |')
print('|---|---|---|---|---|')
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f'| `{k[0][:60]}` | {k[1]} | {v[0]} | {v[1] / 1e3:.2f} | {v[1] / v[0]:.1f} |')
