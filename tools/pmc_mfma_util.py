#!/usr/bin/env python3
"""MFMA utilisation per kernel from one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE; --kernel-trace for the durations):
busy cycles summed over the chip's 1024 SIMDs / (1024 x the launch's shader cycles), the shader cycles taken as GRBM_GUI_ACTIVE / 8 (rocprofv3
sums the 8 XCDs; MI355X_MICROARCH.md, 'DVFS give-back').   python tools/pmc_mfma_util.py <counter_collection.csv> <kernel_trace.csv>"""
import csv, sys, collections
cc, kt = sys.argv[1], sys.argv[2]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(cc)):
    name = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]
    acc[name][r['Counter_Name']].append((r['Dispatch_Id'], float(r['Counter_Value'])))
print('| kernel | launches | avg us | MFMA busy cycles (sum over SIMDs) | shader cycles (GUI_ACTIVE / 8) | MFMA utilisation on those cycles (a LOWER bound on launches under 0.3 ms: the quotient reads high there) | ... at 2.1 GHz over the launch interval |')
print('|---|---|---|---|---|---|---|')
rows = []
for name, c in acc.items():
    if 'SQ_VALU_MFMA_BUSY_CYCLES' not in c or 'GRBM_GUI_ACTIVE' not in c:
        continue
    busy = dict(c['SQ_VALU_MFMA_BUSY_CYCLES']); gui = dict(c['GRBM_GUI_ACTIVE'])
    ids = [i for i in busy if i in gui and i in dur]
    if not ids:
        continue
    n = len(ids)
    b = sum(busy[i] for i in ids) / n; g = sum(gui[i] for i in ids) / n / 8; d = sum(dur[i] for i in ids) / n
    rows.append((d * n, name, n, d, b, g))
for _, name, n, d, b, g in sorted(rows, reverse=True)[:16]:
    if b == 0:
        continue
    print(f'| `{name[:70]}` | {n} | {d:.1f} | {b/1e6:.3f} M | {g/1e3:.1f} k | {100 * b / (1024 * g):.1f} % | {100 * b / (1024 * d * 2100):.1f} % |')
