#!/usr/bin/env python3
"""Markdown table of a rocprofv3 --stats kernel_stats.csv (top kernels by total time).

    python tools/stats_summary.py gpurun_out/prof_x/runc/245_kernel_stats.csv "command that was profiled" [top_n]
"""
import csv
import re
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    cmd = sys.argv[2] if len(sys.argv) > 2 else ''
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    print(f'Command (MI355X box): `{cmd}`\n')
    print(f'All kernels of the run: {sum(int(r["Calls"]) for r in rows)} launches, {tot/1e6:.1f} ms of kernel time.\n')
    print('| kernel | calls | total ms | avg us | share |\n|---|---|---|---|---|')
    for r in rows[:top]:
        n = re.sub(r'\(anonymous namespace\)::', '', r['Name'])
        n = re.sub(r'^void ', '', n)
        n = re.sub(r'\(.*$', '', n)
        print(f"| `{n[:70]}` | {int(r['Calls'])} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['AverageNs'])/1e3:.1f} | "
              f"{100*float(r['TotalDurationNs'])/tot:.1f}% |")


if __name__ == '__main__':
    main()
