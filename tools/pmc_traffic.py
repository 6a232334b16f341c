#!/usr/bin/env python3
"""HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; one counter per pass, as MI355X_MICROARCH.md
prescribes), with that guide's gfx950 correction: FETCH_SIZE (KiB) counts exactly half the bytes of wide (16 B per lane) coalesced
reads, so fetched bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE (KiB) is exact for 16-byte streaming stores.

    python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> [<out.md>] [kernel substring ...]

Writes, per kernel name (template arguments kept), the median counter values and bytes per launch.  bench.py reads the JSON for
roofline.traffic (only an entry whose kernel name matches what it ran)."""
import csv
import json
import re
import statistics
import sys


def medians(path, counter):
    per = {}
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        name = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
        name = re.sub(r'^void ', '', name)
        name = re.sub(r'\(.*$', '', name)
        per.setdefault(name, []).append(float(r['Counter_Value']))
    return {k: (statistics.median(v), len(v)) for k, v in per.items()}


def main():
    fetch, write, out_json = sys.argv[1:4]
    out_md = sys.argv[4] if len(sys.argv) > 4 and sys.argv[4].endswith('.md') else None
    want = [a for a in sys.argv[4:] if not a.endswith('.md')]
    f, w = medians(fetch, 'FETCH_SIZE'), medians(write, 'WRITE_SIZE')
    rows = []
    for k in sorted(set(f) | set(w)):
        if want and not any(s in k for s in want):
            continue
        fk, n = f.get(k, (0.0, 0))
        wk, _ = w.get(k, (0.0, 0))
        rows.append({'kernel': k, 'launches': n, 'fetch_size_kib': fk, 'write_size_kib': wk,
                     'hbm_bytes_per_launch': int(round(2 * fk * 1024 + wk * 1024))})
    json.dump({'correction': 'bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024 (gfx950 wide-read under-count, MI355X_MICROARCH.md HBM section)',
               'kernels': rows}, open(out_json, 'w'), indent=1)
    if out_md:
        with open(out_md, 'w') as o:
            o.write('| kernel | launches | FETCH_SIZE KiB (median) | WRITE_SIZE KiB (median) | HBM bytes / launch (2 F + W) |\n|---|---|---|---|---|\n')
            for r in rows:
                o.write(f"| `{r['kernel'][:80]}` | {r['launches']} | {r['fetch_size_kib']:.0f} | {r['write_size_kib']:.0f} | {r['hbm_bytes_per_launch'] / 1e6:.2f} MB |\n")
    for r in rows:
        print(r)


if __name__ == '__main__':
    main()
