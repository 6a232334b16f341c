#!/usr/bin/env python3
"""Calibrates rocprofv3's FETCH_SIZE per load width on this GPU (MI355X_MICROARCH.md, HBM section: "other access widths are uncalibrated").

    python tools/pmc_calibrate.py            # parent: runs itself under `rocprofv3 --pmc FETCH_SIZE`, prints / writes the factors
    python tools/pmc_calibrate.py --child    # the profiled part: known-size read launches (halo_debug_read)

Each launch reads a 1 GiB buffer exactly once with one access shape; factor = bytes read / (FETCH_SIZE x 1024).  Buffers larger than the
256 MiB Infinity Cache and written by a different launch, so nothing is served from a cache."""
import os, sys, subprocess, shutil, tempfile, csv, statistics, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BYTES = 1 << 30
NAMES = {0: '16 B per lane, global_load_dwordx4', 1: '4 B per lane, 256 B contiguous per wave instruction',
         2: '4 B per lane, 64-byte segments in four rows 16 KiB apart (saved-activation loads)', 3: '16 B per lane, buffer_load_dwordx4 sc1 (fragment loads)'}


def child():
    import torch
    from haloop_amd import _lib
    lib = _lib.lib()
    buf = torch.empty(BYTES // 4, device='cuda', dtype=torch.float32)
    other = torch.empty(BYTES // 4, device='cuda', dtype=torch.float32)
    sink = torch.zeros(4, device='cuda')
    for rep in range(5):
        for pattern in range(4):
            buf.uniform_()                   # rewritten, then a second gigabyte streamed: nothing of buf is left in any cache
            other.uniform_()
            _lib.check(lib.halo_debug_read(buf.data_ptr(), BYTES, pattern, 16384, sink.data_ptr(), torch.cuda.current_stream().cuda_stream), 'halo_debug_read')
    torch.cuda.synchronize()


def main():
    exe = shutil.which('rocprofv3')
    if not exe:
        sys.exit('rocprofv3 not found')
    work = tempfile.mkdtemp(prefix='halo_cal_', dir='/tmp')
    proc = subprocess.run([exe, '--pmc', 'FETCH_SIZE', '--kernel-trace', '--output-format', 'csv', '-d', work, '--', sys.executable,
                           os.path.abspath(__file__), '--child'], cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'), capture_output=True, text=True, timeout=600)
    csvs = [os.path.join(d, f) for d, _, fs in os.walk(work) for f in fs if f.endswith('counter_collection.csv')]
    if proc.returncode != 0 or not csvs:
        sys.exit(f'rocprofv3 failed: {proc.returncode}\n{proc.stderr[-2000:]}')
    per = {}
    for r in csv.DictReader(open(csvs[0])):
        if r['Counter_Name'] == 'FETCH_SIZE' and 'debug_read_kernel' in r['Kernel_Name']:
            pat = int(r['Kernel_Name'].split('<')[1].split('>')[0])
            per.setdefault(pat, []).append(float(r['Counter_Value']))
    shutil.rmtree(work, ignore_errors=True)
    out = {}
    print('| access shape | FETCH_SIZE KiB (median of 5) | bytes read / (FETCH_SIZE x 1024) |\n|---|---|---|')
    for pat in sorted(per):
        med = statistics.median(per[pat])
        out[pat] = BYTES / (med * 1024)
        print(f'| {NAMES[pat]} | {med:.0f} | {out[pat]:.3f} |')
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    json.dump({'bytes_read': BYTES, 'factor_by_pattern': out, 'patterns': NAMES}, open(os.path.join(ROOT, 'gpurun_out', 'fetch_size_calibration.json'), 'w'), indent=1)


if __name__ == '__main__':
    child() if '--child' in sys.argv else main()
