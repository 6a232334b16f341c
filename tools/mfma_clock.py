#!/usr/bin/env python3
"""The clock the chip holds under a bare bf16 MFMA loop on random operands and the rate the matrix pipes sustain there
(halo_debug_mfma_clock; MI355X_MICROARCH.md 'DVFS give-back' item 6: stamped after >= 2 s of back-to-back launches).
    python tools/mfma_clock.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from haloop_amd import _lib
from haloop_amd._lib import check, lib, ptr

lib()
dev = 'cuda'
cus = torch.cuda.get_device_properties(0).multi_processor_count
sink = torch.zeros(1, device=dev)
for wgs_per_cu in (1, 2):
    blocks = cus * wgs_per_cu
    ticks = torch.zeros(2 * blocks, dtype=torch.int64, device=dev)
    for shape, name in ((0, '32x32x16'),):            # the GEMM kernels' instruction
        iters = 200000
        t0 = time.time()
        while time.time() - t0 < 2.5:                       # load the chip before the measured launch
            check(lib().halo_debug_mfma_clock(ptr(ticks), ptr(sink), blocks, iters, shape, 7, None), 'mfma_clock')
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib().halo_debug_mfma_clock(ptr(ticks), ptr(sink), blocks, iters, shape, 11, None), 'mfma_clock')
        e1.record(); torch.cuda.synchronize()
        t = ticks.cpu().numpy().reshape(blocks, 2).astype(np.float64)
        ghz = np.median(t[:, 0] / t[:, 1]) * 0.1
        wall = e0.elapsed_time(e1) * 1e-3
        flop = blocks * 4 * iters * 131072.0
        print(f'{name}, {wgs_per_cu} workgroup(s) of 4 waves per CU: in-kernel clock {ghz:.2f} GHz (min {0.1 * (t[:, 0] / t[:, 1]).min():.2f}, max '
              f'{0.1 * (t[:, 0] / t[:, 1]).max():.2f}); {flop / wall / 1e12:.0f} TFLOP/s over the launch (nominal 2.4 GHz: '
              f'{cus * 4 * 32768 / 32 * 2.4e9 / 1e12:.0f})')
