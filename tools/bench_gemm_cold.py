#!/usr/bin/env python3
"""The bf16 product with its A operand from a tiled image vs from row-major bf16 rows (halo_gemm_split_io), each call behind a 1 GiB
write that empties L2 / MALL: the row-major staging fetches 64 bytes per row and k-tile and is 10-15 % slower from cold caches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from haloop_amd import _lib, ops
_lib.lib(); _lib.lend_scratch(256 << 20); _lib.set_math_mode('bf16')
flush = torch.empty(1 << 28, device='cuda', dtype=torch.float32)     # 1 GiB

def us_cold(fn, reps=12):
    ts = []
    for _ in range(reps):
        flush.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(1e3 * e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]

for M, N, K in [(8192, 3072, 768), (8192, 768, 3072), (8192, 768, 768), (8192, 2304, 768)]:
    x = torch.randn(M, K, device='cuda'); w = torch.randn(N, K, device='cuda') * K ** -0.5
    xi, wi, xb = ops.split_image(x), ops.split_image(w), ops.cast_bf16(x)
    out = torch.empty(M, N, device='cuda')
    a = us_cold(lambda: ops.gemm_split(xi, wi, M, N, K, out=out))
    b = us_cold(lambda: ops.gemm_split_io((xb, None), wi, M, N, K, out=out))
    print(f'[{M} x {N} x {K}] cold: image A {a:6.1f} us   row-major A {b:6.1f} us')
