#!/usr/bin/env python3
"""Split-GEMM time against K and M on the LSTM's shapes: separates the per-k-tile cost from the fixed cost of a launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from haloop_amd import _lib, ops

_lib.lib(); _lib.lend_scratch()
mode = os.environ.get('HALO_MATH', 'bf16x3'); _lib.set_math_mode(mode)
g = torch.Generator().manual_seed(0)


def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print('math', mode, 'wide', os.environ.get('HALO_GEMM_WIDE', 'auto'))
shapes = [(1024, 4096, 1024), (2048, 4096, 1024), (1280, 4096, 1024)] if os.environ.get('SHORT') else \
    [(1280, 4096, K) for K in (128, 256, 512, 1024, 2048, 4096)] + [(M, 4096, 1024) for M in (256, 512, 1024, 2048, 4096)] + \
    [(4096, 1024, K) for K in (320, 640, 1280, 2560)]
for M, N, K in shapes:
    a = torch.randn(M, K, generator=g).cuda(); b = torch.randn(N, K, generator=g).cuda()
    ai, bi = ops.split_image(a), ops.split_image(b)
    out = torch.empty(M, N, device='cuda')
    t = timeit(lambda: ops.gemm_split(ai, bi, M, N, K, out=out))
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    print(f'M{M} N{N} K{K}: {t:7.1f} us  {2.0 * M * N * K / t / 1e6:6.1f} TF-equiv  tiles128 {tiles}  us/k-tile {t / (K / 32):.3f}')
