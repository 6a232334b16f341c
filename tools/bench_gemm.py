#!/usr/bin/env python3
"""Micro-benchmark of the GEMM kernels on the LSTM's shapes (run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from haloop_amd import _lib, ops

_lib.lib(); _lib.lend_scratch()
dev = 'cuda'
g = torch.Generator().manual_seed(0)

def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

mode = os.environ.get('HALO_MATH', 'bf16x3'); _lib.set_math_mode(mode)
print('math', mode, 'ring', os.environ.get('HALO_GEMM_STAGES', 'auto'), os.environ.get('HALO_GEMM_STAGES_BF16', '3'))
print('wide', os.environ.get('HALO_GEMM_WIDE', 'auto'))
for M, N, K in [(1280, 4096, 1024), (1280, 1024, 4096), (4096, 2048, 1280), (4096, 1024, 1280), (1344, 4096, 1024), (4096, 1024, 1344), (1344, 1024, 4096), (12800, 4096, 1024), (8192, 3072, 768), (8192, 768, 3072),
                (8192, 2304, 768), (8192, 768, 768), (3072, 768, 8192), (8300, 50304, 768), (700, 4096, 512)]:
    a = torch.randn(M, K, generator=g).to(dev); b = torch.randn(N, K, generator=g).to(dev)
    ai, bi = ops.split_image(a), ops.split_image(b)
    out = torch.empty(M, N, device=dev)
    t_split = timeit(lambda: ops.gemm_split(ai, bi, M, N, K, out=out))
    t_prep = timeit(lambda: ops.split_image(a))
    t_f32 = 0.0 if M * N > 3e8 else timeit(lambda: ops.gemm(a, b, True, True, M, N, K, out=out), n=5)
    ref = a.double() @ b.double().t()
    ops.gemm_split(ai, bi, M, N, K, out=out)
    err = (out.double() - ref).abs().max().item() / ref.abs().max().item()
    fl = 2.0 * M * N * K
    print(f'M{M} N{N} K{K}: split {t_split:7.1f} us ({fl / t_split / 1e6:6.1f} TF-equiv)  prepA {t_prep:6.1f} us  f32 {t_f32:7.1f} us '
          f'({fl / max(t_f32, 1e-9) / 1e6:6.1f} TF)  rel.err {err:.2e}')
