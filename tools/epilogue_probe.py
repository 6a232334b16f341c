"""Epilogue cost of the tiled split-bf16 GEMM on the GPT shapes: plain store / + bias / + bias + residual add / + bias + GELU.
    HALO_MATH=bf16 python tools/epilogue_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from haloop_amd import _lib, ops
_lib.lib(); _lib.lend_scratch(256 << 20); _lib.set_math_mode(os.environ.get('HALO_MATH', 'bf16'))
g = torch.Generator().manual_seed(0)


def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for M, N, K in [(8192, 768, 768), (8192, 768, 3072), (8192, 3072, 768), (8192, 2304, 768)]:
    a = torch.randn(M, K, generator=g).cuda(); b = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    bias = torch.randn(N, generator=g).cuda()
    ai, bi = ops.split_image(a), ops.split_image(b)
    out = torch.zeros(M, N, device='cuda')
    t = [timeit(lambda: ops.gemm_split(ai, bi, M, N, K, out=out)),
         timeit(lambda: ops.gemm_split(ai, bi, M, N, K, out=out, bias1=bias)),
         timeit(lambda: ops.gemm_split(ai, bi, M, N, K, out=out, bias1=bias, accumulate=True)),
         timeit(lambda: ops.gemm_split(ai, bi, M, N, K, out=out, bias1=bias, gelu=True))]
    fl = 2 * M * N * K
    print(f'{os.environ.get("HALO_MATH", "bf16")} {M}x{N}x{K}: plain {t[0]:.1f} us ({fl / t[0] / 1e9:.2f} PF)  bias {t[1]:.1f}  bias+residual {t[2]:.1f}  bias+gelu {t[3]:.1f}')
