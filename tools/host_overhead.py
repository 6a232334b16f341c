#!/usr/bin/env python3
"""Host-side cost of one operator call (Python + ctypes + allocation), measured on tiny problems where the kernels are
shorter than the host path: what eager (non-graph) loops of small launches are bound by."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from haloop_amd import _lib, ops
from haloop_amd._linear import WeightImages, linear
_lib.lib(); _lib.lend_scratch(); _lib.set_math_mode(os.environ.get('HALO_MATH', 'bf16x3'))
x = torch.randn(640, 512, device='cuda'); w = torch.ones(512, device='cuda')
W = torch.nn.Parameter(torch.randn(512, 512, device='cuda'))
img = WeightImages()

def bench(name, fn, n=2000):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'{name}: host {1e6*(t1-t0)/n:.2f} us per call; with the queue drained {1e6*(t2-t0)/n:.2f} us')

bench('layernorm_fwd [640,512]', lambda: ops.layernorm_fwd(x, w))
bench('linear [640,512]x[512,512] (split image + GEMM)', lambda: linear(img, x, W))
bench('linear [64,512]x[512,512] (f32 GEMM + split-K reduce)', lambda: linear(img, x[:64], W))
if '--profile' in sys.argv:
    pr = cProfile.Profile(); pr.enable()
    for _ in range(2000): linear(img, x, W)
    pr.disable()
    pstats.Stats(pr).sort_stats('tottime').print_stats(14)
