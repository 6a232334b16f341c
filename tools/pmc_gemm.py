#!/usr/bin/env python3
"""One GEMM shape launched a few times, for rocprofv3 --pmc runs:  HALO_MATH=bf16x3|bf16 python tools/pmc_gemm.py [M N K]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from haloop_amd import _lib, ops

_lib.lib(); _lib.lend_scratch()
_lib.set_math_mode(os.environ.get('HALO_MATH', 'bf16x3'))
M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (8192, 3072, 768)
a = torch.randn(M, K, device='cuda'); b = torch.randn(N, K, device='cuda')
ai, bi = ops.split_image(a), ops.split_image(b)
out = torch.empty(M, N, device='cuda')
for _ in range(10):
    ops.gemm_split(ai, bi, M, N, K, out=out)
torch.cuda.synchronize()
