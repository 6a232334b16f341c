#!/usr/bin/env python3
"""One product shape launched a few times, for rocprofv3 --pmc runs:  HALO_MATH=bf16x3|bf16 python tools/pmc_gemm.py [M N K]
In bf16 mode the launch is the GPT path's own (halo_gemm_rows: row-major bf16 activations, weight image, bf16 result)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from haloop_amd import _lib, ops

_lib.lib(); _lib.lend_scratch()
_lib.set_math_mode(os.environ.get('HALO_MATH', 'bf16x3'))
M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (8192, 3072, 768)
a = torch.randn(M, K, device='cuda'); b = torch.randn(N, K, device='cuda')
bi = ops.split_image(b)
if ops.gemm_rows_supported(M, N, K):
    ab = a.bfloat16()
    for _ in range(10):
        ops.gemm_rows(ab, bi, M, N, K, out_bf16=True)
else:
    ai = ops.split_image(a)
    out = torch.empty(M, N, device='cuda')
    for _ in range(10):
        ops.gemm_split(ai, bi, M, N, K, out=out)
torch.cuda.synchronize()
