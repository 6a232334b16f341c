#!/usr/bin/env python3
"""The TN product (halo_gemm_tn_bf16) against the image path it replaces, on the GPT-2 small weight-gradient shapes (bf16 mode)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from haloop_amd import _lib, ops
_lib.lib(); _lib.lend_scratch(256 << 20); _lib.set_math_mode('bf16')


def us(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


for K, M, N in [(8192, 768, 3072), (8192, 3072, 768), (8192, 768, 768), (8192, 2304, 768), (8192, 50304, 768), (1344, 4096, 2048)]:
    dy = torch.randn(K, M, device='cuda'); x = torch.randn(K, N, device='cuda')
    dyb, xb = dy.bfloat16(), x.bfloat16()
    t_tn = us(lambda: ops.gemm_tn(dyb, xb))
    ia, ib = ops.split_image(dy, transposed=True), ops.split_image(x, transposed=True)
    t_img = us(lambda: ops.gemm_split(ia, ib, M, N, K))
    t_prep = us(lambda: (ops.split_image(dy, transposed=True), ops.split_image(x, transposed=True)))
    tf = 2.0 * K * M * N / 1e6
    print(f'K={K} M={M} N={N}: TN {t_tn:7.1f} us ({tf / t_tn:6.0f} TF)   image product {t_img:7.1f} us ({tf / t_img:6.0f} TF)   + its two transposed images {t_prep:6.1f} us')
