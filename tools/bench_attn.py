#!/usr/bin/env python3
"""Times the attention kernels alone on the GPT-2 small shape (B=8, T=1024, 12 heads x 64, causal, packed qkv rows).

    HALO_MATH=bf16x3|bf16|f32 python tools/bench_attn.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from haloop_amd import _lib, ops

_lib.lib()
_lib.set_math_mode(os.environ.get('HALO_MATH', 'bf16x3'))
B = int(os.environ.get('B', '8')); T = int(os.environ.get('T', '1024')); H = 12; hd = 64; C = H * hd
torch.manual_seed(0)
qkv = torch.randn(B * T, 3 * C, device='cuda')
q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
dy = torch.randn(B * T, C, device='cuda')
dqkv = torch.empty_like(qkv)
dq, dk, dv = dqkv[:, :C], dqkv[:, C:2 * C], dqkv[:, 2 * C:]


def event_us(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


y, lse, _ = ops.attention_fwd(q, k, v, B, H, hd, T, T, causal=True, want_lse=True)
t_f = event_us(lambda: ops.attention_fwd(q, k, v, B, H, hd, T, T, causal=True, want_lse=True))
t_b = event_us(lambda: ops.attention_bwd(q, k, v, y, dy, lse, dq, dk, dv, B, H, hd, T, T, causal=True))
flops = 4.0 * B * H * T * T * hd / 2
print(f'attention B={B} T={T} H={H} hd={hd} causal math={_lib.get_math_mode()}: fwd {t_f:.1f} us ({flops / t_f / 1e6:.1f} TFLOP/s)  '
      f'bwd {t_b:.1f} us ({2.5 * flops / t_b / 1e6:.1f} TFLOP/s)  checksum {y.double().sum().item():.6f} {dqkv.double().abs().sum().item():.4f}')

if _lib.get_math_mode() == 'bf16':
    # the same shape from row-major bf16 q | k | v (csrc/attn_b16.hip: what the GPT blocks run in bf16 arithmetic)
    qkvb = qkv.bfloat16()
    qb, kb, vb = qkvb[:, :C], qkvb[:, C:2 * C], qkvb[:, 2 * C:]
    dyb = dy.bfloat16()
    dqkvb = torch.empty_like(qkvb)
    _, lse_b, yb = ops.attention_fwd_b16(qb, kb, vb, B, H, hd, T, T, causal=True, want_lse=True)
    t_f = event_us(lambda: ops.attention_fwd_b16(qb, kb, vb, B, H, hd, T, T, causal=True, want_lse=True))
    t_b = event_us(lambda: ops.attention_bwd_b16(qb, kb, vb, yb, dyb, lse_b, dqkvb[:, :C], dqkvb[:, C:2 * C], dqkvb[:, 2 * C:], B, H, hd, T, T, causal=True))
    print(f'  from bf16 rows: fwd {t_f:.1f} us ({flops / t_f / 1e6:.1f} TFLOP/s)  bwd {t_b:.1f} us ({2.5 * flops / t_b / 1e6:.1f} TFLOP/s)  '
          f'checksum {yb.double().sum().item():.4f} {dqkvb.double().abs().sum().item():.2f}')
