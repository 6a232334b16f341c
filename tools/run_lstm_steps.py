#!/usr/bin/env python3
"""Runs only the LSTM recurrence (H=1024, B=64, T=64 forward + backward) -- the workload the PMC
passes for roofline.traffic are collected on (profiles/README.md has the commands).
LAYERS=2 FUSED=1 runs the two-layer, layer-diagonal fused schedule instead."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from haloop_amd import _lib, ops

_lib.lib(); _lib.lend_scratch(); _lib.set_math_mode(os.environ.get('HALO_MATH', 'bf16x3'))
L = int(os.environ.get('LAYERS', '1'))
_lib.set_lstm_fusion(int(os.environ.get('FUSED', '0')))
T, B, H = int(os.environ.get('T', '64')), 64, 1024
g = torch.Generator().manual_seed(0)
x = (torch.randn(T, B, H, generator=g) * 0.1).cuda()
w = [((torch.rand(4 * H, H, generator=g) - 0.5) * 0.06).cuda() for _ in range(L)]
b = [torch.zeros(4 * H, device='cuda') for _ in range(L)]
for _ in range(3):
    y, _, _, reserve = ops.lstm_fwd(x, w, w, b, b)
    ops.lstm_bwd(x, w, w, torch.ones_like(y), (B * H, H), False, reserve, want_dx=True)
torch.cuda.synchronize()
print('ok')
