#!/usr/bin/env python3
"""Runs only the 2-layer LSTM stack of the benchmark (in 128, H=1024, B=64, T'=21, dropout 0.2) forward + backward -- the workload the
PMC passes for roofline.traffic are collected on (profiles/README.md has the commands).  HALO_MATH selects the arithmetic (default
bf16: the two-layer persistent launches, csrc/lstm_persist2.hip)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from haloop_amd import _lib, ops

_lib.lib(); _lib.lend_scratch(); _lib.set_math_mode(os.environ.get('HALO_MATH', 'bf16'))
T, B, H, IN, L = int(os.environ.get('T', '21')), int(os.environ.get('B', '64')), 1024, 128, 2
g = torch.Generator().manual_seed(0)
x = (torch.randn(T, B, IN, generator=g) * 0.5).cuda()
w_ih = [((torch.rand(4 * H, IN if l == 0 else H, generator=g) - 0.5) * 0.06).cuda() for l in range(L)]
w_hh = [((torch.rand(4 * H, H, generator=g) - 0.5) * 0.06).cuda() for l in range(L)]
b = [torch.zeros(4 * H, device='cuda') for _ in range(L)]
drop = ops.Dropout(0.2, 1, 0)
dy = (torch.randn(B, T, H, generator=g) * 0.01).cuda()
for _ in range(3):
    y, _, _, reserve = ops.lstm_fwd(x, w_ih, w_hh, b, b, drop=drop)
    ops.lstm_bwd(x, w_ih, w_hh, dy, (H, T * H), False, reserve, want_dx=True, drop=drop)
torch.cuda.synchronize()
print('ok', _lib.lstm_chain_info('fwd'), _lib.lstm_chain_info('bwd'))
