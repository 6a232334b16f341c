#!/usr/bin/env python3
"""Where a super-step of the interleaved two-layer launches (csrc/lstm_persist2x.hip: two batch tiles per workgroup) spends its time:
in-kernel 100 MHz stamps of every workgroup, H=1024 B=128 T=21 L=2, bf16 mode, averaged over the inner steps and over workgroups.
    python tools/persist2x_stamps.py [p_drop]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from haloop_amd import _lib, ops

T, B, H, L = 21, 128, 1024, 2
dev = 'cuda'
_lib.set_math_mode('bf16')
p_drop = float(sys.argv[1]) if len(sys.argv) > 1 else 0.2
g = torch.Generator().manual_seed(0)
x = (torch.randn(T, B, 128, generator=g) * 0.5).to(dev)
w_ih = [(torch.rand(4 * H, 128 if l == 0 else H, generator=g) - 0.5).mul(0.06).to(dev) for l in range(L)]
w_hh = [(torch.rand(4 * H, H, generator=g) - 0.5).mul(0.06).to(dev) for l in range(L)]
b = [torch.zeros(4 * H, device=dev) for l in range(L)]
drop = ops.Dropout(p_drop, 1, 0) if p_drop > 0 else ops.NO_DROPOUT
nblk = 256
NAMES = ['tile A: loop top', 'tile A: after barrier A (poll matched)', 'tile A: after barrier B (MFMAs done)', 'tile A: after barrier C (cells done)',
         'tile A: pieces drained + flag (wave 3)', 'tile B: loop top', 'tile B: after barrier A (poll matched)', 'tile B: after barrier B (MFMAs done)',
         'tile B: after barrier C (cells done)', 'tile B: pieces drained + flag (wave 3)']
POINTS = [0, 1, 2, 3, 4, 9, 10, 11, 12, 13]


def report(s, S, label):
    steps = slice(3, T - 1)
    base = s[:, steps, 0]
    print(f'{label}: kernel span (first loop top to last) {(s[:, -1, 0].max() - s[:, 0, 0].min()):.1f} us; per super-step (both tiles) '
          f'{np.diff(s[:, 2:T, 0], axis=1).mean():.3f} us; entry -> first loop top {(s[:, 0, 0] - s[:, 0, 14]).mean():.2f} us; '
          f'last loop top -> exit {(s[:, 0, 15] - s[:, -1, 0]).mean():.2f} us; first entry -> last exit {s[:, 0, 15].max() - s[:, 0, 14].min():.1f} us')
    prev = None
    for k, name in zip(POINTS, NAMES):
        d = s[:, steps, k] - base
        print(f'  {name:46s} +{d.mean():7.3f} us  (min {d.min():6.2f}  max {d.max():6.2f})' + (f'   [{d.mean() - prev:+.3f}]' if prev is not None else ''))
        prev = d.mean()


assert _lib.lib().halo_lstm_persistent2_eligible(T, B, H, L) == 1
stamps = torch.zeros(nblk * (T + 2) * 16, dtype=torch.int64, device=dev)
for _ in range(3):
    ops.lstm_fwd(x, w_ih, w_hh, b, b, drop=drop)
_lib.check(_lib.lib().halo_lstm_persist_stamps(stamps.data_ptr()), 'stamps')
ops.lstm_fwd(x, w_ih, w_hh, b, b, drop=drop)
torch.cuda.synchronize()
_lib.lib().halo_lstm_persist_stamps(None)
report(stamps.cpu().numpy().reshape(nblk, T + 2, 16).astype(np.float64) * 0.01, T + 2, 'FORWARD')

dy = (torch.randn(T, B, H, generator=g) * 0.01).to(dev)
y, _, _, reserve = ops.lstm_fwd(x, w_ih, w_hh, b, b, drop=drop)
ws = ops.lstm_bwd_workspace(x, w_hh)
for _ in range(2):
    ops.lstm_bwd(x, w_ih, w_hh, dy, (B * H, H), False, reserve, drop=drop, workspace=ws)
stamps.zero_()
_lib.check(_lib.lib().halo_lstm_persist_stamps(stamps.data_ptr()), 'stamps')
ops.lstm_bwd(x, w_ih, w_hh, dy, (B * H, H), False, reserve, drop=drop, workspace=ws)
torch.cuda.synchronize()
_lib.lib().halo_lstm_persist_stamps(None)
sb = stamps.cpu().numpy()[:nblk * (T + 1) * 16].reshape(nblk, T + 1, 16).astype(np.float64) * 0.01
report(sb, T + 1, 'BACKWARD')
steps = slice(3, T - 1)
for k, name in ((5, 'tile A: layer-1 wave 4 past barrier B'), (6, 'tile A: layer-1 wave 4 arrives at barrier C'), (7, 'tile A: layer-0 wave 0 arrives at barrier C')):
    d = sb[:, steps, k] - sb[:, steps, 0]
    print(f'  {name:46s} +{d.mean():7.3f} us  (min {d.min():6.2f}  max {d.max():6.2f})')
