#!/usr/bin/env python3
"""Where a combined step of the two-layer persistent LSTM forward (csrc/lstm_persist2.hip) spends its time: in-kernel 100 MHz
stamps of every workgroup, H=1024 B=64 T=21 L=2, bf16 mode, averaged over the inner steps and over workgroups.  Diagnostic only.
    python tools/persist2_stamps.py [p_drop]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from haloop_amd import _lib, ops

T, B, H, L = 21, 64, 1024, 2
dev = 'cuda'
_lib.set_math_mode('bf16')
p_drop = float(sys.argv[1]) if len(sys.argv) > 1 else 0.2
g = torch.Generator().manual_seed(0)
x = (torch.randn(T, B, 128, generator=g) * 0.5).to(dev)
w_ih = [(torch.rand(4 * H, 128 if l == 0 else H, generator=g) - 0.5).mul(0.06).to(dev) for l in range(L)]
w_hh = [(torch.rand(4 * H, H, generator=g) - 0.5).mul(0.06).to(dev) for l in range(L)]
b = [torch.zeros(4 * H, device=dev) for l in range(L)]
drop = ops.Dropout(p_drop, 1, 0) if p_drop > 0 else ops.NO_DROPOUT
nblk = (H // 16) * ((B + 15) // 16)
S = T + 2
stamps = torch.zeros(nblk * S * 16, dtype=torch.int64, device=dev)
assert _lib.lib().halo_lstm_persistent2_eligible(T, B, H, L) == 1
for _ in range(3):
    ops.lstm_fwd(x, w_ih, w_hh, b, b, drop=drop)
_lib.check(_lib.lib().halo_lstm_persist_stamps(stamps.data_ptr()), 'stamps')
ops.lstm_fwd(x, w_ih, w_hh, b, b, drop=drop)
torch.cuda.synchronize()
_lib.lib().halo_lstm_persist_stamps(None)
s = stamps.cpu().numpy().reshape(nblk, S, 16).astype(np.float64) * 0.01     # microseconds
names = {0: 'loop top (wave 0)', 8: 'loop top (wave 4)', 1: 'after barrier A: poll matched', 2: 'after barrier B: MFMAs done, partials written',
         3: 'after barrier C: cell updates done', 4: 'layer 0 piece drained (+ flag), wave 3', 5: 'layer 1 piece drained (+ flag), wave 7',
         7: 'input-half fragments requested (wave 4)', 6: 'input half done (wave 4)'}
steps = slice(3, T - 1)
base = s[:, steps, 0]
print(f'kernel span (first loop top to last): {(s[:, -1, 0].max() - s[:, 0, 0].min()):.1f} us')
print(f'per combined step (loop top to loop top), mean over workgroups: {np.diff(s[:, 2:T, 0], axis=1).mean():.3f} us')
print(f'workgroup entry -> first loop top (weights into registers / LDS): mean {(s[:, 0, 0] - s[:, 0, 14]).mean():.2f} us, max {(s[:, 0, 0] - s[:, 0, 14]).max():.2f};'
      f' entry skew {s[:, 0, 14].max() - s[:, 0, 14].min():.2f}; last loop top -> exit: mean {(s[:, 0, 15] - s[:, -1, 0]).mean():.2f}; first entry -> last exit {s[:, 0, 15].max() - s[:, 0, 14].min():.1f} us')
for k in (0, 8, 1, 2, 3, 4, 5, 7, 6):
    d = s[:, steps, k] - base
    print(f'  {names[k]:52s} +{d.mean():7.3f} us  (min {d.min():6.2f}  max {d.max():6.2f})')

# ---- backward ----
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
Sb = T + 1
s = stamps.cpu().numpy()[:nblk * Sb * 16].reshape(nblk, Sb, 16).astype(np.float64) * 0.01
names = {0: 'loop top (wave 0)', 1: 'after barrier A: poll matched', 2: 'after barrier B: MFMAs done, partials written',
         3: 'after barrier C: cell updates done', 4: 'pieces drained (+ flag), wave 3'}
steps = slice(3, T - 1)
base = s[:, steps, 0]
print(f'BACKWARD kernel span (first loop top to last): {(s[:, -1, 0].max() - s[:, 0, 0].min()):.1f} us')
print(f'per combined step (loop top to loop top), mean over workgroups: {np.diff(s[:, 2:T, 0], axis=1).mean():.3f} us')
print(f'workgroup entry -> first loop top (weights into registers / LDS): mean {(s[:, 0, 0] - s[:, 0, 14]).mean():.2f} us, max {(s[:, 0, 0] - s[:, 0, 14]).max():.2f};'
      f' entry skew {s[:, 0, 14].max() - s[:, 0, 14].min():.2f}; last loop top -> exit: mean {(s[:, 0, 15] - s[:, -1, 0]).mean():.2f}; first entry -> last exit {s[:, 0, 15].max() - s[:, 0, 14].min():.1f} us')
for k in (0, 1, 2, 3, 4):
    d = s[:, steps, k] - base
    print(f'  {names[k]:52s} +{d.mean():7.3f} us  (min {d.min():6.2f}  max {d.max():6.2f})')
