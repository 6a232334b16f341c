#!/usr/bin/env python3
"""Where a step of the persistent LSTM forward spends its time: in-kernel 100 MHz stamps (halo_lstm_persist_stamps) of every
workgroup, H=1024 B=64 T=21, averaged over steps 2..T-1 and over workgroups.  Diagnostic only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from haloop_amd import _lib, ops

T, B, H = 21, 64, 1024
dev = 'cuda'
_lib.set_math_mode(sys.argv[1] if len(sys.argv) > 1 else 'bf16x3')
g = torch.Generator().manual_seed(0)
x = (torch.randn(T, B, H, generator=g) * 0.1).to(dev)
w = [(torch.rand(4 * H, H, generator=g) - 0.5).mul(0.06).to(dev)]
b = [torch.zeros(4 * H, device=dev)]
nblk = (H // 16) * ((B + 15) // 16)
stamps = torch.zeros(nblk * T * 16, dtype=torch.int64, device=dev)
pad = int(os.environ.get('PAD_KB', '0'))
keep = torch.empty(pad * 1024, dtype=torch.uint8, device=dev) if pad else None       # moves the next allocations
for _ in range(3):
    ops.lstm_fwd(x, w, w, b, b)
_lib.check(_lib.lib().halo_lstm_persist_stamps(stamps.data_ptr()), 'stamps')
res = ops.lstm_fwd(x, w, w, b, b)[3]
torch.cuda.synchronize()
fva = res.data_ptr() + _lib.lib().halo_lstm_status_offset(0, T, B, H, H, 1)
print(f'flags VA {fva:#x}  page index mod 8 = {(fva >> 12) & 7}')
_lib.lib().halo_lstm_persist_stamps(None)
raw = stamps.cpu().numpy().reshape(nblk, T, 16)
xcc = raw[:, 0, 15]
s = raw.astype(np.float64) * 0.01     # microseconds
names = {0: 'loop top (wave 0)', 1: 'after barrier A', 2: 'MFMAs done + partials written', 3: 'after barrier B', 4: 'cell update done',
         5: 'after barrier C (wave 4)', 6: 'sc1 stores drained (wave 4)', 7: 'flag stored (wave 4)', 8: 'poll start (wave 5)',
         9: 'poll matched (wave 5)', 10: 'MFMAs done (wave 7)'}
steps = slice(2, T - 1)
base = s[:, steps, 0]
print(f'per-step period (loop top to loop top), mean over workgroups: {np.diff(s[:, 1:, 0], axis=1).mean():.3f} us')
for k in sorted(names):
    d = s[:, steps, k] - base
    print(f'  {names[k]:36s} +{d.mean():7.3f} us  (min {d.min():6.2f}  max {d.max():6.2f})')
# hand-off latency per batch group: from the group's LAST flag store of step t to its members' poll match (wave 5) at step t+1
NBT = (B + 15) // 16
bid = np.arange(nblk)
grp = (bid & 7) // (8 // NBT) if 8 % NBT == 0 and nblk % 8 == 0 else bid % NBT
for gi in range(NBT):
    m = grp == gi
    pub = s[m][:, :-1, 7]
    match = s[m][:, 1:, 9]
    last_pub, first_pub = pub.max(axis=0), pub.min(axis=0)
    print(f'group {gi}: XCC ids {sorted(set(xcc[m].tolist()))} polled copies {np.bincount(raw[m][:, 0, 14], minlength=8).tolist()}')
    print(f'group {gi}: publish skew (last - first flag store) {(last_pub - first_pub)[1:].mean():.3f} us; poll match (wave 5) after the '
          f"group's last flag store {(match - last_pub[None, :])[:, 1:].mean():.3f} us (max {(match - last_pub[None, :])[:, 1:].max():.3f}); "
          f'period {np.diff(s[m][:, 1:, 0], axis=1).mean():.3f} us')
