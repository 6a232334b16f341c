#!/usr/bin/env python3
"""Recurrent-chain time of one LSTM layer (H=1024, B=64, T'=21): persistent launch vs per-step launches, forward and backward,
HIP events recorded by the library around the chain (halo_lstm_chain_events), interleaved rounds in one process."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from haloop_amd import _lib, ops

T, B, H = 21, 64, 1024
dev = 'cuda'
_lib.set_math_mode(sys.argv[1] if len(sys.argv) > 1 else 'bf16x3')
g = torch.Generator().manual_seed(0)
x = (torch.randn(T, B, H, generator=g) * 0.1).to(dev)
w = [(torch.rand(4 * H, H, generator=g) - 0.5).mul(0.06).to(dev)]
b = [torch.zeros(4 * H, device=dev)]
dy = (torch.randn(B, T, H, generator=g) * 0.01).to(dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); e1.record(); torch.cuda.synchronize()
grads = {k: [torch.zeros_like(w[0])] for k in ('dw_ih', 'dw_hh')}
grads.update({k: [torch.zeros_like(b[0])] for k in ('db_ih', 'db_hh')})


def run(direction, persistent, reps=10):
    _lib.set_lstm_persistent(persistent)
    ts = []
    for i in range(reps + 2):
        if direction == 'fwd':
            _lib.lstm_chain_events(e0, e1)
        y, _, _, reserve = ops.lstm_fwd(x, w, w, b, b)
        if direction == 'bwd':
            _lib.lstm_chain_events(e0, e1)
            ops.lstm_bwd(x, w, w, dy, (H, T * H), False, reserve, grads=grads)
        _lib.lstm_chain_events(None, None)
        torch.cuda.synchronize()
        if i >= 2:
            ts.append(1e3 * e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2], ts[0], _lib.lstm_chain_info(direction)


for rnd in range(3):
    for direction in ('fwd', 'bwd'):
        for persistent in (True, False):
            med, mn, info = run(direction, persistent)
            print(f'round {rnd} {direction} {info["kernel"]:26s} launches {info["launches"]:2d}: median {med:7.1f} us, min {mn:7.1f} us '
                  f'({med / T:.2f} us per step)')
_lib.set_lstm_persistent(True)
