#!/usr/bin/env python3
"""The bench's training step at another batch size (B from the environment), eager launches, for kernel traces: python tools/run_step_b.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from haloop_amd import _lib, synth
from haloop_amd.train import LstmCtcTrainer
_lib.lib(); _lib.set_math_mode(os.environ.get('HALO_MATH', 'bf16'))
B = int(os.environ.get('B', '128'))
dev = torch.device('cuda', 0)
enc, rec, _ = bench.build_model(dev)
tr = LstmCtcTrainer(enc, rec, seed=1337, use_graph=False, alias_loss=True)
batch = tuple(t.to(dev) for t in synth.synthetic_batch(B, bench.T, bench.F, bench.V, bench.S, 42))
for _ in range(30):
    tr.step(*batch)
torch.cuda.synchronize()
tr.check_status()
print('ok', float(tr.loss))
