"""A few training steps at batch size B (argv[1]) for a kernel trace: BENCH_CMD="tools/step_b.py 256" bash tools/timeline_once.sh"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from haloop_amd import _lib, rnn, recognizer, synth
from haloop_amd.train import LstmCtcTrainer

_lib.lib()
_lib.set_math_mode('bf16')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
F_, C, H, L, V, T, S = 80, 128, 1024, 2, 32, 80, 10
enc_p, rec_p = synth.make_params(F_, C, H, L, V, 42)
enc = rnn.Encoder(F_, C, H, num_layers=L); rec = recognizer.TemporalClassifier(H, V)
enc.load_state_dict(enc_p); rec.load_state_dict(rec_p)
enc.to('cuda').train(); rec.to('cuda').train()
batch = tuple(t.to('cuda') for t in synth.synthetic_batch(B, T, F_, V, S, 7))
tr = LstmCtcTrainer(enc, rec, lr=1e-4, use_graph=False)
for _ in range(40):
    tr.step(*batch)
torch.cuda.synchronize()
tr.check_status()
_lib.set_status_word(None)
