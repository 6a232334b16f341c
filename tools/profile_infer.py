#!/usr/bin/env python3
"""Forward + greedy decode of the benchmark model (LstmCtcRecognizer, eval, B=64) for a rocprofv3 kernel trace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from haloop_amd import _lib, rnn, recognizer, synth
from haloop_amd.infer import LstmCtcRecognizer
_lib.lib(); _lib.lend_scratch(); _lib.set_math_mode(os.environ.get('HALO_MATH', 'bf16'))
enc_p, rec_p = synth.make_params(80, 128, 1024, 2, 32, 42)
enc = rnn.Encoder(80, 128, 1024, num_layers=2); rec = recognizer.TemporalClassifier(1024, 32)
enc.load_state_dict(enc_p); rec.load_state_dict(rec_p)
reco = LstmCtcRecognizer(enc.cuda(), rec.cuda())
x = synth.synthetic_batch(64, 80, 80, 32, 10, 42)[0].cuda()
for _ in range(30):
    reco.recognize(x, clone=False)
torch.cuda.synchronize()
print('ok')
