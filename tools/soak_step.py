"""Soak of the training step's in-launch hand-offs (the one-launch head's tagged exchange, the persistent recurrences' epochs): many
steps on changing batches at several batch sizes; every loss finite, no bounded wait given up, the head's launch counter = the steps run."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from haloop_amd import _lib, rnn, recognizer, synth
from haloop_amd.train import LstmCtcTrainer

_lib.lib()
_lib.set_math_mode('bf16')
F_, C, H, L, V, T, S = 80, 128, 1024, 2, 32, 80, 10
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
for B in (64, 128, 24):
    enc_p, rec_p = synth.make_params(F_, C, H, L, V, 42)
    enc = rnn.Encoder(F_, C, H, num_layers=L); rec = recognizer.TemporalClassifier(H, V)
    enc.load_state_dict(enc_p); rec.load_state_dict(rec_p)
    enc.to('cuda').train(); rec.to('cuda').train()
    batches = [tuple(t.to('cuda') for t in synth.synthetic_batch(B, T, F_, V, S, 1000 + i)) for i in range(8)]
    tr = LstmCtcTrainer(enc, rec, lr=1e-4, use_graph=False)
    t0 = time.time()
    bad = 0
    losses = []
    for i in range(steps):
        losses.append(tr.step(*batches[i % 8]))
        if i % 500 == 499:
            vals = torch.stack(losses).tolist(); losses = []
            bad += sum(1 for v in vals if not (v == v and abs(v) < 1e4))
            tr.check_status()
    torch.cuda.synchronize()
    tr.check_status()
    launches = int(tr._head_ticket[1].item()) if getattr(tr, '_head_ticket', None) is not None else -1
    print(f'B = {B}: {steps} steps in {time.time() - t0:.1f} s, non-finite losses {bad}, head launches counted {launches}, loss ticket {int(tr._head_ticket[0].item())}', flush=True)
    assert bad == 0 and launches == steps
    _lib.set_status_word(None)
print('soak ok')
