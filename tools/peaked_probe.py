"""How many steps on one fixed batch until LC-2x1024's posteriors are peaked on every frame (tests/test_gpu_bf16_contract.py)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from haloop_amd import _lib, rnn, recognizer, synth
from haloop_amd.train import LstmCtcTrainer

_lib.lib()
_lib.set_math_mode('bf16')
F_, C, H, L, V, B, T, S = 80, 128, 1024, 2, 32, 64, 80, 10
for lr, clip in ((1e-3, 1.0), (3e-3, 1.0), (1e-3, 0.1)):
    enc_p, rec_p = synth.make_params(F_, C, H, L, V, 42)
    enc = rnn.Encoder(F_, C, H, num_layers=L); rec = recognizer.TemporalClassifier(H, V)
    enc.load_state_dict(enc_p); rec.load_state_dict(rec_p)
    enc.to('cuda').eval(); rec.to('cuda').eval()
    batch = tuple(t.to('cuda') for t in synth.synthetic_batch(B, T, F_, V, S, 4242))
    tr = LstmCtcTrainer(enc, rec, lr=lr, use_graph=True, clip_grad_norm=clip)
    for i in range(1, 1201):
        loss = tr.step(*batch)
        if i in (1, 100, 200, 300, 400, 600, 800, 1200):
            with torch.no_grad():
                feats, flen, _ = enc(batch[0], batch[1])
                lp = rec.log_probs(feats)
            top2 = torch.topk(lp, 2, dim=-1).values
            m = top2[..., 0] - top2[..., 1]
            print(f'lr {lr} clip {clip} step {i}: loss {loss.item():.4f} min margin {m.min().item():.4f} 1%-quantile {m.flatten().quantile(0.01).item():.3f}',
                  flush=True)
    _lib.set_status_word(None)
