import os, sys, time, cProfile, pstats
sys.path.insert(0, os.getcwd())
import torch
from haloop_amd import _lib, rnn, recognizer, synth
from haloop_amd.train import LstmCtcTrainer
_lib.lib(); _lib.lend_scratch(); _lib.set_math_mode('bf16')
enc_p, rec_p = synth.make_params(80, 128, 1024, 2, 32, 42)
enc = rnn.Encoder(80, 128, 1024, num_layers=2); rec = recognizer.TemporalClassifier(1024, 32)
enc.load_state_dict(enc_p); rec.load_state_dict(rec_p)
enc.cuda().train(); rec.cuda().train()
x, il, tg, tl = [t.cuda() for t in synth.synthetic_batch(64, 80, 80, 32, 10, 42)]
tr = LstmCtcTrainer(enc, rec, seed=1, use_graph=False, alias_loss=True)
for _ in range(20): tr.step(x, il, tg, tl)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(300): tr.step(x, il, tg, tl)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(28)
