import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from haloop_amd import _lib, rnn, recognizer, synth
from haloop_amd.infer import LstmCtcRecognizer
_lib.lib(); _lib.lend_scratch(); _lib.set_math_mode('bf16')
enc_p, rec_p = synth.make_params(80, 128, 1024, 2, 32, 42)
enc = rnn.Encoder(80, 128, 1024, num_layers=2); rec = recognizer.TemporalClassifier(1024, 32)
enc.load_state_dict(enc_p); rec.load_state_dict(rec_p)
x = synth.synthetic_batch(64, 80, 80, 32, 10, 42)[0].cuda()
for graph in (True, False, True, False):
    reco = LstmCtcRecognizer(enc.cuda(), rec.cuda(), use_graph=graph)
    for _ in range(30): reco.recognize(x, clone=False)
    xs = reco.static_input() if graph else x
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(500): reco.recognize(xs, clone=False)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f'graph={graph}: host {1e6*(t1-t0)/500:.1f} us, wall {1e6*(t2-t0)/500:.1f} us per batch', flush=True)
