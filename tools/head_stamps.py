"""Where the two CTC-head launches of the training step spend their time: build with `make -C haloop_amd/csrc EXTRA=-DHALO_HEAD_STAMPS`
(touch head.hip first), run this on the GPU box, rebuild plain afterwards.  Stamps: the 100 MHz constant clock, thread 0 of every workgroup."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from haloop_amd import _lib, rnn, recognizer, synth
from haloop_amd.train import LstmCtcTrainer

L = _lib.lib()
_lib.set_math_mode('bf16')
F_, C_, H, NL, V, B, T, S = 80, 128, 1024, 2, 32, int(os.environ.get('B', 64)), 80, 10
enc_p, rec_p = synth.make_params(F_, C_, H, NL, V, 42)
enc = rnn.Encoder(F_, C_, H, num_layers=NL); rec = recognizer.TemporalClassifier(H, V)
enc.load_state_dict(enc_p); rec.load_state_dict(rec_p)
enc.to('cuda').train(); rec.to('cuda').train()
batch = tuple(t.to('cuda') for t in synth.synthetic_batch(B, T, F_, V, S, 4242))
tr = LstmCtcTrainer(enc, rec, lr=1e-4, use_graph=False, head_one_launch=os.environ.get('ONE', '1') == '1')
for _ in range(5):
    tr.step(*batch)
torch.cuda.synchronize()
fn = C.CDLL(_lib.LIB_PATH).halo_debug_head_stamps
fn.restype = C.c_int; fn.argtypes = [C.c_void_p]
buf = np.zeros((2, 1024, 16), dtype=np.uint64)
assert fn(buf.ctypes.data) == 0
_lib.set_status_word(None)
one = os.environ.get('ONE', '1') == '1'
names = [['entry', 'partial logits', 'K-slices added', 'partials exchanged', 'log-softmax', 'alpha | beta', 'gradient at the logits', 'products'] if one else
         ['entry', 'product done', 'log-softmax done', 'alpha done', 'ticket drawn'],
         ['entry', 'lp / alpha staged', 'beta done', 'dlogits done', 'products done']]
for k, kn in enumerate(('ctc_head_train' if one else 'ctc_head_fwd', 'ctc_head_bwd')):
    if one and k == 1:
        break
    npts = len(names[k])
    st = buf[k, :B, :npts].astype(np.int64)
    t0 = st[:, 0].min()
    print(f'{kn}: workgroup entry spread {10 * (st[:, 0].max() - t0)} ns; last exit {10 * (st[:, npts - 1].max() - t0)} ns after the first entry')
    for i in range(1, npts):
        d = 10 * (st[:, i] - st[:, i - 1])
        print(f'   {names[k][i - 1]:>24} -> {names[k][i]:<24} mean {d.mean():8.0f} ns   min {d.min():6d}   max {d.max():6d}')
