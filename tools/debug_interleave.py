"""Debug aid: interleaved two-tile launches vs consecutive launches, per-output differences; repeated runs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from haloop_amd import _lib, ops
import test_gpu_lstm_b64 as tt

_lib.lib(); _lib.lend_scratch(); _lib.set_math_mode('bf16')
hal = dict(ops=ops, lib=_lib)
for (T, B, in0, H, p, st) in ((1, 128, 128, 1024, 0.0, False), (2, 128, 128, 1024, 0.0, False), (4, 128, 128, 1024, 0.0, False), (4, 128, 128, 1024, 0.2, True)):
    outs = []
    for mode in (True, True, False, False):
        _lib.set_lstm_interleave(mode)
        a, st_a = tt._lstm_case(hal, T, B, in0, H, 2, p, 5, st)
        outs.append(a)
    _lib.set_lstm_interleave(True)
    print(f'--- T={T} B={B} H={H} p={p} state={st}')
    for k in outs[0]:
        d_rep = (outs[0][k] - outs[1][k]).abs().max().item()
        d_rep2 = (outs[2][k] - outs[3][k]).abs().max().item()
        d = (outs[0][k] - outs[2][k]).abs().max().item()
        sc = outs[2][k].abs().max().item()
        bad = (outs[0][k] != outs[2][k])
        where = ''
        if bad.any() and outs[0][k].dim() == 3:
            idx = bad.nonzero()
            where = f' first bad idx {idx[0].tolist()} last {idx[-1].tolist()} count {bad.sum().item()} of {bad.numel()}; batch rows {sorted(set(idx[:, 1].tolist()))[:20]}'
        print(f'  {k:8s} il-vs-plain {d:.3e} (scale {sc:.2e})  repeat-il {d_rep:.1e} repeat-plain {d_rep2:.1e}{where}')
