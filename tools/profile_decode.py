#!/usr/bin/env python3
"""Greedy attention decode of `transformer:32` (N=64, T=9) a few times, for `rocprofv3 --kernel-trace --stats -- python3 tools/profile_decode.py`."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from haloop_amd import _lib, transformer
from oracle import transformer_ref as ref

_lib.lib(); _lib.lend_scratch(256 << 20)
_lib.set_math_mode(os.environ.get('HALO_MATH', 'bf16x3'))
N, V, HD, H, L = int(os.environ.get('B', '64')), 32, 64, 8, 12
pd = ref.make_decoder_params(V, HD, H, L, 29)
dec = transformer.CTCAttentionDecoder(vocab=V, head_dim=HD, heads=H, p_drop=0.2, layers=L)
dec.load_state_dict(pd); dec.cuda().eval()
g = torch.Generator().manual_seed(0)
feats = torch.randn(N, 10, H * HD, generator=g).cuda()
flen = torch.full((N,), 10).cuda()
tl = torch.full((N,), 8).cuda()
with torch.inference_mode():
    for _ in range(3): out = dec.decode(feats, flen, tl)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = int(os.environ.get('REPS', '10'))
    for _ in range(n): out = dec.decode(feats, flen, tl)
    torch.cuda.synchronize()
print(f'decode N={N} T=9: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per call; lengths {out[1][:8].tolist()}')
