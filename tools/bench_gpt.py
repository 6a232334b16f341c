#!/usr/bin/env python3
"""GPT-2 small scoring throughput (BASELINE config 3 shape): tokens/s of forward_all(reduction='none')."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from haloop_amd import _lib, attention
from oracle import gpt_ref

_lib.lib(); _lib.lend_scratch(256 << 20)
math = os.environ.get('HALO_MATH', 'bf16x3'); _lib.set_math_mode(math)
B = int(os.environ.get('B', '8')); T = 1024
cfg = attention.GPTConfig()
model = attention.GPT(cfg).cuda().eval()
inputs, targets = gpt_ref.synthetic_tokens(B, T, cfg.vocab_size, 3, pad_tail=False)
inputs, targets = inputs.cuda(), targets.cuda()
with torch.inference_mode():
    for _ in range(2): model.forward_all(inputs, targets, reduction='none')
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 5
    for _ in range(n): out = model.forward_all(inputs, targets, reduction='none')
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f'GPT-2 small forward_all B={B} T={T} math={math}: {dt*1e3:.2f} ms  {B*T/dt:,.0f} tokens/s  mean nats/token {out.mean().item():.4f}')
