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

# training direction: forward_all + loss.backward() (ha/attention_loop.py:196-208) and an AdamW step on the HIP kernels
from haloop_amd import ops
model.train()
params = [p for p in model.parameters()]
state = [(torch.zeros_like(p), torch.zeros_like(p)) for p in params]
def train_step(step):
    for p in params: p.grad = None
    loss = model.forward_all(inputs, targets)
    loss.backward()
    for p, (m, v) in zip(params, state):
        ops.adamw(p.data.view(-1), p.grad.view(-1), m.view(-1), v.view(-1), 3e-4, 0.9, 0.95, 1e-8, 0.1 if p.dim() >= 2 else 0.0, step)
    return loss
for i in range(2): train_step(i + 1)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 5
for i in range(n): loss = train_step(i + 3)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f'GPT-2 small train step (fwd+bwd+AdamW) B={B} T={T} math={math}: {dt*1e3:.2f} ms  {B*T/dt:,.0f} tokens/s  loss {loss.item():.4f}')
