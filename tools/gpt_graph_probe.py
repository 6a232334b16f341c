#!/usr/bin/env python3
"""GPT-2 small bf16 train step: eager launches against train.GraphedTrainStep (forward + backward replayed from one HIP graph, AdamW eager)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from haloop_amd import _lib, attention, ops, synth
from haloop_amd.train import GraphedTrainStep
_lib.lib(); _lib.lend_scratch(256 << 20); _lib.set_math_mode(os.environ.get('HALO_MATH', 'bf16'))
B, T = 8, 1024
cfg = attention.GPTConfig()
torch.manual_seed(0)
model = attention.GPT(cfg).cuda().train()
with torch.no_grad():
    model.transformer.wpe.weight.normal_(0, 0.02)
inputs, targets = synth.synthetic_tokens(B, T, cfg.vocab_size, 3, pad_tail=False)
inputs, targets = inputs.cuda(), targets.cuda()
params = list(model.parameters())
opt = ops.AdamWMulti(params, [0.1 if p.dim() >= 2 else 0.0 for p in params], lr=3e-4, betas=(0.9, 0.95), eps=1e-8)


def eager():
    for p in params: p.grad = None
    loss = model.forward_all(inputs, targets)
    loss.backward()
    opt.step()
    return loss


def timed(fn, n=10, warm=3):
    for _ in range(warm): out = fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): out = fn()
    torch.cuda.synchronize()
    return out, (time.perf_counter() - t0) / n


if os.environ.get('MODE', 'both') in ('both', 'eager'):
    loss, t = timed(eager)
    print(f'eager  {t * 1e3:.2f} ms  loss {loss.item():.4f}', flush=True)
if os.environ.get('MODE', 'both') in ('both', 'graph'):
    g = GraphedTrainStep(lambda a, b: model.forward_all(a, b), params)

    def graphed():
        loss = g.step(inputs, targets)
        opt.step()
        return loss
    loss, t = timed(graphed)
    print(f'graph  {t * 1e3:.2f} ms  loss {loss.item():.4f}', flush=True)
