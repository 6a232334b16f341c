"""CPU restatement of haloop's GPT scoring path (ha/attention.py:205-232 forward_all and the blocks
under it) on stock torch ops.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Functional form over a parameter dict keyed by the reference's state-dict names:
    transformer.wte.weight [V,C] (tied to lm_head.weight), transformer.wpe.weight [block,C],
    transformer.h.{i}.ln_1.weight (+.bias), .attn.c_attn.weight [3C,C] (+.bias), .attn.c_proj.weight,
    .ln_2.weight, .mlp.c_fc.weight [4C,C], .mlp.c_proj.weight [C,4C], transformer.ln_f.weight, lm_head.weight
Pinned against the imported reference by tests/golden/g5_gpt_*.npz.
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F


def new_gelu(x):
    """tanh form, ha/attention.py:12-17."""
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x * x * x)))


def make_gpt_params(vocab_size, block_size, n_layer, n_head, n_embd, bias, seed, stable=False):
    """Deterministic, non-degenerate parameters (the reference's own init zeroes wpe)."""
    g = torch.Generator().manual_seed(seed)
    C = n_embd

    def n(shape, std):
        return torch.randn(shape, generator=g, dtype=torch.float32) * std

    p = OrderedDict()
    p['transformer.wte.weight'] = n((vocab_size, C), 0.05)
    p['transformer.wpe.weight'] = n((block_size, C), 0.05)
    if stable:                                            # StableEmbedding's own LayerNorms (ha/attention.py:41)
        for e in ('wte', 'wpe'):
            p[f'transformer.{e}.norm.weight'] = 1.0 + n((C,), 0.1)
            p[f'transformer.{e}.norm.bias'] = n((C,), 0.05)
    for i in range(n_layer):
        pre = f'transformer.h.{i}.'
        p[pre + 'ln_1.weight'] = 1.0 + n((C,), 0.1)
        if bias: p[pre + 'ln_1.bias'] = n((C,), 0.05)
        p[pre + 'attn.c_attn.weight'] = n((3 * C, C), 1.0 / math.sqrt(C))
        if bias: p[pre + 'attn.c_attn.bias'] = n((3 * C,), 0.05)
        p[pre + 'attn.c_proj.weight'] = n((C, C), 0.5 / math.sqrt(C))
        if bias: p[pre + 'attn.c_proj.bias'] = n((C,), 0.05)
        p[pre + 'ln_2.weight'] = 1.0 + n((C,), 0.1)
        if bias: p[pre + 'ln_2.bias'] = n((C,), 0.05)
        p[pre + 'mlp.c_fc.weight'] = n((4 * C, C), 1.0 / math.sqrt(C))
        if bias: p[pre + 'mlp.c_fc.bias'] = n((4 * C,), 0.05)
        p[pre + 'mlp.c_proj.weight'] = n((C, 4 * C), 0.5 / math.sqrt(4 * C))
        if bias: p[pre + 'mlp.c_proj.bias'] = n((C,), 0.05)
    p['transformer.ln_f.weight'] = 1.0 + n((C,), 0.1)
    if bias: p['transformer.ln_f.bias'] = n((C,), 0.05)
    p['lm_head.weight'] = p['transformer.wte.weight']         # weight tying, attention.py:203
    return p


def gpt_forward_all(p, n_layer, n_head, input_ids, target_ids, reduction='mean', masks=None, causal=True):
    """masks (training-mode parity): {'emb': [B,T,C], 'att': [per layer [B,H,T,T]], 'res': [...[B,T,C]], 'mlp': [...]} inverted-dropout
    multipliers at the sites of ha/attention.py:224,90,127,141."""
    B, T = input_ids.shape
    C = p['transformer.wte.weight'].shape[1]
    tok, pos = F.embedding(input_ids, p['transformer.wte.weight']), p['transformer.wpe.weight'][:T][None]
    if 'transformer.wte.norm.weight' in p:                # stable_embedding: each embedding through its own LayerNorm
        tok = F.layer_norm(tok, (C,), p['transformer.wte.norm.weight'], p['transformer.wte.norm.bias'], 1e-5)
        pos = F.layer_norm(pos, (C,), p['transformer.wpe.norm.weight'], p['transformer.wpe.norm.bias'], 1e-5)
    x = tok + pos
    if masks:
        x = x * masks['emb']
    for i in range(n_layer):
        pre = f'transformer.h.{i}.'
        h = F.layer_norm(x, (C,), p[pre + 'ln_1.weight'], p.get(pre + 'ln_1.bias'), 1e-5)
        qkv = F.linear(h, p[pre + 'attn.c_attn.weight'], p.get(pre + 'attn.c_attn.bias'))
        q, k, v = (t.view(B, T, n_head, C // n_head).transpose(1, 2) for t in qkv.split(C, dim=2))
        if masks:
            sc = (q @ k.transpose(-2, -1)) / math.sqrt(k.shape[-1])
            if causal:
                sc = sc.masked_fill(~torch.ones(T, T, dtype=torch.bool).tril(), float('-inf'))
            y = (sc.softmax(-1) * masks['att'][i]) @ v
        else:
            y = F.scaled_dot_product_attention(q, k, v, is_causal=causal)
        y = y.transpose(1, 2).contiguous().view(B, T, C)
        r = F.linear(y, p[pre + 'attn.c_proj.weight'], p.get(pre + 'attn.c_proj.bias'))
        x = x + (r * masks['res'][i] if masks else r)
        h = F.layer_norm(x, (C,), p[pre + 'ln_2.weight'], p.get(pre + 'ln_2.bias'), 1e-5)
        h = new_gelu(F.linear(h, p[pre + 'mlp.c_fc.weight'], p.get(pre + 'mlp.c_fc.bias')))
        m = F.linear(h, p[pre + 'mlp.c_proj.weight'], p.get(pre + 'mlp.c_proj.bias'))
        x = x + (m * masks['mlp'][i] if masks else m)
    x = F.layer_norm(x, (C,), p['transformer.ln_f.weight'], p.get('transformer.ln_f.bias'), 1e-5)
    logits = F.linear(x, p['lm_head.weight'])
    return F.cross_entropy(logits.view(-1, logits.size(-1)), target_ids.reshape(-1), ignore_index=0, reduction=reduction)


def synthetic_tokens(B, T, vocab, seed, pad_tail=True):
    """hap-style batch (ha/score.py:57-70): targets = completions padded with 0, inputs = [eos] + completions[:-1]."""
    g = torch.Generator().manual_seed(seed)
    comp = torch.randint(1, vocab, (B, T), generator=g)
    if pad_tail:
        for b in range(B):
            n = int(torch.randint(T // 2, T + 1, (1,), generator=g))
            comp[b, n:] = 0
    eos = min(50256, vocab - 1)
    inputs = torch.cat([torch.full((B, 1), eos, dtype=torch.long), comp[:, :-1]], dim=1)
    return inputs, comp


def gpt_forward(p, n_layer, n_head, input_ids, past=None):
    """GPT.forward (ha/attention.py:253-279): logits of the last position and the present KV cache [L,2,B,nh,T,hs]."""
    B, T = input_ids.shape
    C = p['transformer.wte.weight'].shape[1]
    t0 = 0 if past is None else past.size(-2)
    x = F.embedding(input_ids, p['transformer.wte.weight']) + p['transformer.wpe.weight'][t0:t0 + T][None]
    present = []
    for i in range(n_layer):
        pre = f'transformer.h.{i}.'
        h = F.layer_norm(x, (C,), p[pre + 'ln_1.weight'], p.get(pre + 'ln_1.bias'), 1e-5)
        qkv = F.linear(h, p[pre + 'attn.c_attn.weight'], p.get(pre + 'attn.c_attn.bias'))
        q, k, v = (t.view(B, T, n_head, C // n_head).transpose(1, 2) for t in qkv.split(C, dim=2))
        if past is not None:
            k, v = torch.cat([past[i, 0], k], dim=-2), torch.cat([past[i, 1], v], dim=-2)
        att = (q @ k.transpose(-2, -1)) * (1.0 / (k.size(-1) ** 0.5))
        bias = k.new_ones(k.size(-2), k.size(-2)).tril()[-T:]
        att = att.masked_fill(bias[None, None] == 0, float('-inf')).softmax(dim=-1)
        y = (att @ v).transpose(1, 2).contiguous().view(B, T, C)
        present.append(torch.stack([k, v]))
        x = x + F.linear(y, p[pre + 'attn.c_proj.weight'], p.get(pre + 'attn.c_proj.bias'))
        h = F.layer_norm(x, (C,), p[pre + 'ln_2.weight'], p.get(pre + 'ln_2.bias'), 1e-5)
        h = new_gelu(F.linear(h, p[pre + 'mlp.c_fc.weight'], p.get(pre + 'mlp.c_fc.bias')))
        x = x + F.linear(h, p[pre + 'mlp.c_proj.weight'], p.get(pre + 'mlp.c_proj.bias'))
    x = F.layer_norm(x, (C,), p['transformer.ln_f.weight'], p.get('transformer.ln_f.bias'), 1e-5)
    return F.linear(x[:, [-1], :], p['lm_head.weight']), torch.stack(present)
