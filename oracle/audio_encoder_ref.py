"""CPU restatement of haloop's non-rotary GPT-block audio encoder (ha/attention_audio.py:64-117 AudioEncoder with
rotary_emb_dim = 0: the `audio-encoder` arch of ha/init.py:132-139) on stock torch ops.  TEST INFRASTRUCTURE ONLY.

    x [B, T, F] -> gelu(conv_pre) -> gelu(conv_subsample, stride 2) -> + frozen sinusoid positions (ha/attention_audio.py:10-16)
      -> dropout -> n_layer bidirectional GPT blocks (ha/attention.py:147-180, causal=False) -> ln_f
Functional form over a parameter dict keyed by the reference's state-dict names (conv_pre.*, conv_subsample.*, transformer.wpe.weight,
transformer.h.{i}.*, transformer.ln_f.*).  Pinned against the imported reference by tests/golden/g7_audio_encoder*.npz.
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

from .gpt_ref import new_gelu


def sinusoids(length, channels, max_timescale=10000):
    """ha/attention_audio.py:10-16"""
    assert channels % 2 == 0
    scales = torch.arange(channels // 2) / (channels // 2 - 1)
    inv_timescales = torch.exp(-math.log(max_timescale) * scales)
    scaled_time = torch.arange(length)[:, None] * inv_timescales[None, :]
    return torch.cat([torch.sin(scaled_time), torch.cos(scaled_time)], dim=1)


def make_params(d_input, n_embd, n_layer, block_size, bias, seed):
    g = torch.Generator().manual_seed(seed)
    C = n_embd

    def n(shape, std):
        return torch.randn(shape, generator=g, dtype=torch.float32) * std

    p = OrderedDict()
    p['conv_pre.weight'] = n((C, d_input, 3), 1.0 / math.sqrt(3 * d_input))
    p['conv_pre.bias'] = n((C,), 0.05)
    p['conv_subsample.weight'] = n((C, C, 3), 1.0 / math.sqrt(3 * C))
    p['conv_subsample.bias'] = n((C,), 0.05)
    p['transformer.wpe.weight'] = sinusoids(block_size, C)
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
    return p


def make_head_and_batch(n_embd, vocab, d_input, B, T, S, seed):
    """The CTC head (ha/recognizer.py TemporalClassifier parameters) and the seeded batch the fixtures were generated on."""
    g = torch.Generator().manual_seed(seed + 1)
    rec_p = {'classifier.weight': torch.randn(vocab, n_embd, generator=g) / n_embd ** 0.5, 'classifier.bias': torch.randn(vocab, generator=g) * 0.05}
    x = torch.randn(B, T, d_input, generator=g)
    il = torch.tensor([T - 5 * i for i in range(B)], dtype=torch.int64)
    tg = torch.randint(1, vocab, (B, S), generator=g)
    tl = torch.randint(max(1, S // 2), S + 1, (B,), generator=g)
    return rec_p, x, il, tg, tl


def subsampled_lengths(input_lengths):
    """ha/attention_audio.py:92-97: conv_subsample has k = 3, s = 2, p = 1."""
    return torch.floor((input_lengths + 2 * 1 - 3) / 2 + 1).int()


def forward(p, n_layer, n_head, x, input_lengths, masks=None):
    """masks (training-mode parity): {'emb': [B,T',C], 'att': [per layer [B,H,T',T']], 'res': [...], 'mlp': [...]} inverted-dropout
    multipliers at the sites of ha/attention_audio.py:110 and ha/attention.py:90,127,141."""
    y = F.gelu(F.conv1d(x.mT, p['conv_pre.weight'], p['conv_pre.bias'], stride=1, padding=1))
    y = F.gelu(F.conv1d(y, p['conv_subsample.weight'], p['conv_subsample.bias'], stride=2, padding=1)).mT
    B, T, C = y.shape
    y = y + p['transformer.wpe.weight'][:T][None]
    if masks:
        y = y * masks['emb']
    for i in range(n_layer):
        pre = f'transformer.h.{i}.'
        h = F.layer_norm(y, (C,), p[pre + 'ln_1.weight'], p.get(pre + 'ln_1.bias'), 1e-5)
        qkv = F.linear(h, p[pre + 'attn.c_attn.weight'], p.get(pre + 'attn.c_attn.bias'))
        q, k, v = (t.view(B, T, n_head, C // n_head).transpose(1, 2) for t in qkv.split(C, dim=2))
        if masks:
            sc = (q @ k.transpose(-2, -1)) / math.sqrt(k.shape[-1])
            a = (sc.softmax(-1) * masks['att'][i]) @ v
        else:
            a = F.scaled_dot_product_attention(q, k, v, is_causal=False)
        a = a.transpose(1, 2).contiguous().view(B, T, C)
        r = F.linear(a, p[pre + 'attn.c_proj.weight'], p.get(pre + 'attn.c_proj.bias'))
        y = y + (r * masks['res'][i] if masks else r)
        h = F.layer_norm(y, (C,), p[pre + 'ln_2.weight'], p.get(pre + 'ln_2.bias'), 1e-5)
        h = new_gelu(F.linear(h, p[pre + 'mlp.c_fc.weight'], p.get(pre + 'mlp.c_fc.bias')))
        m = F.linear(h, p[pre + 'mlp.c_proj.weight'], p.get(pre + 'mlp.c_proj.bias'))
        y = y + (m * masks['mlp'][i] if masks else m)
    y = F.layer_norm(y, (C,), p['transformer.ln_f.weight'], p.get('transformer.ln_f.bias'), 1e-5)
    return y, subsampled_lengths(input_lengths), {}
