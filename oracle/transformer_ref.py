"""CPU restatement of haloop's encoder-decoder attention ASR path (ha/transformer.py, ha/conv.py) on
stock torch ops.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Functional form over parameter dicts keyed by the reference's state-dict names:
  AudioEncoder (transformer.py:202-258):  conv.conv.0.{weight[Cc,F,3],bias}, conv.conv.{1,2}.depthwise.{weight[C,1,3],bias},
      conv.conv.{1,2}.pointwise.{weight[Co,C,1],bias}, h.{i}.ln_time.weight, h.{i}.mix_time.{q,k,v,proj}.weight,
      h.{i}.ln_chan.weight, h.{i}.mix_chan.{0,2}.weight, ln_f.weight
  Decoder (transformer.py:60-199): wte.weight, h.{i}.{ln_time, mix_time.*, mix_memory.*, ln_chan, mix_chan.{0,2}}, ln_f.weight,
      lm_head.weight;  CTCAttentionDecoder (transformer.py:34-57) prefixes these with ``decoder.`` and adds
      ``recognizer.classifier.{weight,bias}``.
Pinned against the imported reference by tests/golden/g6_*.npz (tests/test_oracle_golden.py).

The greedy decoder (transformer.py:124-199) only runs under fp16 autocast in the reference (its KV caches are
float16 and an fp32 index_put into them raises).  ``decoder_decode`` restates it with fp32 arithmetic and
fp16-ROUNDED caches (the K/V bytes the reference keeps), which is also what the HIP path computes; against
the reference's own autocast run it agrees to fp16 tolerance, tokens exactly on peaked fixtures.
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

STX, ETX = 2, 3


def rotate_interleaved(x, t0=0, base=10000):
    """GPT-J style rotary embedding on the last two dims [T, C] (transformer.py:16-31)."""
    *_, T, C = x.shape
    t = torch.arange(t0, t0 + T, dtype=torch.float32)[:, None]
    exp = torch.arange(0, C // 2, dtype=torch.float32)[None, :]
    exp = -2 * exp.repeat_interleave(2, -1) / C
    ang = (base ** exp) * t
    sin, cos = torch.sin(ang), torch.cos(ang)
    odd, even = x[..., 0::2], x[..., 1::2]           # the reference's (swapped) names
    x_ = torch.stack([-even, odd], dim=-1).flatten(-2, -1)
    return x * cos + x_ * sin


def layer_norm(x, w):
    return F.layer_norm(x, w.shape, w, None, 1e-5)


def subsampled_lengths(lengths, strides, ks=3, pad=1):
    """conv.py:35-42: float floor per conv, int32 result."""
    o = lengths
    for s in strides:
        o = o + 2 * pad - ks
        o = torch.floor(o / s + 1)
    return o.int()


def conv_encoder(p, pre, x, strides):
    """x [N, F, T] -> [N, C_out, T'] (conv.py:25-47): gelu after every (separable) conv."""
    x = F.gelu(F.conv1d(x, p[pre + 'conv.0.weight'], p[pre + 'conv.0.bias'], stride=strides[0], padding=1))
    for i, s in enumerate(strides[1:], start=1):
        dw, db = p[pre + f'conv.{i}.depthwise.weight'], p[pre + f'conv.{i}.depthwise.bias']
        x = F.conv1d(x, dw, db, stride=s, padding=1, groups=dw.shape[0])
        x = F.conv1d(x, p[pre + f'conv.{i}.pointwise.weight'], p[pre + f'conv.{i}.pointwise.bias'])
        x = F.gelu(x)
    return x


def _heads(t, heads):
    N, T, C = t.shape
    return t.view(N, T, heads, C // heads).transpose(1, 2)


def attend(q, k, v, mask):
    """Softmax attention with its entropy monitor (transformer.py:413-430); mask True = masked."""
    qk = torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(k.shape[-1])
    if mask is not None:
        qk = qk.masked_fill(mask, float('-inf'))
    att = qk.softmax(dim=-1)
    ent = (-att * torch.log(att + 1e-8)).sum(dim=-1).mean(dim=(0, 1, 2))
    return att @ v, ent


def mha(p, pre, x, memory, heads, key_mask=None, causal=False, rope=False, t0=0, measure_entropy=False, att_mult=None,
        out_mult=None):
    """MultiHeadAttention.forward without caches (transformer.py:289-371).  key_mask [N,S] True = masked.
    With measure_entropy returns (y, entropy) through attend().  att_mult [N,H,T,S] / out_mult [N,T,C]: explicit
    inverted-dropout multipliers of the attention probabilities / of the proj output (training-mode parity)."""
    N, T, C = x.shape
    q = _heads(F.linear(x, p[pre + 'q.weight']), heads)
    k = _heads(F.linear(memory, p[pre + 'k.weight']), heads)
    v = _heads(F.linear(memory, p[pre + 'v.weight']), heads)
    if rope:
        q = rotate_interleaved(q, t0=t0)
        k = rotate_interleaved(k)
    ent = None
    if measure_entropy:
        mask = key_mask[:, None, None, :] if key_mask is not None else None
        if causal and mask is None:
            mask = ~torch.ones(k.size(-2), k.size(-2), dtype=torch.bool).tril()[-T:]
        y, ent = attend(q, k, v, mask)
    elif att_mult is not None:
        sc = torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(k.shape[-1])
        if key_mask is not None:
            sc = sc.masked_fill(key_mask[:, None, None, :], float('-inf'))
        elif causal:
            sc = sc.masked_fill(~torch.ones(T, k.size(-2), dtype=torch.bool).tril(), float('-inf'))
        y = (sc.softmax(dim=-1) * att_mult) @ v
    elif key_mask is not None:
        y = F.scaled_dot_product_attention(q, k, v, attn_mask=~key_mask[:, None, None, :])
    else:
        y = F.scaled_dot_product_attention(q, k, v, is_causal=causal)
    y = F.linear(y.transpose(1, 2).reshape(N, T, C), p[pre + 'proj.weight'])
    if out_mult is not None:
        y = y * out_mult
    return (y, ent) if measure_entropy else y


def block(p, pre, x, heads, causal=False, memory=None, memory_lengths=None, entropies=None, masks=None):
    """Block.forward (transformer.py:464-496): cross- and self-attention both read the SAME ln_time(x).
    ``entropies``: a list that receives (memory_entropy, self_entropy) -> the measure_entropy=True path.
    ``masks``: dropout multipliers {'cross_att','cross_out','self_att','self_out','mlp_out'} (training mode)."""
    me = entropies is not None
    mk = masks or {}
    x_norm = layer_norm(x, p[pre + 'ln_time.weight'])
    m_ent = None
    if memory is not None:
        mask = torch.arange(memory.shape[-2])[None, :] >= memory_lengths[:, None]
        m = mha(p, pre + 'mix_memory.', x_norm, memory, heads, key_mask=mask, measure_entropy=me, att_mult=mk.get('cross_att'),
                out_mult=mk.get('cross_out'))
        if me:
            m, m_ent = m
        x = x + m
    t = mha(p, pre + 'mix_time.', x_norm, x_norm, heads, causal=causal, rope=True, measure_entropy=me, att_mult=mk.get('self_att'),
            out_mult=mk.get('self_out'))
    if me:
        t, t_ent = t
        entropies.append((m_ent, t_ent))
    x = x + t
    h = F.gelu(F.linear(layer_norm(x, p[pre + 'ln_chan.weight']), p[pre + 'mix_chan.0.weight']))
    h = F.linear(h, p[pre + 'mix_chan.2.weight'])
    return x + (h * mk['mlp_out'] if 'mlp_out' in mk else h)


def n_layers(p, pre=''):
    i = 0
    while f'{pre}h.{i}.ln_time.weight' in p:
        i += 1
    return i


def audio_encoder_forward(p, x, input_lengths, heads, strides=(2, 2, 2), in_mult=None, block_masks=None):
    """AudioEncoder.forward (transformer.py:234-258): x [N,T,F] -> (features [N,T',C], lengths int32); eval mode unless the
    dropout multipliers in_mult [N,T',C] / block_masks (one dict per block) are given."""
    y = conv_encoder(p, 'conv.', x.mT, strides).mT
    if in_mult is not None:
        y = y * in_mult
    for i in range(n_layers(p)):
        y = block(p, f'h.{i}.', y, heads, masks=block_masks[i] if block_masks else None)
    return layer_norm(y, p['ln_f.weight']), subsampled_lengths(input_lengths, strides)


def decoder_logits(p, features, prompt, input_lengths, heads, pre='', entropies=None, block_masks=None):
    y = F.embedding(prompt, p[pre + 'wte.weight'])
    for i in range(n_layers(p, pre)):
        y = block(p, f'{pre}h.{i}.', y, heads, causal=True, memory=features, memory_lengths=input_lengths, entropies=entropies,
                  masks=block_masks[i] if block_masks else None)
    return F.linear(layer_norm(y, p[pre + 'ln_f.weight']), p[pre + 'lm_head.weight'])


def decoder_forward(p, features, targets, input_lengths, target_lengths, heads, reduction='mean', pre='', entropies=None,
                    block_masks=None):
    """Decoder.forward without label dropout (transformer.py:73-122)."""
    N, T = targets.shape
    prompt = F.pad(targets, (1, 0), value=STX)
    tg = F.pad(targets, (0, 1), value=0)
    tg[torch.arange(N), target_lengths] = ETX
    logits = decoder_logits(p, features, prompt, input_lengths, heads, pre, entropies, block_masks)
    if reduction == 'sumeach':
        return logits.log_softmax(dim=-1).max(dim=-1).values.sum(dim=-1)
    return F.cross_entropy(logits.view(-1, logits.size(-1)), tg.view(-1), ignore_index=0, reduction=reduction)


def ctc_attention_forward(p, features, condtargets, input_lengths, condtarget_lengths, heads):
    """CTCAttentionDecoder.forward (transformer.py:41-54): decoder CE + 0.3 * CTC on condtargets[:, 1:]."""
    dec = decoder_forward(p, features, condtargets, input_lengths, condtarget_lengths, heads, pre='decoder.')
    lp = F.linear(features, p['recognizer.classifier.weight'], p['recognizer.classifier.bias']).log_softmax(-1)
    ctc = F.ctc_loss(lp.permute(1, 0, 2), condtargets[:, 1:], input_lengths.long(), (condtarget_lengths - 1).long())
    return dec + 0.3 * ctc, dec, ctc


def decoder_decode(p, features, input_lengths, target_lengths, heads, prompt=None, pre='', cache_dtype=torch.float16):
    """Batched greedy decoding, Decoder.decode (transformer.py:124-199), all rows computed every step
    (the reference compacts to the alive rows, which changes nothing per row) with fp16-rounded K/V caches.
    Returns (list of token tensors, output_lengths, log_probs, sum_entropies, step_logprobs [N,T,V])."""
    N, S, C = features.shape
    T = int(target_lengths.max()) + 1
    L = n_layers(p, pre)
    hd = C // heads
    if prompt is None:
        tokens = torch.full((N, T + 1), ETX, dtype=torch.long)
        tokens[:, 0] = STX
        plen = 0
    else:
        P = prompt.shape[-1]
        tokens = torch.full((N, T + 1 + P), ETX, dtype=torch.long)
        tokens[:, 0] = STX
        tokens[:, 1:1 + P] = prompt
        plen = 1                                              # sic: only the first user token is forced (:186)
    rnd = lambda t: t.to(cache_dtype).to(torch.float32)
    mem_k = [rnd(_heads(F.linear(features, p[f'{pre}h.{i}.mix_memory.k.weight']), heads)) for i in range(L)]
    mem_v = [rnd(_heads(F.linear(features, p[f'{pre}h.{i}.mix_memory.v.weight']), heads)) for i in range(L)]
    time_k = [torch.zeros(N, heads, T, hd) for _ in range(L)]
    time_v = [torch.zeros(N, heads, T, hd) for _ in range(L)]
    mem_mask = torch.arange(S)[None, :] >= input_lengths[:, None]
    alive = torch.ones(N, dtype=torch.bool)
    out_len = torch.zeros(N, dtype=input_lengths.dtype)
    log_probs = torch.zeros(N)
    sum_ent = torch.zeros(N)
    steps = []
    for t in range(T):
        if not alive.any():
            break
        y = F.embedding(tokens[:, t:t + 1], p[pre + 'wte.weight'])          # [N,1,C]
        for i in range(L):
            bp = f'{pre}h.{i}.'
            xn = layer_norm(y, p[bp + 'ln_time.weight'])
            q = _heads(F.linear(xn, p[bp + 'mix_memory.q.weight']), heads)
            m = F.scaled_dot_product_attention(q, mem_k[i], mem_v[i], attn_mask=~mem_mask[:, None, None, :])
            y = y + F.linear(m.transpose(1, 2).reshape(N, 1, C), p[bp + 'mix_memory.proj.weight'])
            q = _heads(F.linear(xn, p[bp + 'mix_time.q.weight']), heads)
            time_k[i][:, :, t] = rnd(_heads(F.linear(xn, p[bp + 'mix_time.k.weight']), heads))[:, :, 0]
            time_v[i][:, :, t] = rnd(_heads(F.linear(xn, p[bp + 'mix_time.v.weight']), heads))[:, :, 0]
            q = rotate_interleaved(q, t0=t)
            k = rotate_interleaved(time_k[i][:, :, :t + 1])
            s = F.scaled_dot_product_attention(q, k, time_v[i][:, :, :t + 1])
            y = y + F.linear(s.transpose(1, 2).reshape(N, 1, C), p[bp + 'mix_time.proj.weight'])
            h = F.gelu(F.linear(layer_norm(y, p[bp + 'ln_chan.weight']), p[bp + 'mix_chan.0.weight']))
            y = y + F.linear(h, p[bp + 'mix_chan.2.weight'])
        lp = F.linear(layer_norm(y[:, -1, :], p[pre + 'ln_f.weight']), p[pre + 'lm_head.weight']).log_softmax(-1)
        steps.append(lp)
        val, idx = lp.max(dim=-1)
        sum_ent[alive] += (lp[alive].exp() * lp[alive] / math.log(2)).sum()   # sic: summed over ALL alive rows (:181)
        out_len[alive] += 1
        log_probs[alive] += val[alive]
        new = idx.clone()
        if t < plen:
            new = tokens[:, t + 1].clone()
        tokens[alive, t + 1] = new[alive]
        alive = alive & (new != ETX)
    outputs = [tokens[n, 1:int(out_len[n])] for n in range(N)]
    return outputs, out_len, log_probs, sum_ent, torch.stack(steps, 1)


# ---- deterministic parameters and inputs ----------------------------------------------------------
def _randn(g, shape, std):
    return torch.randn(shape, generator=g, dtype=torch.float32) * std


def _block_params(p, g, pre, C, memory):
    p[pre + 'ln_time.weight'] = 1.0 + _randn(g, (C,), 0.1)
    names = ['mix_time'] + (['mix_memory'] if memory else [])
    for m in names:
        for w in ('q', 'k', 'v'):
            p[f'{pre}{m}.{w}.weight'] = _randn(g, (C, C), 1.0 / math.sqrt(C))
        p[f'{pre}{m}.proj.weight'] = _randn(g, (C, C), 0.5 / math.sqrt(C))
    p[pre + 'ln_chan.weight'] = 1.0 + _randn(g, (C,), 0.1)
    p[pre + 'mix_chan.0.weight'] = _randn(g, (4 * C, C), 1.0 / math.sqrt(C))
    p[pre + 'mix_chan.2.weight'] = _randn(g, (C, 4 * C), 0.5 / math.sqrt(4 * C))


def make_encoder_params(head_dim, heads, layers, input_dim, conv_dim, n_convs, seed):
    """State dict of AudioEncoder(head_dim, heads, layers, input_dim, conv_dim, conv_strides of n_convs entries)."""
    g = torch.Generator().manual_seed(seed)
    C = head_dim * heads
    p = OrderedDict()
    p['conv.conv.0.weight'] = _randn(g, (conv_dim, input_dim, 3), 1.0 / math.sqrt(3 * input_dim))
    p['conv.conv.0.bias'] = _randn(g, (conv_dim,), 0.1)
    for i in range(1, n_convs):
        co = C if i == n_convs - 1 else conv_dim
        p[f'conv.conv.{i}.depthwise.weight'] = _randn(g, (conv_dim, 1, 3), 0.6)
        p[f'conv.conv.{i}.depthwise.bias'] = _randn(g, (conv_dim,), 0.1)
        p[f'conv.conv.{i}.pointwise.weight'] = _randn(g, (co, conv_dim, 1), 1.5 / math.sqrt(conv_dim))
        p[f'conv.conv.{i}.pointwise.bias'] = _randn(g, (co,), 0.1)
    for i in range(layers):
        _block_params(p, g, f'h.{i}.', C, memory=False)
    p['ln_f.weight'] = 1.0 + _randn(g, (C,), 0.1)
    return p


def make_decoder_params(vocab, head_dim, heads, layers, seed, with_ctc=True, sharp=4.0):
    """State dict of CTCAttentionDecoder (or of a bare Decoder when with_ctc is False).  ``sharp`` scales lm_head so the
    greedy choices are decisive (a trained model's are); near-ties would make fp16-vs-fp32 token parity meaningless."""
    g = torch.Generator().manual_seed(seed)
    C = head_dim * heads
    pre = 'decoder.' if with_ctc else ''
    p = OrderedDict()
    p[pre + 'wte.weight'] = _randn(g, (vocab, C), 1.0)
    for i in range(layers):
        _block_params(p, g, f'{pre}h.{i}.', C, memory=True)
    p[pre + 'ln_f.weight'] = 1.0 + _randn(g, (C,), 0.1)
    p[pre + 'lm_head.weight'] = _randn(g, (vocab, C), sharp / math.sqrt(C))
    if with_ctc:
        p['recognizer.classifier.weight'] = _randn(g, (vocab, C), 2.0 / math.sqrt(C))
        p['recognizer.classifier.bias'] = _randn(g, (vocab,), 0.1)
    return p


def synthetic_asr_batch(N, T, F_, vocab, S, seed, ragged=True):
    """Mel-like inputs [N,T,F], lengths, decoder targets in [4, vocab) (0 pad, 1 unk, 2 STX, 3 ETX are reserved)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, T, F_, generator=g)
    il = torch.tensor([T - (5 * i) % (T // 3) for i in range(N)], dtype=torch.int64) if ragged else torch.full((N,), T, dtype=torch.int64)
    tg = torch.randint(4, vocab, (N, S), generator=g)
    tl = torch.randint(max(1, S // 2), S + 1, (N,), generator=g)
    tl[0] = S
    for n in range(N):
        tg[n, int(tl[n]):] = 0
    return x, il, tg, tl
