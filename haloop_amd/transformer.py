"""Drop-in for the inference/scoring side of ha/transformer.py (encoder-decoder attention ASR, `hala`):
AudioEncoder, Block, MultiHeadAttention, Decoder (teacher-forced loss and batched greedy decode with fp16
KV caches), CTCAttentionDecoder, rotate_interleaved, attend, attend_chunked -- on the HIP
operators of csrc/attn.hip, csrc/conv.hip, csrc/gpt.hip and the GEMMs.

Same constructor arguments, attribute and state-dict names as the reference (``h.{i}.ln_time``,
``h.{i}.mix_time.{q,k,v,proj}``, ``h.{i}.mix_memory.*``, ``h.{i}.ln_chan``, ``h.{i}.mix_chan.{0,2}``, ``ln_f``,
``wte``, ``lm_head``, ``recognizer.classifier``), so reference checkpoints load unchanged.

Arithmetic: fp32 state; Linear layers on the split-bf16 (bf16x3) or exact-f32 MFMA GEMMs per
``halo_set_math_mode``; attention, LayerNorm, softmax, rotary and losses in fp32.  The greedy decoder keeps
the reference's float16 cache layout ``[L, 2, N, heads, S|T, head_dim]`` (ha/transformer.py:150-153) -- the
reference itself only decodes under fp16 autocast -- and computes everything around the caches in fp32.

Training: with grad enabled ``AudioEncoder.forward`` and ``Decoder.forward`` (hence CTCAttentionDecoder.forward:
decoder CE + 0.3 CTC) return tensors with a grad_fn whose backward is hand-written end to end (conv front-end,
blocks with shared-x_norm cross/self attention, inverse rotary, exact-GELU MLP, LayerNorm, embedding, CE), so
``loss.backward()`` fills ``.grad`` of every parameter like the reference's autograd does.

Training-mode dropout (p_drop > 0): every site of the reference (encoder input, attention probabilities, both
proj outputs, MLP output) draws from a Philox stream keyed by torch.initial_seed() -- statistically the reference's
dropout, not torch's bit stream; the output dropouts are GEMM epilogues, the probability dropout lives inside the
attention kernels (forward and both backward sweeps recompute the same mask).

Not built (raises): autograd through a bare Block / MultiHeadAttention call, ``kv_cache_parts`` on the public Block / MultiHeadAttention.forward (Decoder.decode drives the caches
itself), arbitrary attention masks (only the key-padding masks Block builds, transformer.py:476).
"""
import os
from collections import namedtuple

import torch
import torch.nn as nn

from . import _lib, ops
from ._linear import (NO_SITES, DropSites, WeightImages, training_images, drop_rows, forward_images, grad_images, linear, linear_dw, linear_dx, ln_linear,
                      normed_image, rowmajor_ok, use_split)
from .attention import LayerNorm
from .conv import ConvEncoder
from .recognizer import TemporalClassifier
from .rnn import DropoutStream

BlockKVCache = namedtuple('BlockKVCache', ['memory', 'time'])
Stats = namedtuple('Stats', ['meme_entropy', 'self_entropy'])

STX, ETX = 2, 3


def _wants_grad(module):
    return torch.is_grad_enabled() and any(p.requires_grad for p in module.parameters())


def _check_device_and_dropout(module, x):
    if not x.is_cuda:
        raise _lib.HaloError(f'haloop_amd.transformer.{type(module).__name__} runs on the HIP device only (no CPU path)')
    if module.training and not _wants_grad(module) and _p_drop(module) > 0:
        raise NotImplementedError('training-mode dropout is built into the autograd path only: enable grad, or call .eval()')


def _p_drop(module):
    """The model's dropout probability (the reference builds every site from one p_drop)."""
    ps = {float(m.p) for m in module.modules() if isinstance(m, nn.Dropout)}
    ps |= {float(m.p_drop) for m in module.modules() if hasattr(m, 'p_drop')}
    if len(ps) > 1:
        raise NotImplementedError(f'one dropout probability per model is built, found {sorted(ps)}')
    return ps.pop() if ps else 0.0


def _require_inference(module, x):
    """Module-level calls (Block, MultiHeadAttention, attend) have no autograd of their own: gradients flow through
    AudioEncoder.forward / Decoder.forward, whose backward is hand-written end to end."""
    if not x.is_cuda:
        raise _lib.HaloError(f'haloop_amd.transformer.{type(module).__name__} runs on the HIP device only (no CPU path)')
    if module.training and _p_drop(module) > 0:
        raise NotImplementedError('training-mode dropout is built into the AudioEncoder / Decoder autograd path only: call .eval()')
    if _wants_grad(module):
        raise NotImplementedError('haloop_amd.transformer.' + type(module).__name__ + ' has no autograd of its own: train through '
                                  'AudioEncoder / Decoder / CTCAttentionDecoder, or call it under torch.no_grad()')


class _GradSink:
    """Collects parameter gradients of one hand-written backward (summing when a parameter is hit twice)."""

    def __init__(self):
        self.g = {}

    def __call__(self, p, grad):
        if p is None or grad is None or not p.requires_grad:
            return
        k = id(p)
        self.g[k] = grad if k not in self.g else self.g[k] + grad


class _EncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, x, *params):
        with training_images():
            out, saved = model._forward_train(x)
        ctx.model, ctx.saved, ctx.params = model, saved, params
        return out

    @staticmethod
    def backward(ctx, dout):
        sink = _GradSink()
        ctx.model._backward_train(ctx.saved, dout.contiguous().float(), sink)
        ctx.saved = None
        return (None, None) + tuple(sink.g.get(id(p)) for p in ctx.params)


class _DecoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, features, prompt, tg, mlen, *params):
        with training_images():
            loss, saved = model._forward_train(features, prompt, tg, mlen)
        ctx.model, ctx.saved, ctx.params = model, saved, params
        return loss

    @staticmethod
    def backward(ctx, grad_rows):
        sink = _GradSink()
        dfeat = ctx.model._backward_train(ctx.saved, grad_rows.contiguous().float(), sink)
        ctx.saved = None
        return (None, dfeat if ctx.needs_input_grad[1] else None, None, None, None) + tuple(sink.g.get(id(p)) for p in ctx.params)


def _rope_table(cache, T, head_dim, device):
    hit = cache.get((head_dim, str(device)))
    if hit is None or hit.T < T:
        hit = ops.RopeTable(max(T, 256), head_dim, device)
        cache[(head_dim, str(device))] = hit
    return hit


_ROPE_TABLES = {}


def rotate_interleaved(x, *, t0=0, base=10000):
    "rotate query or key embedding as in https://arxiv.org/abs/2104.09864 GPT-J style (ha/transformer.py:16-31)"
    *lead, T, C = x.shape
    if not x.is_cuda:
        raise _lib.HaloError('haloop_amd.transformer.rotate_interleaved runs on the HIP device only')
    table = ops.RopeTable(t0 + T, C, x.device, base) if base != 10000 else _rope_table(_ROPE_TABLES, t0 + T, C, x.device)
    y = x.float().contiguous().clone().view(-1, C)
    ops.rope_(y, T, 1, C, table, t0=t0)
    return y.view(*lead, T, C)


def attend(q, k, v, mask):
    """(N, heads, T, hd) x (N, heads, S, hd) -> (N, heads, T, hd), entropy (ha/transformer.py:413-430).
    ``mask`` None, a key-padding mask (N, 1, 1, S) whose True entries form a suffix (the models' case: the tiled kernel), or any
    boolean mask broadcastable to (N, heads, T, S) -- and any head dimension -- through the general one-wave-per-row kernel."""
    N, H, T, hd = q.shape
    S = k.shape[-2]
    lens = None
    if hd not in (16, 32, 64) or (mask is not None and not _is_suffix_padding_mask(mask, N, S)):
        y, ent = ops.attention_masked(q, k, v, mask)
        return y.to(q.dtype), ent.mean()
    if mask is not None:
        lens = _suffix_mask_lengths(mask, N, S)
    q2, k2, v2 = (t.transpose(1, 2).reshape(N * t.shape[2], H * hd).float().contiguous() for t in (q, k, v))
    y, _, ent = ops.attention_fwd(q2, k2, v2, N, H, hd, T, S, key_lengths=lens, want_entropy=True)
    return y.view(N, T, H, hd).transpose(1, 2), ent.mean()


def attend_chunked(q, k, v, mask, chunk_size=32):
    "same result as attend without the monitor (ha/transformer.py:374-410); the kernel is already tiled"
    x, _ = attend(q, k, v, mask)
    return x, torch.tensor(float('-inf'))


def _is_suffix_padding_mask(mask, N, S):
    if mask.numel() != N * S or mask.shape[0] != N or mask.shape[-1] != S:
        return False
    m = mask.reshape(N, S)
    lens = (~m).sum(-1)
    return bool((m == (torch.arange(S, device=m.device)[None, :] >= lens[:, None])).all())


def _suffix_mask_lengths(mask, N, S):
    m = mask.reshape(N, -1, S)
    if m.shape[1] != 1:
        raise NotImplementedError('only key-padding masks of shape (N, 1, 1, S) are built')
    m = m[:, 0]
    lens = (~m).sum(-1).to(torch.int32)
    if not bool((m == (torch.arange(S, device=m.device)[None, :] >= lens[:, None])).all()):
        raise NotImplementedError('only key-padding masks whose masked keys form a suffix are built')
    return lens


class MultiHeadAttention(nn.Module):
    def __init__(self, head_dim: int = 64, heads: int = 12, p_drop: float = 0.1):
        super().__init__()
        self.head_dim = head_dim
        self.heads = heads
        self.q = nn.Linear(head_dim * heads, head_dim * heads, bias=False)
        self.k = nn.Linear(head_dim * heads, head_dim * heads, bias=False)
        self.v = nn.Linear(head_dim * heads, head_dim * heads, bias=False)
        self.proj = nn.Linear(head_dim * heads, head_dim * heads, bias=False)
        self.p_drop = p_drop
        self.dropout = nn.Dropout(p_drop)
        self._images = WeightImages()
        self._tables = {}

    def init_from_flash_mha_(self, mha):
        step = self.head_dim * self.heads
        assert mha.Wqkv.weight.shape[0] == step * 3
        self.q.weight.data = mha.Wqkv.weight.data[0*step:1*step, :]
        self.k.weight.data = mha.Wqkv.weight.data[1*step:2*step, :]
        self.v.weight.data = mha.Wqkv.weight.data[2*step:3*step, :]
        self.proj.weight.data = mha.out_proj.weight.data
        return self

    def read_memory(self, memory):
        N, S, C = memory.shape
        kv = linear(self._images, memory.reshape(N * S, C).float().contiguous(), (self.k.weight, self.v.weight))
        k, v = kv[:, :C], kv[:, C:]
        return (k.reshape(N, S, self.heads, self.head_dim).transpose(-3, -2),
                v.reshape(N, S, self.heads, self.head_dim).transpose(-3, -2))

    # x2d [N*T, C] (already normalised), mem2d [N*S, C] or None for self-attention -> attention output [N*T, C] (before proj)
    def _attend2d(self, x2d, mem2d, N, T, S, key_lengths=None, causal=False, rope=False, t0=0, measure_entropy=False, x_img=None):
        C = self.heads * self.head_dim
        if mem2d is None:
            qkv = linear(self._images, x2d, (self.q.weight, self.k.weight, self.v.weight), a_image=x_img)   # one GEMM, [N*T, 3C]
            q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
        else:
            q = linear(self._images, x2d, self.q.weight, a_image=x_img)
            kv = linear(self._images, mem2d, (self.k.weight, self.v.weight))
            k, v = kv[:, :C], kv[:, C:]
        if rope:
            table = _rope_table(self._tables, max(t0 + T, S), self.head_dim, x2d.device)
            ops.rope_(q, T, self.heads, self.head_dim, table, t0=t0)
            ops.rope_(k, S, self.heads, self.head_dim, table)
        y, _, ent = ops.attention_fwd(q, k, v, N, self.heads, self.head_dim, T, S, causal=causal, key_lengths=key_lengths,
                                      want_entropy=measure_entropy)
        return y, (ent.mean() if measure_entropy else torch.tensor(float('-inf')))

    # training twins of _attend2d: keep q/k/v (after the rotary), the output and the log-sum-exp
    def _attend2d_train(self, x2d, mem2d, N, T, S, key_lengths=None, causal=False, rope=False, site=(ops.NO_DROPOUT, 0), x_img=None):
        C = self.heads * self.head_dim
        if mem2d is None:
            qkv = linear(self._images, x2d, (self.q.weight, self.k.weight, self.v.weight), a_image=x_img)
            q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
        else:
            q = linear(self._images, x2d, self.q.weight, a_image=x_img)
            kv = linear(self._images, mem2d, (self.k.weight, self.v.weight))
            k, v = kv[:, :C], kv[:, C:]
        table = None
        if rope:
            table = _rope_table(self._tables, max(T, S), self.head_dim, x2d.device)
            ops.rope_(q, T, self.heads, self.head_dim, table)
            ops.rope_(k, S, self.heads, self.head_dim, table)
        y, lse, _ = ops.attention_fwd(q, k, v, N, self.heads, self.head_dim, T, S, causal=causal, key_lengths=key_lengths, want_lse=True,
                                      drop=site[0], stream_id=site[1])
        return y, (x2d, mem2d, q, k, v, y, lse, key_lengths, causal, table, N, T, S, site)

    def _attend2d_bwd(self, saved, dy, put, dx_out=None, dmem_out=None):
        """dy: gradient of the attention output (before proj).  Accumulates the input gradient into dx_out (allocates it
        when None) and, for cross-attention, the memory gradient into dmem_out.  -> dx"""
        x2d, mem2d, q, k, v, y, lse, key_lengths, causal, table, N, T, S, site = saved
        C = self.heads * self.head_dim
        dev = x2d.device
        if mem2d is None:
            dqkv = torch.empty(N * T, 3 * C, device=dev, dtype=torch.float32)
            dq, dk, dv = dqkv[:, :C], dqkv[:, C:2 * C], dqkv[:, 2 * C:]
        else:
            dq = torch.empty(N * T, C, device=dev, dtype=torch.float32)
            dkv = torch.empty(N * S, 2 * C, device=dev, dtype=torch.float32)
            dk, dv = dkv[:, :C], dkv[:, C:]
        ops.attention_bwd(q, k, v, y, dy, lse, dq, dk, dv, N, self.heads, self.head_dim, T, S, causal=causal, key_lengths=key_lengths,
                          drop=site[0], stream_id=site[1])
        if table is not None:                                   # the rotation is orthogonal: its transpose rotates back
            ops.rope_(dq, T, self.heads, self.head_dim, table, inverse=True)
            ops.rope_(dk, S, self.heads, self.head_dim, table, inverse=True)
        ws = (self.q.weight, self.k.weight, self.v.weight)
        # each projection gradient feeds a weight-gradient and an input-gradient GEMM: both operand images from one read
        if mem2d is None:
            im, im_t = grad_images(dqkv, C)
            dw = linear_dw(dqkv, x2d, dy_image_t=im_t)          # [3C, C]
            for i, w in enumerate(ws):
                put(w, dw[i * C:(i + 1) * C])
            return linear_dx(self._images, dqkv, ws, out=dx_out, accumulate=dx_out is not None, dy_image=im)
        im, im_t = grad_images(dq, C)
        put(self.q.weight, linear_dw(dq, x2d, dy_image_t=im_t))
        km, km_t = grad_images(dkv, C) if dmem_out is not None else (None, None)
        dwkv = linear_dw(dkv, mem2d, dy_image_t=km_t)
        put(self.k.weight, dwkv[:C]); put(self.v.weight, dwkv[C:])
        if dmem_out is not None:
            linear_dx(self._images, dkv, ws[1:], out=dmem_out, accumulate=True, dy_image=km)
        return linear_dx(self._images, dq, self.q.weight, out=dx_out, accumulate=dx_out is not None, dy_image=im)

    def forward(self, x, memory, *, mask=None, causal=False, measure_entropy=False, kv_cache_parts=None, t0=0, rope=False,
                _key_lengths=None):
        _require_inference(self, x)
        if kv_cache_parts is not None:
            raise NotImplementedError('kv_cache_parts: the fp16 caches are driven by haloop_amd.transformer.Decoder.decode')
        N, T, C = x.shape
        S = memory.shape[1]
        if mask is not None and _key_lengths is None:
            _key_lengths = _suffix_mask_lengths(mask, N, S)
        x2d = x.reshape(N * T, C).float().contiguous()
        mem2d = None if memory is x else memory.reshape(N * S, C).float().contiguous()
        y, ent = self._attend2d(x2d, mem2d, N, T, S, _key_lengths, causal and _key_lengths is None, rope, t0, measure_entropy)
        return linear(self._images, y, self.proj.weight).view(N, T, C), ent


class Block(nn.Module):
    def __init__(self, head_dim: int, heads: int, p_drop: float, memory=False):
        super().__init__()
        self.heads = heads
        self.head_dim = head_dim
        self.ln_time = LayerNorm(head_dim * heads, bias=False)
        self.mix_time = MultiHeadAttention(head_dim=head_dim, heads=heads, p_drop=p_drop)
        self.mix_memory = MultiHeadAttention(head_dim=head_dim, heads=heads, p_drop=p_drop) if memory else None
        self.ln_chan = LayerNorm(head_dim * heads, bias=False)
        self.mix_chan = nn.Sequential(
            nn.Linear(head_dim * heads, head_dim * heads * 4, bias=False),
            nn.GELU(),
            nn.Linear(head_dim * heads * 4, head_dim * heads, bias=False),
            nn.Dropout(p_drop),
        )
        self._images = WeightImages()

    # x2d [N*T, C] is updated IN PLACE (the caller owns it); memory rows [N*S, C]
    def _forward2d(self, x2d, N, T, causal=False, mem2d=None, S=0, memory_lengths=None, measure_entropy=False, time_lengths=None):
        # ln_time(x) feeds up to two GEMMs (cross query, self q|k|v): one LayerNorm pass writes their shared operand image
        x_norm, x_img = normed_image(x2d, self.ln_time.weight, n_out=x2d.shape[1])
        m_ent = torch.tensor(float('-inf'))
        if self.mix_memory is not None:
            mm = self.mix_memory
            m, m_ent = mm._attend2d(x_norm, mem2d, N, T, S, key_lengths=memory_lengths, measure_entropy=measure_entropy, x_img=x_img)
            linear(mm._images, m, mm.proj.weight, out=x2d, accumulate=True)                      # x += cross(ln(x), memory)
        mt = self.mix_time
        t, t_ent = mt._attend2d(x_norm, None, N, T, T, key_lengths=time_lengths, causal=causal and time_lengths is None, rope=True,
                                measure_entropy=measure_entropy, x_img=x_img)                    # the SAME x_norm (:476-494)
        linear(mt._images, t, mt.proj.weight, out=x2d, accumulate=True)
        M, C = x2d.shape
        if rowmajor_ok(M, 4 * C, C) and C % 32 == 0:        # the MLP's activations go on as row-major bf16 (halo_gemm_split_io)
            h, _ = ln_linear(self._images, x2d, self.ln_chan.weight, None, self.mix_chan[0].weight, gelu='erf', out_rowmajor=True)
            linear(self._images, None, self.mix_chan[2].weight, out=x2d, accumulate=True, a_rowmajor=h, shape=(M, 4 * C))
            return m_ent, t_ent
        h, _ = ln_linear(self._images, x2d, self.ln_chan.weight, None, self.mix_chan[0].weight, gelu='erf')
        linear(self._images, h, self.mix_chan[2].weight, out=x2d, accumulate=True)
        return m_ent, t_ent

    def _forward2d_train(self, x0, N, T, causal=False, mem2d=None, S=0, memory_lengths=None, sites=NO_SITES):
        """Dropout sites in forward order: [cross-attention probabilities, cross proj output,] self-attention probabilities,
        self proj output, MLP output (ha/transformer.py:356,371,457); each output dropout is the GEMM's epilogue."""
        x_norm, x_img = normed_image(x0, self.ln_time.weight, n_out=x0.shape[1])
        xa, sv_m, s_mo = x0, None, None
        if self.mix_memory is not None:
            mm = self.mix_memory
            ym, sv_m = mm._attend2d_train(x_norm, mem2d, N, T, S, key_lengths=memory_lengths, site=sites.next(), x_img=x_img)
            s_mo = sites.next()
            xa = linear(mm._images, ym, mm.proj.weight, out=x0.clone(), accumulate=True, drop=s_mo[0], stream_id=s_mo[1])
        mt = self.mix_time
        yt, sv_t = mt._attend2d_train(x_norm, None, N, T, T, causal=causal, rope=True, site=sites.next(), x_img=x_img)
        s_to = sites.next()
        xb = linear(mt._images, yt, mt.proj.weight, out=xa.clone(), accumulate=True, drop=s_to[0], stream_id=s_to[1])
        a, hn = ln_linear(self._images, xb, self.ln_chan.weight, None, self.mix_chan[0].weight, want_normed=True)
        # gelu(a) is only ever a GEMM operand (mix_chan[2] now, its weight gradient later): write its two images, not the matrix
        g_img, g_img_t = forward_images(a, x0.shape[1], ops.PAIR_GELU_ERF)
        g = ops.gelu_fwd(a, exact=True) if g_img is None else None
        s_co = sites.next()
        xc = linear(self._images, g, self.mix_chan[2].weight, out=xb.clone(), accumulate=True, drop=s_co[0], stream_id=s_co[1],
                    a_image=g_img, shape=a.shape)
        return xc, (x0, sv_m, sv_t, xb, hn, a, g, g_img_t, s_mo, s_to, s_co)

    def _backward2d(self, saved, dxc, put, dmem_out=None):
        x0, sv_m, sv_t, xb, hn, a, g, g_img_t, s_mo, s_to, s_co = saved
        w0, w2 = self.mix_chan[0].weight, self.mix_chan[2].weight
        C, M = x0.shape[1], x0.shape[0]
        dmlp = drop_rows(dxc, s_co)                                             # gradient at the MLP output, before its dropout
        dm_img, dm_img_t = grad_images(dmlp, a.shape[1])
        put(w2, linear_dw(dmlp, g, dy_image_t=dm_img_t, x_image_t=g_img_t, shapes=(dmlp.shape, a.shape)))
        dg = linear_dx(self._images, dmlp, w2, dy_image=dm_img)
        if use_split(M, C, a.shape[1]) and use_split(a.shape[1], C, M):
            da_img, da_img_t = ops.image_pair(dg, ops.PAIR_GELU_ERF_BWD, a)    # da = dg * gelu'(a), as its two operand images only
            put(w0, linear_dw(None, hn, dy_image_t=da_img_t, shapes=(a.shape, hn.shape)))
            d_ln = linear_dx(self._images, None, w0, dy_image=da_img, shape=a.shape)
        else:
            da = ops.gelu_bwd(dg, a, exact=True)
            put(w0, linear_dw(da, hn))
            d_ln = linear_dx(self._images, da, w0)
        dxb, dw, _ = ops.layernorm_bwd(d_ln, xb, self.ln_chan.weight, dxc)
        put(self.ln_chan.weight, dw)
        mt = self.mix_time
        dto = drop_rows(dxb, s_to)
        im, im_t = grad_images(dto, C)
        put(mt.proj.weight, linear_dw(dto, sv_t[5], dy_image_t=im_t))
        dxn = mt._attend2d_bwd(sv_t, linear_dx(mt._images, dto, mt.proj.weight, dy_image=im), put)
        if sv_m is not None:
            mm = self.mix_memory
            dmo = drop_rows(dxb, s_mo)
            im, im_t = grad_images(dmo, C)
            put(mm.proj.weight, linear_dw(dmo, sv_m[5], dy_image_t=im_t))
            mm._attend2d_bwd(sv_m, linear_dx(mm._images, dmo, mm.proj.weight, dy_image=im), put, dx_out=dxn, dmem_out=dmem_out)
        dx0, dw, _ = ops.layernorm_bwd(dxn, x0, self.ln_time.weight, dxb)      # both attentions read the same ln_time(x)
        put(self.ln_time.weight, dw)
        return dx0

    def forward(self, x, time_mask=None, causal=False, memory=None, memory_lengths=None, measure_entropy=False,
                kv_cache_parts=BlockKVCache(memory=None, time=None), t0=0):
        _require_inference(self, x)
        if kv_cache_parts.memory is not None or kv_cache_parts.time is not None:
            raise NotImplementedError('kv_cache_parts: the fp16 caches are driven by haloop_amd.transformer.Decoder.decode')
        if t0 != 0:
            raise NotImplementedError('t0 != 0 only occurs with kv caches')
        N, T, C = x.shape
        x2d = x.reshape(N * T, C).float().clone()
        mem2d, S, mlen = None, 0, None
        if self.mix_memory is not None:
            S = memory.shape[1]
            mem2d = memory.reshape(N * S, C).float().contiguous()
            mlen = memory_lengths.to(device=x.device, dtype=torch.int32)
        tlen = _suffix_mask_lengths(time_mask, N, T) if time_mask is not None else None
        ents = self._forward2d(x2d, N, T, causal, mem2d, S, mlen, measure_entropy, tlen)
        return x2d.view(N, T, C), ents


class Decoder(nn.Module):
    def __init__(self, *, vocab: int, head_dim: int, heads: int, p_drop: float, layers: int):
        super().__init__()
        self.wte = nn.Embedding(vocab, head_dim * heads)
        self.h = nn.ModuleList([Block(head_dim=head_dim, heads=heads, p_drop=p_drop, memory=True) for _ in range(layers)])
        self.ln_f = LayerNorm(head_dim * heads, bias=False)
        self.lm_head = nn.Linear(head_dim * heads, vocab, bias=False)
        self._images = WeightImages()
        self._graphs = {}                                                 # captured greedy decodes, see _decode_graph
        self.dropout_stream = DropoutStream()                             # Philox (seed, offset) per training forward

    def forward(self, features, targets, input_lengths=None, target_lengths=None, star_penalty=None, measure_entropy=False,
                drop_labels=None, reduction='mean'):
        _check_device_and_dropout(self, features)
        dev = features.device
        targets = targets.to(dev)
        N, T = targets.shape
        # prompt: STX a b c / target: a b c ETX PAD (ha/transformer.py:84-98)
        prompt = nn.functional.pad(targets, (1, 0), value=STX)
        tg = nn.functional.pad(targets, (0, 1), value=0)
        tg.scatter_(1, target_lengths.to(device=dev, dtype=torch.long).view(N, 1), ETX)       # tg[n, target_lengths[n]] = ETX, capture-safe
        T = T + 1
        if (drop_labels is None and self.training) or drop_labels:
            # label dropout (ha/transformer.py:100-103): integer data preparation, drawn from torch's generator like the reference
            keep = torch.empty_like(prompt).bernoulli_(0.9).bool()
            prompt = torch.where(keep, prompt, torch.ones_like(prompt))
        S, C = features.shape[1], features.shape[2]
        mlen = input_lengths.to(device=dev, dtype=torch.int32)
        if _wants_grad(self) or (torch.is_grad_enabled() and features.requires_grad):
            if measure_entropy or reduction == 'sumeach':
                raise NotImplementedError("measure_entropy / reduction='sumeach' are inference-only here: call under torch.no_grad()")
            params = [p for p in self.parameters() if p.requires_grad]
            per_tok = _DecoderFn.apply(self, features, prompt, tg, mlen, *params)
            if reduction == 'none':
                loss = per_tok
            elif reduction == 'sum':
                loss = per_tok.sum()
            elif reduction == 'mean':
                loss = per_tok.sum() / (tg != 0).sum()
            else:
                raise ValueError(f'{reduction} is not a valid value for reduction')
            ninf = [torch.tensor(float('-inf'))] * len(self.h)
            return loss, Stats(meme_entropy=list(ninf), self_entropy=list(ninf))._asdict()
        mem2d = features.reshape(N * S, C).float().contiguous()
        stats = Stats(meme_entropy=[], self_entropy=[])
        y = ops.embed_fwd(prompt, self.wte.weight, None)
        for block in self.h:
            m_ent, t_ent = block._forward2d(y, N, T, True, mem2d, S, mlen, measure_entropy)
            stats.meme_entropy.append(m_ent)
            stats.self_entropy.append(t_ent)
        logits = linear(self._images, ops.layernorm_fwd(y, self.ln_f.weight), self.lm_head.weight)   # [N*T, V]
        if reduction == 'sumeach':
            loss = ops.logprob_max(logits)[0].view(N, T).sum(dim=-1)
        else:
            per_tok = ops.cross_entropy_fwd(logits, tg.reshape(-1), ignore_index=0)
            if reduction == 'none':
                loss = per_tok
            elif reduction == 'sum':
                loss = per_tok.sum()
            elif reduction == 'mean':
                loss = per_tok.sum() / (tg != 0).sum()
            else:
                raise ValueError(f'{reduction} is not a valid value for reduction')
        return loss, stats._asdict()

    @torch.no_grad()
    def _forward_train(self, features, prompt, tg, mlen):
        N, T = prompt.shape
        S, C = features.shape[1], features.shape[2]
        mem2d = features.detach().reshape(N * S, C).float().contiguous()
        y = ops.embed_fwd(prompt, self.wte.weight, None)
        sites = DropSites(self.dropout_stream.next(_p_drop(self), self.training))
        blocks = []
        for block in self.h:
            y, sv = block._forward2d_train(y, N, T, True, mem2d, S, mlen, sites)
            blocks.append(sv)
        xf = ops.layernorm_fwd(y, self.ln_f.weight)
        logits = linear(self._images, xf, self.lm_head.weight)
        loss, row_lse = ops.cross_entropy_fwd_lse(logits, tg.reshape(-1), ignore_index=0)
        return loss, (prompt, tg.reshape(-1), blocks, y, xf, logits, row_lse, (N, S, C))

    @torch.no_grad()
    def _backward_train(self, saved, grad_rows, put):
        prompt, tg, blocks, y_last, xf, logits, row_lse, (N, S, C) = saved
        dlogits = ops.cross_entropy_bwd_(logits, tg, row_lse, grad_rows, ignore_index=0)
        put(self.lm_head.weight, linear_dw(dlogits, xf))
        dy, dw, _ = ops.layernorm_bwd(linear_dx(self._images, dlogits, self.lm_head.weight), y_last, self.ln_f.weight)
        put(self.ln_f.weight, dw)
        dmem = torch.zeros(N * S, C, device=xf.device, dtype=torch.float32)
        for block, sv in zip(reversed(self.h), reversed(blocks)):
            dy = block._backward2d(sv, dy, put, dmem_out=dmem)
        dwte = torch.zeros_like(self.wte.weight)
        ops.embed_bwd(prompt, dy, dwte, None)
        put(self.wte.weight, dwte)
        return dmem.view(N, S, C)

    @torch.no_grad()
    def _decode_core(self, mem2d, mlen, tokens_init, plen, N, S, T):
        """All launches of one batched greedy decode, host-sync free (capturable in a HIP graph).
        mem2d [N*S, C] fp32 features, mlen [N] int32, tokens_init [N, W] int64 (STX, optional prompt, ETX fill)."""
        dev = mem2d.device
        C = mem2d.shape[1]
        L = len(self.h)
        heads, head_dim = self.h[0].heads, self.h[0].head_dim
        tokens = tokens_init.clone()
        mem_cache = torch.zeros((L, 2, N, heads, S, head_dim), dtype=torch.float16, device=dev)
        time_cache = torch.zeros((L, 2, N, heads, T, head_dim), dtype=torch.float16, device=dev)
        for l, block in enumerate(self.h):                                   # cross-attention caches, warmed once (:324-334)
            mm = block.mix_memory
            kv = linear(mm._images, mem2d, (mm.k.weight, mm.v.weight))
            ops.kv_cache_store(kv, C, mem_cache[l, 0], mem_cache[l, 1], N, S, heads, head_dim, 0)
        table = ops.RopeTable(T, head_dim, dev)
        alive = torch.ones(N, dtype=torch.uint8, device=dev)
        out_len = torch.zeros(N, dtype=torch.int32, device=dev)
        log_probs = torch.zeros(N, dtype=torch.float32, device=dev)
        sum_entropies = torch.zeros(N, dtype=torch.float32, device=dev)
        # every step computes all N rows; rows that are no longer alive are ignored by the update (the reference
        # compacts to the alive rows instead, which changes no alive row's result) and nothing syncs with the host
        for t in range(T):
            y = ops.embed_fwd(tokens[:, t:t + 1], self.wte.weight, None)         # [N, C]
            for l, block in enumerate(self.h):
                mm, mt = block.mix_memory, block.mix_time
                xn = ops.layernorm_fwd(y, block.ln_time.weight)
                # both attentions read the same ln_time(x) (:476-494): one GEMM gives the cross query and the self q | k | v
                a = linear(mt._images, xn, (mm.q.weight, mt.q.weight, mt.k.weight, mt.v.weight))      # [N, 4C]
                m = ops.attention_decode(a, mem_cache[l, 0], mem_cache[l, 1], S, key_lengths=mlen)
                linear(mm._images, m, mm.proj.weight, out=y, accumulate=True)
                # cache store (fp16) + rotary + attention over the t + 1 cached positions in one launch
                s = ops.attention_decode_step(a[:, C:2 * C], a[:, 2 * C:3 * C], a[:, 3 * C:], time_cache[l, 0], time_cache[l, 1], t + 1,
                                              table=table)
                linear(mt._images, s, mt.proj.weight, out=y, accumulate=True)
                h = linear(block._images, ops.layernorm_fwd(y, block.ln_chan.weight), block.mix_chan[0].weight, gelu='erf')
                linear(block._images, h, block.mix_chan[2].weight, out=y, accumulate=True)
            logits = linear(self._images, ops.layernorm_fwd(y, self.ln_f.weight), self.lm_head.weight)
            val, idx, negent = ops.logprob_max(logits, want_entropy=True)
            ops.greedy_update(val, idx, negent, tokens, t, plen, ETX, alive, out_len, log_probs, sum_entropies)
        return tokens, out_len, log_probs, sum_entropies

    def _fused_decode_ok(self, C):
        """The fused decode launches (csrc/decode.hip): split-bf16 arithmetic, so not in exact-f32 mode; HALO_DECODE_FUSED=0 keeps
        the operator-per-launch path (same results up to the products' rounding)."""
        return (os.environ.get('HALO_DECODE_FUSED', '1') != '0' and _lib.get_math_mode() != 'f32'
                and ops.decode_linear_supported(C, True) and ops.decode_linear_supported(2 * C, False)
                and ops.decode_linear_supported(4 * C, False) and self.h[0].head_dim in (16, 32, 64, 128))

    @torch.no_grad()
    def _decode_images(self):
        """Decode images of every product of a step, rebuilt when a parameter changes: per layer the merged projection
        [mm.q; mt.q; mt.k; mt.v], the merged output projection [mm.proj | mt.proj] (K = 2C: cross | self attention outputs), the two
        MLP weights; and lm_head."""
        stamp = tuple((p._version, p.data_ptr()) for p in self.parameters())
        if getattr(self, '_dec_images', None) is None or self._dec_images[0] != stamp:
            layers = []
            for block in self.h:
                mm, mt = block.mix_memory, block.mix_time
                layers.append((ops.decode_image(torch.cat([mm.q.weight, mt.q.weight, mt.k.weight, mt.v.weight], 0).float().contiguous()),
                               ops.decode_image(torch.cat([mm.proj.weight, mt.proj.weight], 1).float().contiguous()),
                               ops.decode_image(block.mix_chan[0].weight.detach().float().contiguous()),
                               ops.decode_image(block.mix_chan[2].weight.detach().float().contiguous())))
            head = ops.decode_image(self.lm_head.weight.detach().float().contiguous())
            self._dec_images = (stamp, layers, head)
        return self._dec_images[1], self._dec_images[2]

    @torch.no_grad()
    def _decode_core_fused(self, mem2d, mlen, tokens_init, plen, N, S, T):
        """_decode_core on the fused launches: 5 per layer and step, 2 per step for the head (+ the cache warm-up)."""
        dev = mem2d.device
        C = mem2d.shape[1]
        L = len(self.h)
        heads, head_dim = self.h[0].heads, self.h[0].head_dim
        V = self.lm_head.weight.shape[0]
        layers, head_img = self._decode_images()
        tokens = tokens_init.clone()
        mem_cache = torch.empty((L, 2, N, heads, S, head_dim), dtype=torch.float16, device=dev)
        time_cache = torch.zeros((L, 2, N, heads, T, head_dim), dtype=torch.float16, device=dev)
        # cross-attention caches, warmed once (:324-334): the memory keys / values of all layers are ONE product and one store
        kv_weights = tuple(w for block in self.h for w in (block.mix_memory.k.weight, block.mix_memory.v.weight))
        ops.decode_memory_caches(linear(self._images, mem2d, kv_weights), mem_cache)
        table = ops.RopeTable(T, head_dim, dev)
        alive = torch.ones(2, N, dtype=torch.uint8, device=dev)             # double-buffered by step parity (halo_decode_token)
        out_len = torch.zeros(N, dtype=torch.int32, device=dev)
        log_probs = torch.zeros(N, dtype=torch.float32, device=dev)
        sum_entropies = torch.zeros(N, dtype=torch.float32, device=dev)
        a = torch.empty(N, 4 * C, device=dev, dtype=torch.float32)           # cross query | self q | k | v, then the MLP hidden rows
        att = torch.empty(N, 2 * C, device=dev, dtype=torch.float32)         # cross | self attention outputs
        hid = torch.empty(N, 4 * C, device=dev, dtype=torch.float32)
        logits = torch.empty(N, V, device=dev, dtype=torch.float32)
        wte = self.wte.weight.detach()
        y = ops.embed_fwd(tokens[:, 0:1], wte, None)                         # [N, C]; later steps: written by decode_token
        # The two accumulating products of a layer (K = 2C, 4C into C features: 128 workgroups at N = 64, each streaming its weights and
        # its rows over the whole K) run as two K-slices on twice the workgroups: the residual stream is the pair (y, side) -- slice 0
        # writes y = (y + side) + its half, slice 1 its half to the OTHER side buffer; the LayerNorm launches read y + side.  Every sum in
        # a fixed order (HALO_DECODE_KSPLIT=0: one slice, the round-4 launches).
        ksplit = os.environ.get('HALO_DECODE_KSPLIT', '1') != '0' and (2 * C) % 256 == 0
        sa, sb = (torch.empty(N, C, device=dev, dtype=torch.float32) for _ in range(2)) if ksplit else (None, None)
        for t in range(T):
            side = None                                                      # (y alone: the embedding of the step's token)
            for l, block in enumerate(self.h):
                w_qkv, w_proj, w_fc, w_fc2 = layers[l]
                ops.decode_linear(y, w_qkv, 4 * C, a, ln_weight=block.ln_time.weight, x_side=side)      # both attentions read ln_time(x) (:476-494)
                ops.decode_attention_pair(a, mem_cache[l, 0], mem_cache[l, 1], mlen, time_cache[l, 0], time_cache[l, 1], t + 1, table, att)
                ops.decode_linear(att, w_proj, C, y, accumulate=True, side_in=side, side_out=sa)         # x += cross proj + self proj
                ops.decode_linear(y, w_fc, 4 * C, hid, ln_weight=block.ln_chan.weight, gelu=True, x_side=sa)
                ops.decode_linear(hid, w_fc2, C, y, accumulate=True, side_in=sa, side_out=sb)
                side = sb
            ops.decode_linear(y, head_img, V, logits, ln_weight=self.ln_f.weight, x_side=side)
            ops.decode_token(logits, tokens, t, plen, ETX, alive, out_len, log_probs, sum_entropies, wte, y if t + 1 < T else None)
        return tokens, out_len, log_probs, sum_entropies

    def _decode_graph(self, key, mem2d, mlen, tokens, plen, N, S, T):
        """One HIP graph per (shape, parameter version): ~13 launches x layers x steps become one replay."""
        stamp = tuple((p._version, p.data_ptr()) for p in self.parameters())
        entry = self._graphs.get(key)
        if entry is None or entry['stamp'] != stamp:
            core = self._decode_core_fused if self._fused_decode_ok(mem2d.shape[1]) else self._decode_core
            static = (mem2d.clone(), mlen.clone(), tokens.clone())
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                core(*static, plen, N, S, T)                            # warm-up: weight images are built outside the capture
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                outs = core(*static, plen, N, S, T)
            # the graph reads the cached weight images by address: keep them alive as long as the graph
            held = [list(m._images._cache.values()) for m in self.modules() if hasattr(m, '_images')] + [getattr(self, '_dec_images', None)]
            if len(self._graphs) >= 8:
                self._graphs.pop(next(iter(self._graphs)))
            entry = dict(stamp=stamp, graph=graph, static=static, outs=outs, held=held)
            self._graphs[key] = entry
        for dst, src in zip(entry['static'], (mem2d, mlen, tokens)):
            dst.copy_(src)
        entry['graph'].replay()
        return tuple(o.clone() for o in entry['outs'])

    @torch.no_grad()
    def decode(self, features, input_lengths, target_lengths, prompt=None):
        "Perform batched greedy decoding (ha/transformer.py:124-199)."
        _require_inference(self, features)
        dev = features.device
        N, S, C = features.shape
        T = int(target_lengths.max().item()) + 1
        if prompt is None:
            tokens = torch.full((N, T + 1), ETX, dtype=torch.long, device=dev)
            tokens[:, 0] = STX
            plen = 0
        else:
            P = prompt.shape[-1]
            tokens = torch.full((N, T + 1 + P), ETX, dtype=torch.long, device=dev)
            tokens[:, 0] = STX
            tokens[:, 1:1 + P] = prompt.to(dev)
            plen = 1
        mlen = input_lengths.to(device=dev, dtype=torch.int32).contiguous()
        mem2d = features.reshape(N * S, C).float().contiguous()
        if os.environ.get('HALO_DECODE_GRAPH', '1') != '0':
            key = (N, S, T, tokens.shape[1], plen, str(dev), _lib.get_math_mode(), self._fused_decode_ok(C))
            tokens, out_len, log_probs, sum_entropies = self._decode_graph(key, mem2d, mlen, tokens, plen, N, S, T)
        else:
            core = self._decode_core_fused if self._fused_decode_ok(C) else self._decode_core
            tokens, out_len, log_probs, sum_entropies = core(mem2d, mlen, tokens, plen, N, S, T)
        output_lengths = out_len.to(input_lengths.dtype)
        lens = output_lengths.tolist()
        outputs = torch.nested.nested_tensor([p[1:l] for p, l in zip(tokens, lens)])
        alignments = [None] * N
        return outputs, output_lengths, alignments, log_probs, sum_entropies


class CTCAttentionDecoder(nn.Module):
    "CTC loss on the encoder, CE loss on the decoder (ha/transformer.py:34-57)"
    def __init__(self, *, vocab: int, head_dim: int, heads: int, p_drop: float, layers: int):
        super().__init__()
        self.decoder = Decoder(vocab=vocab, head_dim=head_dim, heads=heads, p_drop=p_drop, layers=layers)
        self.recognizer = TemporalClassifier(feat_dim=head_dim * heads, vocab_size=vocab)

    def forward(self, features, condtargets, input_lengths=None, condtarget_lengths=None, star_penalty=None,
                measure_entropy=False, drop_labels=False):
        # remove prompts for CTC. we assume there is only one prompt token
        targets = condtargets[:, 1:]
        target_lengths = condtarget_lengths - 1 if condtarget_lengths is not None else None
        decoder_loss, decoder_stats = self.decoder(features, condtargets, input_lengths, condtarget_lengths, star_penalty,
                                                   measure_entropy, drop_labels)
        recognizer_loss, recognizer_stats = self.recognizer(features, targets, input_lengths, target_lengths, star_penalty)
        return decoder_loss + 0.3 * recognizer_loss, {**decoder_stats, **recognizer_stats}

    def decode(self, features, input_lengths, target_lengths, prompt=None):
        return self.decoder.decode(features, input_lengths, target_lengths, prompt=prompt)


class AudioEncoder(nn.Module):
    def __init__(self, *, head_dim: int = 64, heads: int = 12, p_drop: float = 0.2, layers: int = 12, input_dim: int = 80,
                 conv_dim: int = 256, conv_strides: tuple = (2, 2, 2)):
        super().__init__()
        self.head_dim = head_dim
        self.heads = heads
        self.conv = ConvEncoder(input_dim=input_dim, hidden_dim=conv_dim, output_dim=head_dim * heads, strides=conv_strides)
        self.drop = nn.Dropout(p_drop)
        self.h = nn.ModuleList([Block(head_dim=head_dim, heads=heads, p_drop=p_drop) for _ in range(layers)])
        self.ln_f = LayerNorm(head_dim * heads, bias=False)
        self.dropout_stream = DropoutStream()
        self._graphs = {}                                                 # captured inference forwards, see _forward_graph

    def subsampled_lengths(self, input_lengths):
        return self.conv.subsampled_lengths(input_lengths)

    def forward(self, x, input_lengths, measure_entropy=False):
        """x [N, T, F] -> (features [N, T', C], lengths int32, stats); no time mask, like the reference (:245-247)."""
        _check_device_and_dropout(self, x)
        if _wants_grad(self):
            if measure_entropy:
                raise NotImplementedError('measure_entropy is inference-only here: call under torch.no_grad()')
            out = _EncoderFn.apply(self, x, *[p for p in self.parameters() if p.requires_grad])
            ninf = [torch.tensor(float('-inf'))] * len(self.h)
            return out, self.conv.subsampled_lengths(input_lengths), Stats(meme_entropy=list(ninf), self_entropy=list(ninf))._asdict()
        out_lengths = self.conv.subsampled_lengths(input_lengths)
        if not measure_entropy and os.environ.get('HALO_ENCODER_GRAPH', '1') != '0':
            out = self._forward_graph(x.float().contiguous())
            ninf = [torch.tensor(float('-inf'))] * len(self.h)
            return out, out_lengths, Stats(meme_entropy=list(ninf), self_entropy=list(ninf))._asdict()
        out, stats = self._forward_core(x, measure_entropy)
        return out, out_lengths, stats._asdict()

    @torch.no_grad()
    def _forward_core(self, x, measure_entropy=False):
        y = self.conv.forward_cl(x)                                              # channels-last: no .mT round trip
        N, T, C = y.shape
        y2d = y.view(N * T, C)
        stats = Stats(meme_entropy=[], self_entropy=[])
        for block in self.h:
            m_ent, t_ent = block._forward2d(y2d, N, T, measure_entropy=measure_entropy)
            stats.meme_entropy.append(m_ent)
            stats.self_entropy.append(t_ent)
        return ops.layernorm_fwd(y2d, self.ln_f.weight).view(N, T, C), stats

    def _forward_graph(self, x):
        """Inference forward replayed from one HIP graph per (input shape, arithmetic mode, parameter version): the ~17 small
        launches per block are host-launch bound when issued eagerly."""
        key = (tuple(x.shape), str(x.device), _lib.get_math_mode())
        stamp = tuple((p._version, p.data_ptr()) for p in self.parameters())
        entry = self._graphs.get(key)
        if entry is None or entry['stamp'] != stamp:
            static = x.clone()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self._forward_core(static)                              # warm-up: weight images are built outside the capture
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out, _ = self._forward_core(static)
            held = [list(m._images._cache.values()) for m in self.modules() if hasattr(m, '_images')] + [getattr(self, '_dec_images', None)]
            if len(self._graphs) >= 8:
                self._graphs.pop(next(iter(self._graphs)))
            entry = dict(stamp=stamp, graph=graph, static=static, out=out, held=held)
            self._graphs[key] = entry
        entry['static'].copy_(x)
        entry['graph'].replay()
        return entry['out'].clone()

    @torch.no_grad()
    def _forward_train(self, x):
        y, conv_saved = self.conv._forward_cl_train(x)
        N, T, C = y.shape
        sites = DropSites(self.dropout_stream.next(_p_drop(self), self.training))
        s_in = sites.next()
        y2d = drop_rows(y.view(N * T, C), s_in)                                    # self.drop(x), ha/transformer.py:239
        blocks = []
        for block in self.h:
            y2d, sv = block._forward2d_train(y2d, N, T, sites=sites)
            blocks.append(sv)
        out = ops.layernorm_fwd(y2d, self.ln_f.weight)
        return out.view(N, T, C), (conv_saved, blocks, y2d, (N, T, C), s_in)

    @torch.no_grad()
    def _backward_train(self, saved, dout, put):
        conv_saved, blocks, y_last, (N, T, C), s_in = saved
        dy, dw, _ = ops.layernorm_bwd(dout.reshape(N * T, C), y_last, self.ln_f.weight)
        put(self.ln_f.weight, dw)
        for block, sv in zip(reversed(self.h), reversed(blocks)):
            dy = block._backward2d(sv, dy, put)
        self.conv._backward_cl(conv_saved, drop_rows(dy, s_in).view(N, T, C), put)
