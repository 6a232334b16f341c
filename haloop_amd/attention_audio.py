"""Drop-in for ha/attention_audio.py: the GPT-block audio encoders that `hac` builds for the `audio-encoder*` archs
(ha/init.py:132-160) and BASELINE config 5 names.

Built: ``AudioEncoder(config)`` with ``config.rotary_emb_dim == 0`` (the `audio-encoder` arch) -- Whisper-style front-end
``gelu(conv_pre) -> gelu(conv_subsample, stride 2)`` (ha/attention_audio.py:69-71,100-103), frozen sinusoid positions
(:10-16,87-88), dropout, ``n_layer`` bidirectional pre-LN GPT blocks (ha/attention.py:147-180 with ``config.causal = False``) and
``ln_f`` -- forward AND backward on the HIP operators: the two dense convolutions are channels-last unfold + GEMM (bias and exact
GELU in the epilogue at inference), the blocks are the ones haloop_amd.attention.GPT runs, and with grad enabled ``forward``
returns features whose ``grad_fn`` is the hand-written backward (conv_subsample back-propagates to its input through GEMM + fold,
``halo_col2im_cl``).  Same constructor, attribute and state-dict names as the reference (``conv_pre.*``, ``conv_subsample.*``,
``transformer.{wpe,h.{i}.*,ln_f}``), same return triple ``(features [B, T', C], lengths int32, {})``.

Not built (raises NotImplementedError, like haloop_amd.attention.Block): the rotary variants (``rotary_emb_dim != 0`` and
``StridingAudioEncoder``, which asserts it): the reference's rotary attention.Block needs flash_attn (ha/attention.py:155) and
cannot be constructed without it either.
"""
import math

import torch
import torch.nn as nn

from . import _lib, ops
from ._linear import DropSites, WeightImages, drop_rows, linear, linear_dw, linear_dx, training_images
from .attention import Block, LayerNorm, block_backward, block_forward, block_forward_train, prefetch_block_weights
from .rnn import DropoutStream


def sinusoids(length, channels, max_timescale=10000):
    """Returns sinusoids for positional embedding (ha/attention_audio.py:10-16; a constant table built once at construction)."""
    assert channels % 2 == 0
    scales = torch.arange(channels // 2) / (channels // 2 - 1)
    inv_timescales = torch.exp(-math.log(max_timescale) * scales)
    scaled_time = torch.arange(length)[:, None] * inv_timescales[None, :]
    return torch.cat([torch.sin(scaled_time), torch.cos(scaled_time)], dim=1)


class StridingAudioEncoder(nn.Module):
    def __init__(self, config):
        super().__init__()
        raise NotImplementedError('StridingAudioEncoder asserts rotary embeddings (ha/attention_audio.py:31), i.e. flash_attn blocks: not built')


class _EncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, x, *params):
        with training_images():
            out, saved = model._forward_train(x)
        ctx.model, ctx.saved, ctx.params = model, saved, params
        return out

    @staticmethod
    def backward(ctx, dout):
        grads = {}

        def put(p, g):
            if p is not None and g is not None and p.requires_grad:
                grads[id(p)] = g if id(p) not in grads else grads[id(p)] + g

        ctx.model._backward_train(ctx.saved, dout.contiguous().float(), put)
        ctx.saved = None
        return (None, None) + tuple(grads.get(id(p)) for p in ctx.params)


class AudioEncoder(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        if config.rotary_emb_dim:
            raise NotImplementedError('rotary AudioEncoder variants need flash_attn blocks in the reference (ha/attention.py:155) and are '
                                      'not built; the `audio-encoder` arch sets rotary_emb_dim = 0 (ha/init.py:133-135)')
        # whisper style convolutions
        self.conv_pre = nn.Conv1d(config.d_input, config.n_embd, kernel_size=3, stride=1, padding=1)
        self.conv_subsample = nn.Conv1d(config.n_embd, config.n_embd, kernel_size=3, stride=2, padding=1)
        self.transformer = nn.ModuleDict(dict(
            wpe=nn.Embedding(config.block_size, config.n_embd),
            drop=nn.Dropout(config.dropout),
            h=nn.ModuleList([Block(config) for _ in range(config.n_layer)]),
            ln_f=LayerNorm(config.n_embd, bias=config.bias),
        ))
        self.transformer.wpe.weight.data = sinusoids(config.block_size, config.n_embd)
        self.transformer.wpe.requires_grad_(False)
        self._images = WeightImages()
        self.dropout_stream = DropoutStream()

    def subsampled_lengths(self, input_lengths):
        # https://github.com/vdumoulin/conv_arithmetic (ha/attention_audio.py:92-97): float floor, int32 result
        p, k, s = self.conv_subsample.padding[0], self.conv_subsample.kernel_size[0], self.conv_subsample.stride[0]
        o = input_lengths + 2 * p - k
        o = torch.floor(o / s + 1)
        return o.int()

    def _check(self, x):
        if not x.is_cuda:
            raise _lib.HaloError('haloop_amd.attention_audio.AudioEncoder runs on the HIP device only (no CPU path)')

    def forward(self, x, input_lengths, measure_entropy=False):
        """x [B, T, F] -> (features [B, T', C], lengths int32, {}); ``measure_entropy`` is accepted and, as in the reference
        (ha/attention_audio.py:113-117), does not change what is returned."""
        self._check(x)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            out = _EncoderFn.apply(self, x, *[p for p in self.parameters() if p.requires_grad])
            return out, self.subsampled_lengths(input_lengths), {}
        if self.training and self.config.dropout > 0:
            raise NotImplementedError('training-mode dropout is built into the autograd path only: enable grad, or call .eval()')
        return self._forward_infer(x), self.subsampled_lengths(input_lengths), {}

    def _conv(self, conv, x3d, gelu):
        col, To = ops.im2col_cl(x3d, conv.kernel_size[0], conv.stride[0], conv.padding[0])
        y = linear(self._images, col, conv.weight, bias=conv.bias.detach(), gelu='erf' if gelu else False)
        return y, col, To

    @torch.no_grad()
    def _forward_infer(self, x):
        cfg = self.config
        x = x.float().contiguous()
        B = x.shape[0]
        y, _, T1 = self._conv(self.conv_pre, x, True)                               # F.gelu(conv_pre(x)), channels-last
        y, _, T = self._conv(self.conv_subsample, y.view(B, T1, -1), True)
        assert T <= cfg.block_size, f'Cannot forward sequence of length {T}, block size is only {cfg.block_size}'
        ops.add_rows_bcast_(y, self.transformer.wpe.weight.detach()[:T].contiguous(), T)   # x + wpe(pos)
        for blk in self.transformer.h:
            block_forward(self._images, blk, y, B, T, cfg)
        ln_f = self.transformer.ln_f
        return ops.layernorm_fwd(y, ln_f.weight, ln_f.bias).view(B, T, -1)

    # ---- training: forward keeping what the backward needs, and the backward -----------------------------------------
    @torch.no_grad()
    def _forward_train(self, x):
        cfg = self.config
        x = x.float().contiguous()
        B, T0, _ = x.shape
        a1, col1, T1 = self._conv(self.conv_pre, x, False)                          # pre-activations kept for the GELU backward
        y1 = ops.gelu_fwd(a1, exact=True)
        a2, col2, T = self._conv(self.conv_subsample, y1.view(B, T1, -1), False)
        y = ops.gelu_fwd(a2, exact=True)
        assert T <= cfg.block_size, f'Cannot forward sequence of length {T}, block size is only {cfg.block_size}'
        ops.add_rows_bcast_(y, self.transformer.wpe.weight.detach()[:T].contiguous(), T)
        # dropout sites in forward order: the embedding dropout (ha/attention_audio.py:110), then three per block
        sites = DropSites(self.dropout_stream.next(cfg.dropout, self.training))
        s_emb = sites.next()
        y = drop_rows(y, s_emb)
        blocks = []
        prefetch_block_weights(self._images, self.transformer.h, B * T)
        for blk in self.transformer.h:
            y, sv = block_forward_train(self._images, blk, y, B, T, cfg, sites)
            blocks.append(sv)
        ln_f = self.transformer.ln_f
        out = ops.layernorm_fwd(y, ln_f.weight, ln_f.bias)
        return out.view(B, T, -1), (col1, a1, col2, a2, blocks, y, s_emb, (B, T0, T1, T))

    @torch.no_grad()
    def _backward_train(self, saved, dout, put):
        cfg = self.config
        col1, a1, col2, a2, blocks, y_last, s_emb, (B, T0, T1, T) = saved
        C = cfg.n_embd
        ln_f = self.transformer.ln_f
        dy, dw, db = ops.layernorm_bwd(dout.reshape(B * T, C), y_last, ln_f.weight, None, ln_f.bias is not None)
        put(ln_f.weight, dw); put(ln_f.bias, db)
        for blk, sv in zip(reversed(self.transformer.h), reversed(blocks)):
            dy = block_backward(self._images, blk, sv, dy, B, T, cfg, put)
        dy = drop_rows(dy, s_emb)                                                   # wpe is frozen: nothing to collect for it
        # conv_subsample: y = gelu(col2 W2^T + b2)
        da2 = ops.gelu_bwd(dy, a2, exact=True)
        c2 = self.conv_subsample
        put(c2.weight, linear_dw(da2, col2).view_as(c2.weight))
        put(c2.bias, ops.colsum(da2))
        dcol2 = linear_dx(self._images, da2, c2.weight)                             # [B*T, C*3]
        dy1 = ops.col2im_cl(dcol2, B, T1, C, c2.kernel_size[0], c2.stride[0], c2.padding[0])
        # conv_pre: y1 = gelu(col1 W1^T + b1); its input is data (mel frames)
        da1 = ops.gelu_bwd(dy1.view(B * T1, C), a1, exact=True)
        c1 = self.conv_pre
        put(c1.weight, linear_dw(da1, col1).view_as(c1.weight))
        put(c1.bias, ops.colsum(da1))
