"""Drop-in for the scoring path of ha/attention.py: GPT.forward_all (attention.py:205-232) and the
modules under it, forward only, on the HIP operators of csrc/gpt.hip and the GEMMs.

Same constructor (a config object with the GPTConfig fields of ha/init.py:25-36), same attribute
and state-dict names (``transformer.{wte,wpe}.weight``, ``transformer.h.{i}.{ln_1,attn.c_attn,
attn.c_proj,ln_2,mlp.c_fc,mlp.c_proj}``, ``transformer.ln_f``, tied ``lm_head.weight``), so
checkpoints load unchanged and ``hap`` (ha/score.py:72-73) can call ``forward_all(...,
reduction='none')`` as is.  Arithmetic is fp32 state with split-bf16 (bf16x3) or exact-f32 MFMA
GEMMs per ``halo_set_math_mode``; attention, LayerNorm, softmax and the loss are fp32.

Training: with grad enabled ``forward_all`` returns a loss with a ``grad_fn`` (one autograd.Function whose
backward is the hand-written HIP backward: cross-entropy, lm_head, LayerNorm, GELU, attention and embedding
gradients), so ``loss.backward()``, ``clip_grad_norm_`` and the optimizers of ha/attention_loop.py work unchanged.

Generation: ``forward(input_ids, past)`` / ``forward_context`` / ``generate`` keep the reference's fp32 KV cache
layout ``[L, 2, B, nh, T, hs]`` (attend_cached, ha/attention.py:64-93); attention reads the cache in place.

Training-mode dropout (config.dropout > 0): Philox masks at the reference's four kinds of site (embeddings, attention
probabilities inside the attention kernels, c_proj and MLP outputs as GEMM epilogues), see haloop_amd/transformer.py.

``stable_embedding`` (ha/attention.py:30-61: each embedding followed by its own LayerNorm) is built, forward and backward.

Not built (raises NotImplementedError): rotary (flash_attn) blocks -- the reference itself cannot construct them without
flash_attn.
"""
import math
import os
from dataclasses import dataclass, asdict

import torch
import torch.nn as nn

from . import _lib, ops
from ._linear import (SMALL_M, DropSites, training_images, WeightImages, drop_rows, forward_images, grad_images, linear, linear_dw, linear_dx, rowmajor_ok,
                      ln_linear, use_split)
from .rnn import DropoutStream


@dataclass
class GPTConfig:
    """ha/init.py:25-36."""
    block_size: int = 1024
    vocab_size: int = 50304
    n_layer: int = 12
    n_head: int = 12
    n_embd: int = 768
    dropout: float = 0.0
    bias: bool = False
    stable_embedding: bool = False
    causal: bool = True
    d_input: int = 1
    rotary_emb_dim: int = 0

    def state_dict(self):
        return asdict(self)


def new_gelu(x):
    """Kept for surface parity (ha/attention.py:12-17); the fused path applies it in the GEMM epilogue."""
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


class LayerNorm(nn.Module):
    def __init__(self, ndim, bias):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(ndim))
        self.bias = nn.Parameter(torch.zeros(ndim)) if bias else None

    def forward(self, input):
        shp = input.shape
        return ops.layernorm_fwd(input.reshape(-1, shp[-1]).contiguous(), self.weight, self.bias, 1e-5).view(shp)


class StableEmbedding(nn.Embedding):
    """nn.Embedding followed by its own LayerNorm (ha/attention.py:30-61); GPT runs the gather and the norm on the HIP ops."""

    def __init__(self, num_embeddings, embedding_dim, **kw):
        super().__init__(num_embeddings, embedding_dim, **kw)
        self.norm = nn.LayerNorm(embedding_dim)

    def reset_parameters(self):
        nn.init.xavier_uniform_(self.weight)
        self._fill_padding_idx_with_zero()


class MonitoredSelfAttention(nn.Module):
    def __init__(self, config):
        super().__init__()
        assert config.n_embd % config.n_head == 0
        self.c_attn = nn.Linear(config.n_embd, 3 * config.n_embd, bias=config.bias)
        self.c_proj = nn.Linear(config.n_embd, config.n_embd, bias=config.bias)
        self.resid_dropout = nn.Dropout(config.dropout)
        self.n_head, self.n_embd, self.dropout, self.causal = config.n_head, config.n_embd, config.dropout, config.causal


class MLP(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.c_fc = nn.Linear(config.n_embd, 4 * config.n_embd, bias=config.bias)
        self.c_proj = nn.Linear(4 * config.n_embd, config.n_embd, bias=config.bias)
        self.dropout = nn.Dropout(config.dropout)


class Block(nn.Module):
    def __init__(self, config):
        super().__init__()
        if config.rotary_emb_dim:
            raise NotImplementedError('rotary blocks need flash_attn in the reference and are not built here')
        self.ln_1 = LayerNorm(config.n_embd, bias=config.bias)
        self.attn = MonitoredSelfAttention(config)
        self.ln_2 = LayerNorm(config.n_embd, bias=config.bias)
        self.mlp = MLP(config)


# ---- one pre-LN GPT block (ha/attention.py:147-180), shared by GPT and haloop_amd.attention_audio.AudioEncoder ----------------
def block_forward(images, blk, x, B, T, cfg):
    """Inference: the residual stream x [B*T, C] is updated in place."""
    C, H = cfg.n_embd, cfg.n_head
    qkv, _ = ln_linear(images, x, blk.ln_1.weight, blk.ln_1.bias, blk.attn.c_attn.weight, bias=blk.attn.c_attn.bias)
    y, _, _ = ops.attention_fwd(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], B, H, C // H, T, T, causal=cfg.causal)
    linear(images, y, blk.attn.c_proj.weight, bias=blk.attn.c_proj.bias, out=x, accumulate=True)           # x += c_proj(y)
    if rowmajor_ok(B * T, 4 * C, C) and C % 32 == 0:
        # gelu(c_fc(ln_2(x))) leaves its GEMM as row-major bf16 (hi, lo) and the c_proj GEMM stages it from there: neither the fp32
        # activations nor an operand image of them are written
        h, _ = ln_linear(images, x, blk.ln_2.weight, blk.ln_2.bias, blk.mlp.c_fc.weight, bias=blk.mlp.c_fc.bias, gelu=True, out_rowmajor=True)
        linear(images, None, blk.mlp.c_proj.weight, bias=blk.mlp.c_proj.bias, out=x, accumulate=True, a_rowmajor=h, shape=(B * T, 4 * C))
        return x
    h, _ = ln_linear(images, x, blk.ln_2.weight, blk.ln_2.bias, blk.mlp.c_fc.weight, bias=blk.mlp.c_fc.bias, gelu=True)
    linear(images, h, blk.mlp.c_proj.weight, bias=blk.mlp.c_proj.bias, out=x, accumulate=True)              # x += mlp(h)
    return x


def prefetch_block_weights(images, blocks, M):
    """The weight images of every block's four Linears (forward image + the transposed one its backward reads) in a few multi-matrix
    launches at the head of a training forward, instead of one small launch per Linear on first use: they all went stale together
    at the optimizer step.  Only where linear() will take the split GEMM."""
    if _lib.get_math_mode() == 'f32' or M <= SMALL_M:
        return
    sets = [(w,) for blk in blocks for w in (blk.attn.c_attn.weight, blk.attn.c_proj.weight, blk.mlp.c_fc.weight, blk.mlp.c_proj.weight)
            if w.shape[0] >= 64 and w.shape[1] >= 64]
    images.prefetch_pairs(sets)


def block_forward_train(images, blk, x0, B, T, cfg, sites):
    """Training forward: returns (x_out, saved).  Dropout sites in forward order (ha/attention.py:90,127,141): attention
    probabilities, c_proj output, MLP output; the output dropouts are GEMM epilogues."""
    C, H = cfg.n_embd, cfg.n_head
    qkv, h1 = ln_linear(images, x0, blk.ln_1.weight, blk.ln_1.bias, blk.attn.c_attn.weight, bias=blk.attn.c_attn.bias, want_normed=True)
    s_att, s_res, s_mlp = sites.next(), sites.next(), sites.next()
    y, lse, _ = ops.attention_fwd(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], B, H, C // H, T, T, causal=cfg.causal, want_lse=True,
                                  drop=s_att[0], stream_id=s_att[1])
    # y and gelu(a) each feed one Linear now and its weight gradient later: both operand images come out of one read, and
    # gelu(a) is never written in fp32
    y_img, y_img_t = forward_images(y, C)
    x1 = linear(images, y, blk.attn.c_proj.weight, bias=blk.attn.c_proj.bias, residual=x0, drop=s_res[0],
                stream_id=s_res[1], a_image=y_img)
    a, h2 = ln_linear(images, x1, blk.ln_2.weight, blk.ln_2.bias, blk.mlp.c_fc.weight, bias=blk.mlp.c_fc.bias, want_normed=True)
    g_img, g_img_t = forward_images(a, C, ops.PAIR_GELU)
    g = ops.gelu_fwd(a) if g_img is None else None
    x = linear(images, g, blk.mlp.c_proj.weight, bias=blk.mlp.c_proj.bias, residual=x1, drop=s_mlp[0],
               stream_id=s_mlp[1], a_image=g_img, shape=a.shape)
    return x, (x0, h1, qkv, y, y_img_t, lse, x1, h2, a, g, g_img_t, s_att, s_res, s_mlp)


def block_backward(images, blk, saved, dx, B, T, cfg, put):
    """dx: gradient w.r.t. the block's output [B*T, C]; returns the gradient w.r.t. its input; parameter gradients go to put(p, g)."""
    C, H = cfg.n_embd, cfg.n_head
    M = B * T
    img = images
    x0, h1, qkv, y, y_img_t, lse, x1, h2, a, g, g_img_t, s_att, s_res, s_mlp = saved
    # x = x1 + drop(c_proj(gelu(c_fc(ln_2(x1)))))
    dm = drop_rows(dx, s_mlp)
    dm_img, dm_img_t = grad_images(dm, 4 * C)
    put(blk.mlp.c_proj.weight, linear_dw(dm, g, dy_image_t=dm_img_t, x_image_t=g_img_t, shapes=(dm.shape, a.shape)))
    if blk.mlp.c_proj.bias is not None: put(blk.mlp.c_proj.bias, ops.colsum(dm))
    dg = linear_dx(img, dm, blk.mlp.c_proj.weight, dy_image=dm_img)
    if blk.mlp.c_fc.bias is None and use_split(M, C, 4 * C) and use_split(4 * C, C, M):
        # da = dg * gelu'(a) is only ever a GEMM operand: write its two images, not the fp32 matrix
        da_img, da_img_t = ops.image_pair(dg, ops.PAIR_GELU_BWD, a)
        put(blk.mlp.c_fc.weight, linear_dw(None, h2, dy_image_t=da_img_t, shapes=(a.shape, h2.shape)))
        d_ln2 = linear_dx(img, None, blk.mlp.c_fc.weight, dy_image=da_img, shape=a.shape)
        del da_img, da_img_t
    else:
        da = ops.gelu_bwd(dg, a)
        put(blk.mlp.c_fc.weight, linear_dw(da, h2))
        if blk.mlp.c_fc.bias is not None: put(blk.mlp.c_fc.bias, ops.colsum(da))
        d_ln2 = linear_dx(img, da, blk.mlp.c_fc.weight)
    dx1, dw, db = ops.layernorm_bwd(d_ln2, x1, blk.ln_2.weight, dx, blk.ln_2.bias is not None)
    put(blk.ln_2.weight, dw); put(blk.ln_2.bias, db)
    # x1 = x0 + drop(c_proj(attention(c_attn(ln_1(x0)))))
    dr = drop_rows(dx1, s_res)
    dr_img, dr_img_t = grad_images(dr, C)
    put(blk.attn.c_proj.weight, linear_dw(dr, y, dy_image_t=dr_img_t, x_image_t=y_img_t))
    if blk.attn.c_proj.bias is not None: put(blk.attn.c_proj.bias, ops.colsum(dr))
    dy = linear_dx(img, dr, blk.attn.c_proj.weight, dy_image=dr_img)
    dqkv = torch.empty_like(qkv)
    ops.attention_bwd(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], y, dy, lse, dqkv[:, :C], dqkv[:, C:2 * C], dqkv[:, 2 * C:],
                      B, H, C // H, T, T, causal=cfg.causal, drop=s_att[0], stream_id=s_att[1])
    dq_img, dq_img_t = grad_images(dqkv, C)
    put(blk.attn.c_attn.weight, linear_dw(dqkv, h1, dy_image_t=dq_img_t))
    if blk.attn.c_attn.bias is not None: put(blk.attn.c_attn.bias, ops.colsum(dqkv))
    dx0, dw, db = ops.layernorm_bwd(linear_dx(img, dqkv, blk.attn.c_attn.weight, dy_image=dq_img), x0, blk.ln_1.weight, dx1,
                                    blk.ln_1.bias is not None)
    put(blk.ln_1.weight, dw); put(blk.ln_1.bias, db)
    return dx0


# ---- the same block in `bf16` arithmetic with ROW-MAJOR bf16 activations between the launches (round 3) ---------------------------
# Every Linear's input and output gradient exist once, as row-major bf16 written by the launch that produced them (the LayerNorm
# kernels, the two GELU passes); the forward and input-gradient products stage
# their A operand from those rows (halo_gemm_split_io), the weight-gradient products read both operands from them and transpose on the way
# from LDS into the MFMA (halo_gemm_tn_bf16).  Gone against block_forward_train / block_backward: the four operand-image launches of the
# activations per block and direction (gelu pair, gelu-backward pair, dy pairs, transposed LayerNorm outputs) and the fp32 round trips of
# gelu(a)'s gradient and of the normalised rows.  Same bf16 operand values, same fp32 accumulation.
def rowmajor_train_ok(cfg, blocks, M, training):
    """bf16 arithmetic, no biases, no output dropout, enough rows that the products without split-K fill the chip."""
    if os.environ.get('HALO_GPT_ROWMAJOR', '1') == '0' or _lib.get_math_mode() != 'bf16':
        return False
    C = cfg.n_embd
    if cfg.bias or (training and cfg.dropout > 0.0) or M % 32 != 0 or C % 32 != 0 or not rowmajor_ok(M, 4 * C, C) or not rowmajor_ok(M, C, C):
        return False
    if C // cfg.n_head not in (32, 64):               # the matrix-core attention kernels write the bf16 outputs
        return False
    return all(blk.attn.c_attn.bias is None and blk.mlp.c_fc.bias is None for blk in blocks)


def rows_ok(M, C):
    """The 256-row-tile products (halo_gemm_rows: single-pass bf16 arithmetic) take every Linear of a block of width C at M rows."""
    return os.environ.get('HALO_GPT_ROWS', '1') != '0' and ops.gemm_rows_supported(M, C, C) and C % 32 == 0


def block_forward_train_rm(images, blk, x0, B, T, cfg, sites):
    C, H, M = cfg.n_embd, cfg.n_head, B * T
    w = lambda lin: images.split((lin.weight,))
    rows = rows_ok(M, C)
    s_att, s_res, s_mlp = sites.next(), sites.next(), sites.next()
    if rows:
        _lib.lend_scratch(128 << 20, device=x0.device)      # K-slice slabs of the lm_head's input gradient and of the weight gradients' tails
        # round 5: every activation-by-weight product on halo_gemm_rows (256-row tiles cut to whole rounds of the CUs, A staged from the
        # row-major bf16 rows the producing launch left); c_fc's result and the MLP's hidden activations stay bf16 (what the reference's
        # autocast path holds there, ha/attention_loop.py:164) -- the fp32 [M, 4C] round trip between c_fc and new_gelu is gone
        h1b = ops.layernorm_bf16(x0, blk.ln_1.weight, blk.ln_1.bias)
        if C // H == 64 and os.environ.get('HALO_GPT_ATTN_B16', '1') != '0':
            # q | k | v stay bf16 between the c_attn product and the attention launches (forward and backward stage the rows as they are, two
            # tiles in flight: csrc/attn_b16.hip); the attention output is kept as bf16 only
            qkv = ops.gemm_rows(h1b, w(blk.attn.c_attn), M, 3 * C, C, out_bf16=True)
            y, lse, yb = ops.attention_fwd_b16(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], B, H, C // H, T, T, causal=cfg.causal, want_lse=True)
        else:
            qkv = ops.gemm_rows(h1b, w(blk.attn.c_attn), M, 3 * C, C)
            y, lse, yb = ops.attention_fwd_bf16(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], B, H, C // H, T, T, causal=cfg.causal,
                                                drop=s_att[0], stream_id=s_att[1])
        x1 = ops.gemm_rows(yb, w(blk.attn.c_proj), M, C, C, residual=x0)
        h2b = ops.layernorm_bf16(x1, blk.ln_2.weight, blk.ln_2.bias)
        if os.environ.get('HALO_GPT_GELU_EPILOGUE', '0') != '0':         # new_gelu in the c_fc product's epilogue (built, measured, off: at one
            # workgroup per CU nothing covers the epilogue's arithmetic -- 13.57 against 13.45 ms per step on one box)
            gb, a = ops.gemm_rows_gelu(h2b, w(blk.mlp.c_fc), M, 4 * C, C, keep_pre=True)
        else:
            a = ops.gemm_rows(h2b, w(blk.mlp.c_fc), M, 4 * C, C, out_bf16=True)
            gb = ops.gelu_b16(a)
        x = ops.gemm_rows(gb, w(blk.mlp.c_proj), M, C, 4 * C, residual=x1)
        return x, (x0, h1b, qkv, y, yb, lse, x1, h2b, a, gb, s_att)
    # (the normalised rows twice from one launch: the tiled image for the forward product, which stages an image 10-15 % faster than
    # rows from cold caches, and the row-major rows for the weight-gradient product)
    h1b, h1i = ops.layernorm_bf16(x0, blk.ln_1.weight, blk.ln_1.bias, want_image=True)
    qkv = ops.gemm_split(h1i, w(blk.attn.c_attn), M, 3 * C, C)
    del h1i
    y, lse, yb = ops.attention_fwd_bf16(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], B, H, C // H, T, T, causal=cfg.causal,
                                        drop=s_att[0], stream_id=s_att[1])
    x1 = ops.gemm_split_io((yb, None), w(blk.attn.c_proj), M, C, C, residual=x0)
    h2b, h2i = ops.layernorm_bf16(x1, blk.ln_2.weight, blk.ln_2.bias, want_image=True)
    a = ops.gemm_split(h2i, w(blk.mlp.c_fc), M, 4 * C, C)
    del h2i
    gb = ops.gelu_bf16(a)
    x = ops.gemm_split_io((gb, None), w(blk.mlp.c_proj), M, C, 4 * C, residual=x1)
    return x, (x0, h1b, qkv, y, yb, lse, x1, h2b, a, gb, s_att)


def block_backward_rm(images, blk, saved, dx, dxb, B, T, cfg, put):
    """dx / dxb: the gradient w.r.t. the block's output as fp32 and as row-major bf16 -> the same pair for its input."""
    C, H, M = cfg.n_embd, cfg.n_head, B * T
    wt = lambda lin: images.split_t((lin.weight,))
    x0, h1b, qkv, y, yb, lse, x1, h2b, a, gb, s_att = saved
    rows = a.dtype == torch.bfloat16                     # the forward ran on halo_gemm_rows
    # the four weight gradients contract over the same M token rows: collected here, ONE grouped launch at the end of the block
    # (whole-K tiles, no K-slices or reduce launches); HALO_GPT_DW_GROUP=0: one launch each, as round 4
    grouped = os.environ.get('HALO_GPT_DW_GROUP', '1') != '0' and M % 32 == 0
    # the two input gradients that only a LayerNorm backward reads leave their products as bf16 rows (that launch adds the fp32 residual
    # gradient to them in fp32)
    b16_ln = rows and C % 4 == 0 and C <= 2048 and os.environ.get('HALO_GPT_DLN_B16', '0') != '0'        # (measured: no gain, 12.40 against 12.40 ms; off)
    todo = []

    def dweight(p, dy_b, x_b):
        if grouped:
            todo.append((p, dy_b, x_b))
        else:
            put(p, ops.gemm_tn(dy_b, x_b))
    # x = x1 + c_proj(gelu(c_fc(ln_2(x1))))
    dweight(blk.mlp.c_proj.weight, dxb, gb)
    if rows:
        dab = ops.gelu_bwd_b16(ops.gemm_rows(dxb, wt(blk.mlp.c_proj), M, 4 * C, C, out_bf16=True), a)    # d a = (dx W) gelu'(a), bf16 throughout
    else:
        dab = ops.gelu_bwd_bf16(ops.gemm_split_io((dxb, None), wt(blk.mlp.c_proj), M, 4 * C, C), a)
    dweight(blk.mlp.c_fc.weight, dab, h2b)
    d_ln2 = ops.gemm_rows(dab, wt(blk.mlp.c_fc), M, C, 4 * C, out_bf16=b16_ln) if rows else ops.gemm_split_io((dab, None), wt(blk.mlp.c_fc), M, C, 4 * C)
    dx1, dw, db, dx1b = ops.layernorm_bwd(d_ln2, x1, blk.ln_2.weight, dx, blk.ln_2.bias is not None, want_bf16=True)
    put(blk.ln_2.weight, dw); put(blk.ln_2.bias, db)
    # x1 = x0 + c_proj(attention(c_attn(ln_1(x0))))
    dweight(blk.attn.c_proj.weight, dx1b, yb)
    dqkvb = torch.empty(M, 3 * C, device=dx.device, dtype=torch.bfloat16)
    if qkv.dtype == torch.bfloat16:                      # the forward kept q | k | v as bf16 rows: the output gradient arrives as bf16 rows too
        dyb = ops.gemm_rows(dx1b, wt(blk.attn.c_proj), M, C, C, out_bf16=True)
        ops.attention_bwd_b16(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], yb, dyb, lse, dqkvb[:, :C], dqkvb[:, C:2 * C], dqkvb[:, 2 * C:],
                              B, H, C // H, T, T, causal=cfg.causal)
    else:
        dy = ops.gemm_rows(dx1b, wt(blk.attn.c_proj), M, C, C) if rows else ops.gemm_split_io((dx1b, None), wt(blk.attn.c_proj), M, C, C)
        ops.attention_bwd_bf16(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], y, dy, lse, dqkvb[:, :C], dqkvb[:, C:2 * C], dqkvb[:, 2 * C:],
                               B, H, C // H, T, T, causal=cfg.causal, drop=s_att[0], stream_id=s_att[1])
    dweight(blk.attn.c_attn.weight, dqkvb, h1b)
    d_ln1 = ops.gemm_rows(dqkvb, wt(blk.attn.c_attn), M, C, 3 * C, out_bf16=b16_ln) if rows else ops.gemm_split_io((dqkvb, None), wt(blk.attn.c_attn), M, C, 3 * C)
    dx0, dw, db, dx0b = ops.layernorm_bwd(d_ln1, x0, blk.ln_1.weight, dx1, blk.ln_1.bias is not None, want_bf16=True)
    put(blk.ln_1.weight, dw); put(blk.ln_1.bias, db)
    if todo:
        for (p, _, _), g in zip(todo, ops.gemm_tn_group([(d, x_) for _, d, x_ in todo])):
            put(p, g)
    return dx0, dx0b


class _GPTLoss(torch.autograd.Function):
    """Per-token NLL of GPT.forward_all with its hand-written backward (the autograd graph the reference gets from
    torch for ha/attention.py:205-232).  The parameters ride along as inputs so that loss.backward() fills their
    .grad exactly like the reference's, and clip_grad_norm_ / the optimizer of ha/attention_loop.py:203-215 work as is."""

    @staticmethod
    def forward(ctx, model, input_ids, target_ids, *params):
        with training_images():
            per_tok, saved = model._forward_train(input_ids, target_ids)
        ctx.model, ctx.saved, ctx.params = model, saved, params
        return per_tok

    @staticmethod
    def backward(ctx, grad_per_tok):
        grads = ctx.model._backward_train(ctx.saved, grad_per_tok.contiguous().float())
        ctx.saved = None
        return (None, None, None) + tuple(grads.get(id(p)) for p in ctx.params)


class GPT(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.transformer = nn.ModuleDict(dict(
            wte=(StableEmbedding if config.stable_embedding else nn.Embedding)(config.vocab_size, config.n_embd),
            wpe=(StableEmbedding if config.stable_embedding else nn.Embedding)(config.block_size, config.n_embd),
            drop=nn.Dropout(config.dropout),
            h=nn.ModuleList([Block(config) for _ in range(config.n_layer)]),
            ln_f=LayerNorm(config.n_embd, bias=config.bias),
        ))
        with torch.no_grad():
            self.transformer.wpe.weight.mul_(0)
            self.transformer.wte.weight.mul_(0.02)
        self.lm_head = nn.Linear(config.n_embd, config.vocab_size, bias=False)
        self.transformer.wte.weight = self.lm_head.weight       # weight tying
        self._images = WeightImages()
        self.dropout_stream = DropoutStream()                   # Philox (seed, offset) per training forward

    # ---- one Linear: y = x W^T + b, with the epilogue fused ------------------------------------
    def _linear(self, x2d, lin, out=None, gelu=False, accumulate=False, site=(ops.NO_DROPOUT, 0), a_image=None, shape=None):
        return linear(self._images, x2d, lin.weight, bias=lin.bias, out=out, gelu=gelu, accumulate=accumulate, drop=site[0],
                      stream_id=site[1], a_image=a_image, shape=shape)

    def _embed(self, input_ids, t0=0, keep=False):
        """tok_emb + pos_emb [B*T, C]; with stable_embedding each goes through its own LayerNorm first.  keep: also return
        what the backward needs (raw token rows, raw position rows)."""
        tr = self.transformer
        if not self.config.stable_embedding:
            return ops.embed_fwd(input_ids, tr.wte.weight, tr.wpe.weight, t0), None
        T = input_ids.shape[1]
        et = ops.embed_fwd(input_ids, tr.wte.weight, None)
        ep = tr.wpe.weight.detach()[t0:t0 + T].contiguous()
        x = ops.layernorm_fwd(et, tr.wte.norm.weight, tr.wte.norm.bias)
        ops.add_rows_bcast_(x, ops.layernorm_fwd(ep, tr.wpe.norm.weight, tr.wpe.norm.bias), T)
        return x, ((et, ep) if keep else None)

    @torch.no_grad()
    def _trunk(self, input_ids, past=None, want_present=False):
        """Embedding + blocks + ln_f.  With ``past`` [L, 2, B, nh, T0, hs] (or want_present) keys/values go through a
        fp32 cache in the reference's layout (attend_cached, ha/attention.py:64-93) and ``present`` is returned."""
        cfg = self.config
        B, T = input_ids.shape
        t0 = 0 if past is None else past.size(-2)
        assert t0 + T <= cfg.block_size, f'Cannot forward sequence of length {t0 + T}, block size is only {cfg.block_size}'
        C, H = cfg.n_embd, cfg.n_head
        tr = self.transformer
        x, _ = self._embed(input_ids, t0)                                                # [B*T, C], the residual stream
        present = None
        if past is not None or want_present:
            present = torch.empty(cfg.n_layer, 2, B, H, t0 + T, C // H, device=x.device, dtype=torch.float32)
            if t0:
                present[..., :t0, :] = past
        for i, blk in enumerate(tr.h):
            qkv, _ = ln_linear(self._images, x, blk.ln_1.weight, blk.ln_1.bias, blk.attn.c_attn.weight, bias=blk.attn.c_attn.bias)
            if present is not None:
                ops.kv_cache_store(qkv[:, C:], C, present[i, 0], present[i, 1], B, T, H, C // H, t0)
                y = ops.attention_cached_fwd(qkv, present[i, 0], present[i, 1], T, t0 + T, causal=cfg.causal)
            else:
                y, _, _ = ops.attention_fwd(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], B, H, C // H, T, T, causal=cfg.causal)
            self._linear(y, blk.attn.c_proj, out=x, accumulate=True)                     # x += c_proj(y)
            if rowmajor_ok(B * T, 4 * C, C) and C % 32 == 0:          # as block_forward: the MLP's activations as row-major bf16
                h, _ = ln_linear(self._images, x, blk.ln_2.weight, blk.ln_2.bias, blk.mlp.c_fc.weight, bias=blk.mlp.c_fc.bias, gelu=True,
                                 out_rowmajor=True)
                linear(self._images, None, blk.mlp.c_proj.weight, bias=blk.mlp.c_proj.bias, out=x, accumulate=True, a_rowmajor=h,
                       shape=(B * T, 4 * C))
                continue
            h, _ = ln_linear(self._images, x, blk.ln_2.weight, blk.ln_2.bias, blk.mlp.c_fc.weight, bias=blk.mlp.c_fc.bias, gelu=True)
            self._linear(h, blk.mlp.c_proj, out=x, accumulate=True)                      # x += mlp(h)
        return ops.layernorm_fwd(x, tr.ln_f.weight, tr.ln_f.bias), present

    @torch.no_grad()
    def _trunk_rows(self, input_ids):
        """Embedding + blocks + ln_f in single-pass bf16 arithmetic on halo_gemm_rows (scoring without a KV cache): the residual stream
        fp32, updated in place by the products' residual epilogues; every Linear input row-major bf16.  -> ln_f(x) as row-major bf16."""
        cfg = self.config
        B, T = input_ids.shape
        assert T <= cfg.block_size, f'Cannot forward sequence of length {T}, block size is only {cfg.block_size}'
        C, H, M = cfg.n_embd, cfg.n_head, B * T
        tr = self.transformer
        w = lambda lin: self._images.split((lin.weight,))
        x, _ = self._embed(input_ids, 0)
        for blk in tr.h:
            h1b = ops.layernorm_bf16(x, blk.ln_1.weight, blk.ln_1.bias)
            if C // H == 64 and os.environ.get('HALO_GPT_ATTN_B16', '1') != '0':
                # q | k | v stay bf16 between the c_attn product and the attention launch, which stages them as they are
                qkv = ops.gemm_rows(h1b, w(blk.attn.c_attn), M, 3 * C, C, out_bf16=True)
                _, _, yb = ops.attention_fwd_b16(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], B, H, C // H, T, T, causal=cfg.causal)
            else:
                qkv = ops.gemm_rows(h1b, w(blk.attn.c_attn), M, 3 * C, C)
                _, _, yb = ops.attention_fwd_bf16(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], B, H, C // H, T, T, causal=cfg.causal)
            ops.gemm_rows(yb, w(blk.attn.c_proj), M, C, C, out=x, residual=x)                   # x += c_proj(y)
            h2b = ops.layernorm_bf16(x, blk.ln_2.weight, blk.ln_2.bias)
            if os.environ.get('HALO_GPT_GELU_EPILOGUE', '0') != '0':
                gb = ops.gemm_rows_gelu(h2b, w(blk.mlp.c_fc), M, 4 * C, C)
            else:
                gb = ops.gelu_b16(ops.gemm_rows(h2b, w(blk.mlp.c_fc), M, 4 * C, C, out_bf16=True))
            ops.gemm_rows(gb, w(blk.mlp.c_proj), M, C, 4 * C, out=x, residual=x)                # x += mlp(x)
        return ops.layernorm_bf16(x, tr.ln_f.weight, tr.ln_f.bias)

    def forward_all(self, input_ids, target_ids, past=None, reduction='mean'):
        if not input_ids.is_cuda:
            raise _lib.HaloError('haloop_amd.attention.GPT runs on the HIP device only (no CPU path)')
        B, T = input_ids.shape
        V = self.config.vocab_size
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            if past is not None:
                raise NotImplementedError('training through a KV-cache continuation is not built: call under torch.no_grad()')
            params = [p for p in self.parameters() if p.requires_grad]
            loss = _GPTLoss.apply(self, input_ids, target_ids, *params)          # per-token NLL with a grad_fn
            return self._reduce(loss, target_ids.reshape(-1), reduction)
        if self.training and self.config.dropout > 0:
            raise NotImplementedError('training-mode dropout is built into the autograd path only: enable grad, or call .eval()')
        targets = target_ids.reshape(-1)
        C = self.config.n_embd
        if past is None and B * T > SMALL_M and V % 8 == 0 and rows_ok(B * T, C) and rowmajor_train_ok(self.config, self.transformer.h, B * T, False):
            loss, _, _ = ops.gemm_rows_ce(self._trunk_rows(input_ids), self._images.split((self.lm_head.weight,)), B * T, V, C, targets, ignore_index=0)
            return self._reduce(loss, targets, reduction)
        x, _ = self._trunk(input_ids, past)
        if use_split(B * T, V, C) and B * T > SMALL_M:
            # lm_head + cross-entropy in the GEMM's epilogue: the [rows, V] logits are never written
            loss, _, _ = ops.gemm_split_ce(ops.split_image(x), self._images.split((self.lm_head.weight,)), B * T, V, C, targets, ignore_index=0)
            return self._reduce(loss, targets, reduction)
        # otherwise in row chunks so the logits stay bounded (206 MB per 1024 rows at V=50304)
        loss = torch.empty(B * T, device=x.device, dtype=torch.float32)
        chunk = max(64, min(B * T, (1 << 28) // (4 * V)))
        for r0 in range(0, B * T, chunk):
            r1 = min(B * T, r0 + chunk)
            logits = self._linear(x[r0:r1], self.lm_head)
            loss[r0:r1] = ops.cross_entropy_fwd(logits, targets[r0:r1], ignore_index=0)
        return self._reduce(loss, targets, reduction)

    @staticmethod
    def _reduce(loss, targets, reduction):
        if reduction == 'none':
            return loss
        if reduction == 'sum':
            return loss.sum()
        if reduction == 'mean':
            return loss.sum() / (targets != 0).sum()
        raise ValueError(f'unknown reduction {reduction!r}')

    # ---- training: forward that keeps what the backward needs, and the backward itself -------------------------
    @torch.no_grad()
    def _forward_train(self, input_ids, target_ids):
        cfg = self.config
        B, T = input_ids.shape
        assert T <= cfg.block_size, f'Cannot forward sequence of length {T}, block size is only {cfg.block_size}'
        C, H = cfg.n_embd, cfg.n_head
        tr = self.transformer
        # dropout sites in forward order (ha/attention.py:224,90,127,141): embeddings, then per block the attention
        # probabilities, the c_proj output and the MLP output; output dropouts are GEMM epilogues
        sites = DropSites(self.dropout_stream.next(cfg.dropout, self.training))
        s_emb = sites.next()
        x, emb_saved = self._embed(input_ids, 0, keep=True)
        x = drop_rows(x, s_emb)
        blocks = []
        prefetch_block_weights(self._images, tr.h, B * T)
        fwd = block_forward_train_rm if rowmajor_train_ok(cfg, tr.h, B * T, self.training) else block_forward_train
        for blk in tr.h:
            x, sv = fwd(self._images, blk, x, B, T, cfg, sites)
            blocks.append(sv)
        targets = target_ids.reshape(-1)
        if fwd is block_forward_train_rm and rows_ok(B * T, C) and cfg.vocab_size % 8 == 0:
            # round 5: ln_f's rows as row-major bf16, the lm_head product on halo_gemm_rows with the cross-entropy statistics in its
            # epilogue (from the fp32 accumulators) and the logits KEPT AS bf16 (1.65 GB of fp32 at B = 8, T = 1024 no longer written)
            xf = ops.layernorm_bf16(x, tr.ln_f.weight, tr.ln_f.bias)
            loss, row_lse, logits = ops.gemm_rows_ce(xf, self._images.split((self.lm_head.weight,)), B * T, cfg.vocab_size, C, targets,
                                                     ignore_index=0, want_logits=True, want_lse=True)
            return loss, (input_ids, targets, blocks, x, xf, logits, row_lse, s_emb, emb_saved)
        xf = ops.layernorm_fwd(x, tr.ln_f.weight, tr.ln_f.bias)
        if use_split(B * T, cfg.vocab_size, C) and B * T > SMALL_M:          # statistics in the GEMM epilogue; the logits are kept for the backward
            loss, row_lse, logits = ops.gemm_split_ce(ops.split_image(xf), self._images.split((self.lm_head.weight,)), B * T, cfg.vocab_size, C,
                                                      targets, ignore_index=0, want_logits=True, want_lse=True)
        else:
            logits = self._linear(xf, self.lm_head)
            loss, row_lse = ops.cross_entropy_fwd_lse(logits, targets, ignore_index=0)
        return loss, (input_ids, targets, blocks, x, xf, logits, row_lse, s_emb, emb_saved)

    @torch.no_grad()
    def _backward_train(self, saved, grad_per_tok):
        cfg = self.config
        input_ids, targets, blocks, x_last, xf, logits, row_lse, s_emb, emb_saved = saved
        B, T = input_ids.shape
        C, H = cfg.n_embd, cfg.n_head
        tr = self.transformer
        img = self._images
        grads = {}

        def put(p, g):
            if p is not None and p.requires_grad:
                grads[id(p)] = g if id(p) not in grads else grads[id(p)] + g

        M, V = logits.shape
        if logits.dtype == torch.bfloat16:
            # the stored bf16 logits become d loss / d logits IN PLACE: the row-major bf16 operand of both gradient products
            dl = ops.cross_entropy_bwd_bf16_(logits, targets, row_lse, grad_per_tok, ignore_index=0)
            dw_head = ops.gemm_tn_group([(dl, xf)])[0]                                       # [V, C]; the tied wte gradient lands here too
            dxf = ops.gemm_rows(dl, img.split_t((self.lm_head.weight,)), M, C, V)
            del dl, logits
        elif use_split(M, C, V) and use_split(V, C, M):
            # d loss / d logits goes straight into the two operand images of the lm_head's backward products
            dl_img, dl_img_t = ops.cross_entropy_bwd_images(logits, targets, row_lse, grad_per_tok, ignore_index=0)
            dw_head = linear_dw(None, xf, dy_image_t=dl_img_t, shapes=((M, V), xf.shape))    # [V, C]; the tied wte gradient lands here too
            dxf = linear_dx(img, None, self.lm_head.weight, dy_image=dl_img, shape=(M, V))
            del dl_img, dl_img_t
        else:
            dlogits = ops.cross_entropy_bwd_(logits, targets, row_lse, grad_per_tok, ignore_index=0)
            dw_head = linear_dw(dlogits, xf)
            dxf = linear_dx(img, dlogits, self.lm_head.weight)
        rm = len(blocks) > 0 and len(blocks[0]) == 11           # block_forward_train_rm's record
        dx, dw, db, *dxb = ops.layernorm_bwd(dxf, x_last, tr.ln_f.weight, None, tr.ln_f.bias is not None, want_bf16=rm)
        put(tr.ln_f.weight, dw); put(tr.ln_f.bias, db)
        for blk, sv in zip(reversed(tr.h), reversed(blocks)):
            if rm:
                dx, dxb[0] = block_backward_rm(img, blk, sv, dx, dxb[0], B, T, cfg, put)
            else:
                dx = block_backward(img, blk, sv, dx, B, T, cfg, put)
        dwpe = torch.zeros_like(tr.wpe.weight)
        dx = drop_rows(dx, s_emb)
        if emb_saved is not None:                                                # StableEmbedding: through the two LayerNorms first
            et, ep = emb_saved
            dpos = torch.empty_like(ep)
            ops.embed_bwd(input_ids, dx, None, dpos, 0)                          # dpos[t] = sum_b dx[b, t]
            dx, dw, db = ops.layernorm_bwd(dx, et, tr.wte.norm.weight, None, True)
            put(tr.wte.norm.weight, dw); put(tr.wte.norm.bias, db)
            dep, dw, db = ops.layernorm_bwd(dpos, ep, tr.wpe.norm.weight, None, True)
            put(tr.wpe.norm.weight, dw); put(tr.wpe.norm.bias, db)
            dwpe[:T] = dep
            ops.embed_bwd(input_ids, dx, dw_head, None, 0)                       # tied: token rows add into the lm_head gradient
        else:
            ops.embed_bwd(input_ids, dx, dw_head, dwpe, 0)
        put(self.lm_head.weight, dw_head)
        put(tr.wpe.weight, dwpe)
        return grads

    def forward_context(self, input_ids):
        """(ln_f(x) [B, T, C], present [L, 2, B, nh, T, hs]) -- ha/attention.py:234-251"""
        self._check_inference(input_ids)
        x, present = self._trunk(input_ids, None, want_present=True)
        return x.view(*input_ids.shape, -1), present

    def forward(self, input_ids, past=None):
        """(logits of the last position [B, 1, V], present) -- ha/attention.py:253-279"""
        self._check_inference(input_ids)
        B, T = input_ids.shape
        x, present = self._trunk(input_ids, past, want_present=True)
        last = x.view(B, T, -1)[:, -1, :].contiguous()
        return self._linear(last, self.lm_head).view(B, 1, -1), present

    def _check_inference(self, input_ids):
        if not input_ids.is_cuda:
            raise _lib.HaloError('haloop_amd.attention.GPT runs on the HIP device only (no CPU path)')
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise NotImplementedError('GPT.forward / forward_context (generation) are inference paths: call under torch.no_grad() / '
                                      'inference_mode; training goes through forward_all')


@torch.inference_mode()
def generate(self, input_ids, max_new_tokens, temperature=1.0, top_k=None, stop_token=50256):
    """Sampling loop of ha/attention.py:282-321 (same control flow and KV-cache use; the draw itself is torch.multinomial
    on the device, as in the reference)."""
    past = None
    for _ in range(max_new_tokens):
        if input_ids.size(1) >= self.config.block_size:
            past = None
            logits, _ = self(input_ids[:, -self.config.block_size:], past=None)
        elif past is None:
            logits, past = self(input_ids, past=None)
        else:
            logits, past = self(input_ids[:, [-1]], past=past)
        logits = logits[:, -1, :] / temperature
        if top_k is not None:
            v, _ = torch.topk(logits, min(top_k, logits.size(-1)))
            logits[logits < v[:, [-1]]] = -float('Inf')
        probs = torch.softmax(logits, dim=-1)
        input_ids_next = torch.multinomial(probs, num_samples=1)
        if input_ids_next == stop_token:
            break
        input_ids = torch.cat((input_ids, input_ids_next), dim=1)
        yield input_ids_next
