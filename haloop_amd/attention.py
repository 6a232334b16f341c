"""Drop-in for the scoring path of ha/attention.py: GPT.forward_all (attention.py:205-232) and the
modules under it, forward only, on the HIP operators of csrc/gpt.hip and the GEMMs.

Same constructor (a config object with the GPTConfig fields of ha/init.py:25-36), same attribute
and state-dict names (``transformer.{wte,wpe}.weight``, ``transformer.h.{i}.{ln_1,attn.c_attn,
attn.c_proj,ln_2,mlp.c_fc,mlp.c_proj}``, ``transformer.ln_f``, tied ``lm_head.weight``), so
checkpoints load unchanged and ``hap`` (ha/score.py:72-73) can call ``forward_all(...,
reduction='none')`` as is.  Arithmetic is fp32 state with split-bf16 (bf16x3) or exact-f32 MFMA
GEMMs per ``halo_set_math_mode``; attention, LayerNorm, softmax and the loss are fp32.

Not built yet (raises NotImplementedError): the backward pass (``hala`` training), the KV-cache
``past`` argument / ``generate``, ``stable_embedding`` and rotary (flash_attn) blocks.
"""
import math
from dataclasses import dataclass, asdict

import torch
import torch.nn as nn

from . import _lib, ops


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


class _WeightImages:
    """Split (bf16 hi/lo, tiled) images of Linear weights, rebuilt only when a weight changes."""

    def __init__(self):
        self._cache = {}

    def get(self, w):
        key = id(w)
        hit = self._cache.get(key)
        if hit is None or hit[0] != w._version or hit[1] != w.data_ptr():
            hit = (w._version, w.data_ptr(), ops.split_image(w.detach().contiguous()))
            self._cache[key] = hit
        return hit[2]


class GPT(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        if config.stable_embedding:
            raise NotImplementedError('stable_embedding (LayerNorm-ed embeddings) is not built yet')
        if not config.causal:
            raise NotImplementedError('only the causal LM configuration is built')
        self.transformer = nn.ModuleDict(dict(
            wte=nn.Embedding(config.vocab_size, config.n_embd),
            wpe=nn.Embedding(config.block_size, config.n_embd),
            drop=nn.Dropout(config.dropout),
            h=nn.ModuleList([Block(config) for _ in range(config.n_layer)]),
            ln_f=LayerNorm(config.n_embd, bias=config.bias),
        ))
        with torch.no_grad():
            self.transformer.wpe.weight.mul_(0)
            self.transformer.wte.weight.mul_(0.02)
        self.lm_head = nn.Linear(config.n_embd, config.vocab_size, bias=False)
        self.transformer.wte.weight = self.lm_head.weight       # weight tying
        self._images = _WeightImages()

    # ---- one Linear: y = x W^T + b, with the epilogue fused ------------------------------------
    def _linear(self, x2d, lin, out=None, gelu=False, accumulate=False):
        M, K = x2d.shape
        N = lin.weight.shape[0]
        if _lib.get_math_mode() == 'bf16x3' and K >= 64 and N >= 64:
            return ops.gemm_split(ops.split_image(x2d), self._images.get(lin.weight), M, N, K, out=out, bias1=lin.bias,
                                  gelu=gelu, accumulate=accumulate)
        return ops.gemm(x2d, lin.weight, True, True, M, N, K, out=out, bias1=lin.bias, gelu=gelu, accumulate=accumulate)

    @torch.no_grad()
    def _trunk(self, input_ids):
        cfg = self.config
        B, T = input_ids.shape
        assert T <= cfg.block_size, f'Cannot forward sequence of length {T}, block size is only {cfg.block_size}'
        tr = self.transformer
        x = ops.embed_fwd(input_ids, tr.wte.weight, tr.wpe.weight, 0)                    # [B*T, C], the residual stream
        for blk in tr.h:
            h = ops.layernorm_fwd(x, blk.ln_1.weight, blk.ln_1.bias)
            qkv = self._linear(h, blk.attn.c_attn)
            y = ops.attention_causal_fwd(qkv, B, T, cfg.n_head)
            self._linear(y, blk.attn.c_proj, out=x, accumulate=True)                     # x += c_proj(y)
            h = ops.layernorm_fwd(x, blk.ln_2.weight, blk.ln_2.bias)
            h = self._linear(h, blk.mlp.c_fc, gelu=True)
            self._linear(h, blk.mlp.c_proj, out=x, accumulate=True)                      # x += mlp(h)
        return ops.layernorm_fwd(x, tr.ln_f.weight, tr.ln_f.bias)

    def forward_all(self, input_ids, target_ids, past=None, reduction='mean'):
        if past is not None:
            raise NotImplementedError('KV-cache continuation is not built yet')
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise NotImplementedError('haloop_amd.attention.GPT is forward-only so far: call it under torch.no_grad() / '
                                      'inference_mode (ha/score.py does); training backward is the next row of SURVEY.md 8f')
        if not input_ids.is_cuda:
            raise _lib.HaloError('haloop_amd.attention.GPT runs on the HIP device only (no CPU path)')
        if self.training and self.config.dropout > 0:
            raise NotImplementedError('dropout in the GPT path is not built; call .eval()')
        B, T = input_ids.shape
        V = self.config.vocab_size
        x = self._trunk(input_ids)
        targets = target_ids.reshape(-1)
        # lm_head + cross-entropy in row chunks so the [rows, V] logits stay bounded (206 MB per 1024 rows at V=50304)
        loss = torch.empty(B * T, device=x.device, dtype=torch.float32)
        chunk = max(64, min(B * T, (1 << 28) // (4 * V)))
        for r0 in range(0, B * T, chunk):
            r1 = min(B * T, r0 + chunk)
            logits = self._linear(x[r0:r1], self.lm_head)
            loss[r0:r1] = ops.cross_entropy_fwd(logits, targets[r0:r1], ignore_index=0)
        if reduction == 'none':
            return loss
        valid = (targets != 0)
        if reduction == 'sum':
            return loss.sum()
        if reduction == 'mean':
            return loss.sum() / valid.sum()
        raise ValueError(f'unknown reduction {reduction!r}')

    def forward(self, input_ids, past=None):
        raise NotImplementedError('generation (KV cache) is not built yet; forward_all is')
