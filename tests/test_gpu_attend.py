"""`attend` / `attend_chunked` as the reference's own tests/test_attention.py calls them (ha/transformer.py:374-430): fp16 operands, head
dimension 7, T = 10 queries against S = 21 keys under a (T, S) triangle mask, ten seeds -- the chunked form against the full form at the
reference's tolerance, and both against the definition in float64 (softmax(q k^T / sqrt(hd), masked) v; entropy -sum att log(att + 1e-8))."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _definition(q, k, v, mask):
    qk = q.double() @ k.double().transpose(-2, -1) / math.sqrt(k.shape[-1])
    if mask is not None:
        qk = qk.masked_fill(mask, float('-inf'))
    att = qk.softmax(-1)
    return att @ v.double(), (-att * torch.log(att + 1e-8)).sum(-1).mean()


@pytest.mark.parametrize('seed', range(10))
def test_eq_attend_and_chunked(seed):
    from haloop_amd.transformer import attend, attend_chunked
    torch.manual_seed(seed)
    T, S = 10, 21
    q = torch.randn(2, 3, T, 7, device='cuda', dtype=torch.float16)
    k = torch.randn(2, 3, S, 7, device='cuda', dtype=torch.float16)
    v = torch.randn(2, 3, S, 7, device='cuda', dtype=torch.float16)
    causal_mask = torch.triu(q.new_ones(T, S), diagonal=1).bool()
    chunked = attend_chunked(q, k, v, causal_mask, chunk_size=2)[0]
    full, entropy = attend(q, k, v, causal_mask)
    assert chunked.dtype == q.dtype and chunked.shape == q.shape
    assert torch.allclose(chunked, full, atol=2e-3)                         # the reference's assertion
    want, want_ent = _definition(q.cpu(), k.cpu(), v.cpu(), causal_mask.cpu())
    np.testing.assert_allclose(full.float().cpu().numpy(), want.numpy(), atol=2e-3, rtol=0)      # fp16 output rounding
    assert abs(entropy.item() - want_ent.item()) <= 1e-5


@pytest.mark.parametrize('shape', [(2, 4, 33, 47, 64), (1, 2, 5, 300, 24), (3, 1, 1, 1, 5)])
def test_attend_with_general_masks(shape):
    """Per-batch, per-head and broadcast masks, head dimensions on and off the tiled kernels' list; fp32 in, fp32 out."""
    from haloop_amd import ops
    from haloop_amd.transformer import attend
    N, H, T, S, hd = shape
    g = torch.Generator().manual_seed(sum(shape))
    q, k, v = (torch.randn(N, H, n, hd, generator=g).cuda() for n in (T, S, S))
    for mshape in ((N, H, T, S), (N, 1, T, S), (T, S), (N, 1, 1, S), None):
        mask = None
        if mshape is not None:
            mask = torch.rand(mshape, generator=g) < 0.4
            mask[..., 0] = False                                           # every query keeps a key
            mask = mask.cuda()
        got, ent = attend(q, k, v, mask)
        want, want_ent = _definition(q.cpu(), k.cpu(), v.cpu(), None if mask is None else mask.cpu())
        np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), atol=2e-5 if hd not in (16, 32, 64) or mask is not None else 5e-2, rtol=0)
        assert abs(ent.item() - want_ent.item()) <= (1e-5 if hd not in (16, 32, 64) or mask is not None else 5e-2)
    y, e = ops.attention_masked(q, k, v, None)
    assert y.shape == q.shape and e.shape == (N, H, T)


@pytest.mark.parametrize('mode', ['bf16', 'bf16x3'])
@pytest.mark.parametrize('N,heads,hd,T,causal,ragged,p', [(2, 3, 64, 300, True, False, 0.0), (2, 2, 64, 333, True, True, 0.1),
                                                         (1, 4, 32, 257, True, False, 0.0), (2, 2, 64, 200, False, True, 0.0)])
def test_attention_backward_dispatch_forms_agree(monkeypatch, mode, N, heads, hd, T, causal, ragged, p):
    """The backward sweeps' two dispatch forms -- two query blocks per wave + one tile per workgroup, longest first, against one block per
    wave + long/short tile pairs -- walk the same tiles in the same order per output element: bitwise equal gradients, on ragged sizes
    (T not a multiple of the 128-query tile), key lengths and dropout too.  (The defaults pick per shape; the environment forces either.)"""
    from haloop_amd import _lib, ops
    prev = _lib.get_math_mode()
    _lib.set_math_mode(mode)
    try:
        C = heads * hd
        g = torch.Generator().manual_seed(T)
        qkv = torch.randn(N * T, 3 * C, generator=g).cuda()
        dy = torch.randn(N * T, C, generator=g).cuda()
        lens = torch.tensor([T - (41 * n) % T for n in range(N)], dtype=torch.int32).cuda() if ragged else None
        q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
        drop = ops.Dropout(p, 5, 0) if p > 0 else ops.NO_DROPOUT
        y, lse, _ = ops.attention_fwd(q, k, v, N, heads, hd, T, T, causal=causal, key_lengths=lens, want_lse=True, drop=drop, stream_id=3)
        out = {}
        for name, qb, lf in (('new', '2', '1'), ('old', '1', '0')):
            monkeypatch.setenv('HALO_ATTN_BWD_QB', qb)
            monkeypatch.setenv('HALO_ATTN_DKV_LF', lf)
            d = torch.full_like(qkv, float('nan'))
            ops.attention_bwd(q, k, v, y, dy, lse, d[:, :C], d[:, C:2 * C], d[:, 2 * C:], N, heads, hd, T, T, causal=causal, key_lengths=lens,
                              drop=drop, stream_id=3)
            out[name] = d
        assert torch.isfinite(out['new']).all()
        assert torch.equal(out['new'], out['old'])
    finally:
        _lib.set_math_mode(prev)
