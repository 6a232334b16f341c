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
