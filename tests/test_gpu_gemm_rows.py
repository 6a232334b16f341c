"""halo_gemm_rows / halo_gemm_rows_ce / halo_cross_entropy_bwd_bf16 / halo_gelu_b16 (round 5: the GPT path's activation-by-weight products on
256-row x 96 / 192 / 288-column tiles with row-major bf16 activations) against the fp64 product of the same bf16 values, and the GPT
training step that runs on them against the operand-image path."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.fixture
def bf16_mode():
    from haloop_amd import _lib
    _lib.lib(); _lib.lend_scratch(256 << 20)
    prev = _lib.get_math_mode()
    _lib.set_math_mode('bf16')
    yield
    _lib.set_math_mode(prev)


def _operands(M, N, K, seed):
    g = torch.Generator().manual_seed(seed)
    a = torch.randn(M, K, generator=g).to(DEV).bfloat16()
    w = torch.randn(N, K, generator=g).to(DEV)
    return a, w, g


# every tile width (N = 768 -> 96 columns, 3072 -> 192, 2304 -> 288 at 8192 rows), ragged rows / columns, one k-block, forced widths
@pytest.mark.parametrize('M,N,K,tn', [(8192, 768, 768, 0), (8192, 2304, 768, 0), (8192, 3072, 768, 0), (8192, 768, 3072, 0), (1000, 200, 160, 3),
                                      (777, 1000, 96, 6), (300, 584, 32, 9), (256, 96, 64, 0), (33, 8, 32, 0), (4096, 1536, 512, 9)])
def test_rows_product_matches_fp64(bf16_mode, monkeypatch, M, N, K, tn):
    from haloop_amd import ops
    if tn:
        monkeypatch.setenv('HALO_GEMM_ROWS_TN', str(tn))
    a, w, g = _operands(M, N, K, M + N + K)
    img = ops.split_image(w)
    assert ops.gemm_rows_supported(M, N, K)
    rows = torch.cat([torch.arange(0, min(M, 40)), torch.arange(max(0, M - 40), M), torch.randint(0, M, (64,), generator=g)]).to(DEV)
    want = a[rows].double() @ w.bfloat16().double().t()
    tol = 2e-6 * K ** 0.5 * 16 + 1e-5                                  # fp32 accumulation of exact bf16 products
    got = ops.gemm_rows(a, img, M, N, K)
    assert (got[rows].double() - want).abs().max().item() <= tol
    # + residual, into a fresh tensor and in place
    r = torch.randn(M, N, generator=g).to(DEV)
    got_r = ops.gemm_rows(a, img, M, N, K, residual=r)
    assert (got_r[rows].double() - (want + r[rows].double())).abs().max().item() <= tol + 1e-6
    x = r.clone()
    ops.gemm_rows(a, img, M, N, K, out=x, residual=x)
    assert torch.equal(x, got_r)
    # bf16 result = the fp32 result rounded once
    assert torch.equal(ops.gemm_rows(a, img, M, N, K, out_bf16=True), got.bfloat16())
    # A as a tiled image: the same bf16 operand values
    assert torch.equal(ops.gemm_rows(ops.split_image(a.float()), img, M, N, K), got)


@pytest.mark.parametrize('M,N,K', [(2048, 768, 16384), (8192, 768, 50304)])
def test_a_very_long_contraction_runs_as_two_k_slices(bf16_mode, M, N, K):
    """K >= 16384 under a result that the narrow tiles would cover in one round (the lm_head's input gradient): wider tiles, two K-slices
    through the lent scratch and a sum launch -- the fp64 product of the same bf16 values; without scratch the one-launch form, same values."""
    from haloop_amd import _lib, ops
    a, w, g = _operands(M, N, K, 7)
    img = ops.split_image(w)
    rows = torch.randint(0, M, (48,), generator=g).to(DEV)
    want = a[rows].double() @ w.bfloat16().double().t()
    got = ops.gemm_rows(a, img, M, N, K)
    tol = 2e-6 * K ** 0.5 * 16 + 1e-5
    assert (got[rows].double() - want).abs().max().item() <= tol
    _lib.lib().halo_set_scratch(None, 0)
    try:
        one = ops.gemm_rows(a, img, M, N, K)
    finally:
        _lib.check(_lib.lib().halo_set_scratch(_lib._scratch.data_ptr(), _lib._scratch.numel()), 'halo_set_scratch')
    assert (one[rows].double() - want).abs().max().item() <= tol
    assert (one - got).abs().max().item() <= 2 * tol


def test_rows_product_refusals(bf16_mode):
    from haloop_amd import _lib, ops
    a, w, _ = _operands(64, 64, 48, 1)                                  # K % 32 != 0
    assert not ops.gemm_rows_supported(64, 64, 48)
    with pytest.raises(_lib.HaloError):
        ops.gemm_rows(a, ops.split_image(w), 64, 64, 48)
    _lib.set_math_mode('bf16x3')                                       # single-pass bf16 only
    assert not ops.gemm_rows_supported(64, 64, 64)


@pytest.mark.parametrize('M,V,K', [(8192, 50304, 768), (515, 1000, 128), (256, 192, 64), (100, 8, 32)])
def test_lm_head_with_cross_entropy_epilogue_and_bf16_logits(bf16_mode, M, V, K):
    from haloop_amd import ops
    g = torch.Generator().manual_seed(V + M)
    x = (torch.randn(M, K, generator=g) * 0.5).to(DEV).bfloat16()
    w = (torch.randn(V, K, generator=g) * 0.2).to(DEV)
    tg = torch.randint(0, V, (M,), generator=g).to(DEV)
    tg[::7] = 0                                                         # ignored rows
    img = ops.split_image(w)
    loss, lse, logits = ops.gemm_rows_ce(x, img, M, V, K, tg, ignore_index=0, want_logits=True, want_lse=True)
    loss2, _, none = ops.gemm_rows_ce(x, img, M, V, K, tg, ignore_index=0)                       # scoring: no logits written
    assert none is None and torch.equal(loss, loss2)
    rows = torch.cat([torch.arange(0, min(M, 48)), torch.randint(0, M, (48,), generator=g)]).to(DEV)
    ref = x[rows].double() @ w.bfloat16().double().t()
    want = torch.logsumexp(ref, -1) - ref.gather(1, tg[rows][:, None])[:, 0]
    want = torch.where(tg[rows] == 0, torch.zeros_like(want), want)
    np.testing.assert_allclose(loss[rows].cpu().numpy(), want.cpu().numpy(), atol=2e-5 * K ** 0.5 + 1e-5)   # from the fp32 accumulators
    live = (tg[rows] != 0).cpu().numpy()
    np.testing.assert_allclose(lse[rows].cpu().numpy()[live], torch.logsumexp(ref, -1).cpu().numpy()[live], atol=2e-5 * K ** 0.5 + 1e-5)
    assert torch.equal(logits[rows], ref.float().bfloat16()) or (logits[rows].float() - ref.float()).abs().max().item() <= 2 ** -7 * ref.abs().max().item()
    # the backward to the logits, in place, against torch on the same bf16 logits
    gr = torch.rand(M, generator=g).to(DEV)
    lg = logits[rows].float()
    want_d = (torch.softmax(lg, -1) * torch.exp(torch.logsumexp(lg, -1) - lse[rows])[:, None] - torch.nn.functional.one_hot(tg[rows], V)) * gr[rows][:, None]
    want_d[tg[rows] == 0] = 0
    d = ops.cross_entropy_bwd_bf16_(logits, tg, lse, gr, ignore_index=0)
    assert d.data_ptr() == logits.data_ptr()
    assert (d[rows].float() - want_d).abs().max().item() <= 2 ** -8 * max(1e-3, want_d.abs().max().item()) + 1e-6


def test_gelu_on_bf16_rows(bf16_mode):
    from haloop_amd import ops
    g = torch.Generator().manual_seed(4)
    a = (torch.randn(512, 3072, generator=g) * 2).to(DEV).bfloat16()
    dg = torch.randn(512, 3072, generator=g).to(DEV).bfloat16()
    for exact in (False, True):
        assert torch.equal(ops.gelu_b16(a, exact), ops.gelu_fwd(a.float(), exact).bfloat16())
        assert torch.equal(ops.gelu_bwd_b16(dg, a, exact), ops.gelu_bwd(dg.float(), a.float(), exact).bfloat16())


def test_gelu_in_the_epilogue_equals_the_separate_pass(bf16_mode):
    from haloop_amd import ops
    for M, N, K in ((8192, 3072, 768), (300, 584, 64)):
        a, w, _ = _operands(M, N, K, 9)
        img = ops.split_image(w)
        pre = ops.gemm_rows(a, img, M, N, K, out_bf16=True)
        for exact in (False, True):
            g, p2 = ops.gemm_rows_gelu(a, img, M, N, K, exact=exact, keep_pre=True)
            assert torch.equal(p2, pre) and torch.equal(g, ops.gelu_b16(pre, exact))
            assert torch.equal(ops.gemm_rows_gelu(a, img, M, N, K, exact=exact), g)


def _train(monkeypatch, rows, seed=3):
    from haloop_amd import attention
    monkeypatch.setenv('HALO_GPT_ROWS', '1' if rows else '0')
    torch.manual_seed(seed)
    cfg = attention.GPTConfig(block_size=1024, vocab_size=2048, n_layer=2, n_head=12, n_embd=768)
    model = attention.GPT(cfg).to(DEV).train()
    with torch.no_grad():
        model.transformer.wpe.weight.normal_(0, 0.02)
    g = torch.Generator().manual_seed(seed)
    inputs = torch.randint(1, cfg.vocab_size, (8, 1024), generator=g).to(DEV)
    targets = torch.randint(1, cfg.vocab_size, (8, 1024), generator=g).to(DEV)
    loss = model.forward_all(inputs, targets, reduction='mean')
    loss.backward()
    model.eval()
    with torch.inference_mode():
        per_tok = model.forward_all(inputs, targets, reduction='none')
    return loss.item(), {k: p.grad.detach().clone() for k, p in model.named_parameters()}, per_tok


def test_gpt_step_on_the_row_tiles_against_the_128_tile_path(bf16_mode, monkeypatch):
    """The training step and the scoring pass on halo_gemm_rows (bf16 MLP activations, bf16 logits for the backward) against the same
    step on the 128 x 128-tile launches (fp32 activations between them): the same bf16 operand values enter every product except where an
    activation is now rounded to bf16 once more -- held at the bf16 gates (loss 2e-3 relative, every gradient's direction and norm)."""
    loss_r, g_r, tok_r = _train(monkeypatch, True)
    loss_o, g_o, tok_o = _train(monkeypatch, False)
    assert abs(loss_r - loss_o) <= 2e-3 * abs(loss_o), (loss_r, loss_o)
    assert (tok_r - tok_o).abs().max().item() <= 2e-2                   # nats per token, the BASELINE gate of the bf16 arithmetic
    for k in g_o:
        a, b = g_r[k].double().flatten(), g_o[k].double().flatten()
        cos = float(a @ b / (a.norm() * b.norm() + 1e-30))
        assert cos >= 0.998, (k, cos)
        assert abs(float(a.norm()) - float(b.norm())) <= 2e-2 * float(b.norm()), k


@pytest.mark.parametrize('N,T,heads,causal', [(8, 1024, 12, True), (2, 200, 3, True), (3, 129, 2, False), (1, 64, 1, True)])
def test_attention_forward_from_bf16_rows(bf16_mode, N, T, heads, causal):
    """halo_attention_fwd_b16 (bf16 q / k / v rows, two tiles in flight) against fp64 softmax attention of the same bf16 values and against
    the fp32-input matrix-core launch on those values."""
    from haloop_amd import ops
    hd, C = 64, heads * 64
    g = torch.Generator().manual_seed(T + heads)
    qkv = (torch.randn(N * T, 3 * C, generator=g) * 0.7).to(DEV).bfloat16()
    q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    y, lse, yb = ops.attention_fwd_b16(q, k, v, N, heads, hd, T, T, causal=causal, want_y=True, want_lse=True)
    qf = qkv.float()
    y0, lse0, _ = ops.attention_fwd_bf16(qf[:, :C], qf[:, C:2 * C], qf[:, 2 * C:], N, heads, hd, T, T, causal=causal)
    torch.testing.assert_close(y, y0, rtol=0, atol=2e-2)
    torch.testing.assert_close(lse, lse0, rtol=0, atol=2e-2)
    assert torch.equal(yb, y.bfloat16())
    n0 = N - 1
    Q = q[n0 * T:(n0 + 1) * T].double().view(T, heads, hd).transpose(0, 1)
    K = k[n0 * T:(n0 + 1) * T].double().view(T, heads, hd).transpose(0, 1)
    V = v[n0 * T:(n0 + 1) * T].double().view(T, heads, hd).transpose(0, 1)
    S = Q @ K.transpose(1, 2) / 8.0
    if causal:
        S = S.masked_fill(torch.ones(T, T, device=DEV).triu(1).bool(), float('-inf'))
    want = (torch.softmax(S, -1) @ V).transpose(0, 1).reshape(T, C)
    assert (y[n0 * T:(n0 + 1) * T].double() - want).abs().max().item() <= 2e-2
    np.testing.assert_allclose(lse[n0].double().cpu().numpy(), torch.logsumexp(S, -1).cpu().numpy(), atol=2e-2)


@pytest.mark.parametrize('N,T,heads,causal', [(8, 1024, 12, True), (2, 200, 3, True), (3, 129, 2, False)])
def test_attention_backward_from_bf16_rows(bf16_mode, N, T, heads, causal):
    """halo_attention_bwd_b16 against autograd through fp64 softmax attention of the same bf16 q / k / v / dy (sampled batch entry), and
    against the fp32-input matrix-core backward on those values."""
    from haloop_amd import ops
    hd, C = 64, heads * 64
    g = torch.Generator().manual_seed(T * 3 + heads)
    qkv = (torch.randn(N * T, 3 * C, generator=g) * 0.7).to(DEV).bfloat16()
    dyb = (torch.randn(N * T, C, generator=g) * 0.5).to(DEV).bfloat16()
    q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    _, lse, yb = ops.attention_fwd_b16(q, k, v, N, heads, hd, T, T, causal=causal, want_lse=True)
    d = torch.empty_like(qkv)
    ops.attention_bwd_b16(q, k, v, yb, dyb, lse, d[:, :C], d[:, C:2 * C], d[:, 2 * C:], N, heads, hd, T, T, causal=causal)
    # the fp32-input launches on the same values
    qf = qkv.float()
    y0, lse0, _ = ops.attention_fwd_bf16(qf[:, :C], qf[:, C:2 * C], qf[:, 2 * C:], N, heads, hd, T, T, causal=causal)
    d0 = torch.empty_like(qkv)
    ops.attention_bwd_bf16(qf[:, :C], qf[:, C:2 * C], qf[:, 2 * C:], y0, dyb.float(), lse0, d0[:, :C], d0[:, C:2 * C], d0[:, 2 * C:], N, heads, hd, T, T,
                           causal=causal)
    scale = d0.float().abs().max().item()
    assert (d.float() - d0.float()).abs().max().item() <= 3e-2 * scale
    # autograd in fp64 on one batch entry
    n0 = N - 1
    x = qkv[n0 * T:(n0 + 1) * T].double().clone().requires_grad_(True)
    Q, K, V = (x[:, i * C:(i + 1) * C].view(T, heads, hd).transpose(0, 1) for i in range(3))
    S = Q @ K.transpose(1, 2) / 8.0
    if causal:
        S = S.masked_fill(torch.ones(T, T, device=DEV).triu(1).bool(), float('-inf'))
    Y = (torch.softmax(S, -1) @ V).transpose(0, 1).reshape(T, C)
    Y.backward(dyb[n0 * T:(n0 + 1) * T].double())
    want = x.grad
    got = d[n0 * T:(n0 + 1) * T].double()
    cos = float((got * want).sum() / (got.norm() * want.norm()))
    assert cos >= 0.9995, cos
    assert (got - want).abs().max().item() <= 3e-2 * want.abs().max().item()


def test_layernorm_backward_from_bf16_gradient_rows(bf16_mode):
    """halo_layernorm_bwd_b16 = halo_layernorm_bwd_bf16 on the same (bf16-valued) incoming gradient, bit for bit."""
    from haloop_amd import ops
    g = torch.Generator().manual_seed(2)
    x = torch.randn(1000, 768, generator=g).to(DEV)
    dres = torch.randn(1000, 768, generator=g).to(DEV)
    w = (torch.rand(768, generator=g) + 0.5).to(DEV)
    dyb = torch.randn(1000, 768, generator=g).to(DEV).bfloat16()
    a = ops.layernorm_bwd(dyb, x, w, dres, has_bias=False, want_bf16=True)
    b = ops.layernorm_bwd(dyb.float(), x, w, dres, has_bias=False, want_bf16=True)
    for u, v in zip(a, b):
        assert (u is None and v is None) or torch.equal(u, v)
