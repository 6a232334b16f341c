"""The GPT block's `bf16` training path with row-major bf16 activations between the launches (haloop_amd/attention.py:
block_forward_train_rm / block_backward_rm; halo_gemm_split_io epilogues, halo_gemm_tn_bf16, halo_layernorm_bf16,
halo_layernorm_bwd_bf16) against the operand-image path it replaces on the same weights and batch: the same bf16 operand values and
fp32 accumulation, so loss and every gradient agree to summation order.  (Both are pinned to the reference by the fixtures of
tests/test_gpu_parity.py at the shapes the fixtures hold; this shape -- 8192 rows, where the row-major path is taken -- has no fixture.)"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _run(monkeypatch, rowmajor, seed=3):
    from haloop_amd import _lib, attention
    monkeypatch.setenv('HALO_GPT_ROWMAJOR', '1' if rowmajor else '0')
    monkeypatch.setenv('HALO_GPT_ROWS', '0')            # (the 128-tile launches this test was written for; tests/test_gpu_gemm_rows.py holds the row tiles)
    torch.manual_seed(seed)
    cfg = attention.GPTConfig(block_size=1024, vocab_size=2048, n_layer=2, n_head=12, n_embd=768)
    model = attention.GPT(cfg).to(DEV).train()
    with torch.no_grad():
        model.transformer.wpe.weight.normal_(0, 0.02)
    g = torch.Generator().manual_seed(seed)
    inputs = torch.randint(1, cfg.vocab_size, (8, 1024), generator=g).to(DEV)
    targets = torch.randint(1, cfg.vocab_size, (8, 1024), generator=g).to(DEV)
    taken = attention.rowmajor_train_ok(cfg, model.transformer.h, 8 * 1024, True)
    loss = model.forward_all(inputs, targets, reduction='mean')
    loss.backward()
    return taken, loss.item(), {k: p.grad.detach().clone() for k, p in model.named_parameters()}


def test_rowmajor_training_path_equals_the_image_path(monkeypatch):
    from haloop_amd import _lib
    _lib.lib(); _lib.lend_scratch(256 << 20)
    prev = _lib.get_math_mode()
    _lib.set_math_mode('bf16')
    try:
        on, loss_rm, g_rm = _run(monkeypatch, True)
        off, loss_img, g_img = _run(monkeypatch, False)
    finally:
        _lib.set_math_mode(prev)
    assert on and not off
    assert abs(loss_rm - loss_img) <= 1e-5 * abs(loss_img)
    for k in g_img:
        a, b = g_rm[k].float(), g_img[k].float()
        scale = float(b.norm()) / b.numel() ** 0.5
        assert float((a - b).abs().max()) <= 2e-3 * scale + 1e-9, k
        assert abs(float(a.norm()) - float(b.norm())) <= 1e-4 * float(b.norm()), k


def test_rowmajor_path_is_not_taken_where_it_does_not_apply(monkeypatch):
    from haloop_amd import _lib, attention
    monkeypatch.setenv('HALO_GPT_ROWMAJOR', '1')
    cfg = attention.GPTConfig(block_size=1024, vocab_size=2048, n_layer=1, n_head=12, n_embd=768)
    blocks = [attention.Block(cfg)]
    prev = _lib.get_math_mode()
    try:
        _lib.set_math_mode('bf16x3')
        assert not attention.rowmajor_train_ok(cfg, blocks, 8192, True)          # split-bf16 keeps the operand images
        _lib.set_math_mode('bf16')
        assert attention.rowmajor_train_ok(cfg, blocks, 8192, True)
        assert not attention.rowmajor_train_ok(cfg, blocks, 1024, True)          # too few tiles for products without split-K
        cfg_d = attention.GPTConfig(block_size=1024, vocab_size=2048, n_layer=1, n_head=12, n_embd=768, dropout=0.1)
        assert not attention.rowmajor_train_ok(cfg_d, blocks, 8192, True)        # output dropouts are epilogues of the image products
        cfg_b = attention.GPTConfig(block_size=1024, vocab_size=2048, n_layer=1, n_head=12, n_embd=768, bias=True)
        assert not attention.rowmajor_train_ok(cfg_b, [attention.Block(cfg_b)], 8192, True)
    finally:
        _lib.set_math_mode(prev)


def test_bf16_producers_match_the_fp32_operators():
    """The launches that write row-major bf16 (cast, the two GELU passes, LayerNorm forward and backward) against the fp32 operators."""
    from haloop_amd import _lib, ops
    _lib.lib(); _lib.lend_scratch(256 << 20)
    prev = _lib.get_math_mode()
    _lib.set_math_mode('bf16')
    try:
        g = torch.Generator().manual_seed(9)
        M, C = 4096, 512
        x = torch.randn(M, C, generator=g).to(DEV)
        w = (torch.randn(4 * C, C, generator=g) * C ** -0.5).to(DEV)
        xb = ops.cast_bf16(x)
        assert torch.equal(xb, x.bfloat16())
        a = ops.gemm_split_io((xb, None), ops.split_image(w), M, 4 * C, C)           # A staged from the row-major rows
        assert torch.equal(a, ops.gemm_split(ops.split_image(x), ops.split_image(w), M, 4 * C, C))
        dy = torch.randn(M, C, generator=g).to(DEV)
        dg = torch.randn(M, 4 * C, generator=g).to(DEV)
        for exact in (False, True):
            assert torch.equal(ops.gelu_bf16(a, exact), ops.gelu_fwd(a, exact).bfloat16())
            assert torch.equal(ops.gelu_bwd_bf16(dg, a, exact), ops.gelu_bwd(dg, a, exact).bfloat16())
        # LayerNorm as row-major bf16, and its backward's bf16 copy
        lw, lb = (1 + 0.1 * torch.randn(C, generator=g)).to(DEV), (0.1 * torch.randn(C, generator=g)).to(DEV)
        img_ref, y_ref = ops.layernorm_image(x, lw, lb, want_y=True)
        assert torch.equal(ops.layernorm_bf16(x, lw, lb), y_ref.bfloat16())
        rows, img = ops.layernorm_bf16(x, lw, lb, want_image=True)              # rows and tiled image from one launch
        wimg = ops.split_image(w)
        assert torch.equal(rows, y_ref.bfloat16())
        assert torch.equal(ops.gemm_split(img, wimg, M, 4 * C, C), ops.gemm_split(img_ref, wimg, M, 4 * C, C))
        dx, dw, db, dxb = ops.layernorm_bwd(dy, x, lw, dy, has_bias=True, want_bf16=True)
        dx2, dw2, db2 = ops.layernorm_bwd(dy, x, lw, dy, has_bias=True)
        assert torch.equal(dx, dx2) and torch.equal(dw, dw2) and torch.equal(db, db2) and torch.equal(dxb, dx.bfloat16())
    finally:
        _lib.set_math_mode(prev)


@pytest.mark.parametrize('mode', ['bf16', 'bf16x3'])
def test_attention_bf16_outputs_are_the_rounded_fp32_outputs(mode):
    from haloop_amd import _lib, ops
    prev = _lib.get_math_mode()
    _lib.set_math_mode(mode)
    try:
        N, heads, hd, T = 3, 4, 64, 320
        C = heads * hd
        g = torch.Generator().manual_seed(21)
        qkv = torch.randn(N * T, 3 * C, generator=g).to(DEV)
        dy = torch.randn(N * T, C, generator=g).to(DEV)
        q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
        y, lse, _ = ops.attention_fwd(q, k, v, N, heads, hd, T, T, causal=True, want_lse=True)
        y2, lse2, yb = ops.attention_fwd_bf16(q, k, v, N, heads, hd, T, T, causal=True)
        assert torch.equal(y, y2) and torch.equal(lse, lse2) and torch.equal(yb, y.bfloat16())
        dqkv = torch.empty_like(qkv)
        ops.attention_bwd(q, k, v, y, dy, lse, dqkv[:, :C], dqkv[:, C:2 * C], dqkv[:, 2 * C:], N, heads, hd, T, T, causal=True)
        dqkvb = torch.zeros(N * T, 3 * C, device=DEV, dtype=torch.bfloat16)
        ops.attention_bwd_bf16(q, k, v, y, dy, lse, dqkvb[:, :C], dqkvb[:, C:2 * C], dqkvb[:, 2 * C:], N, heads, hd, T, T, causal=True)
        assert torch.equal(dqkvb, dqkv.bfloat16())
    finally:
        _lib.set_math_mode(prev)
