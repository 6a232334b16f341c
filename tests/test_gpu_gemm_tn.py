"""halo_gemm_tn_bf16: C = A^T B from row-major bf16 operands whose row is the contraction index (the weight gradient of a Linear without
transposed operand images), against the fp64 product of the same bf16 values and against the image path on the same operands."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('K,M,N', [(64, 128, 128), (8192, 768, 3072), (8192, 3072, 768), (1024, 776, 2304), (96, 8, 40), (2048, 50304, 768)])
def test_tn_product_matches_fp64(K, M, N):
    from haloop_amd import _lib, ops
    _lib.lend_scratch(256 << 20)
    g = torch.Generator().manual_seed(K + M + N)
    a = torch.randn(K, M, generator=g).cuda().bfloat16()
    b = torch.randn(K, N, generator=g).cuda().bfloat16()
    got = ops.gemm_tn(a, b)
    rows = torch.randint(0, M, (64,), generator=g).cuda()              # a sample of output rows in fp64 (the full product is large)
    want = a.double()[:, rows].t() @ b.double()
    err = (got[rows].double() - want).abs().max().item()
    assert err <= 2e-6 * K ** 0.5 * 16 + 1e-5, err                      # fp32 accumulation of exact bf16 products
    # C += through the accumulate flag
    base = torch.randn(M, N, generator=g).cuda()
    acc = ops.gemm_tn(a, b, out=base.clone(), accumulate=True)
    torch.testing.assert_close(acc[rows], (base[rows].double() + want).float(), rtol=0, atol=err + 1e-4)


def test_tn_product_equals_the_image_path_in_bf16_mode():
    """The same bf16 operand values through the transposed operand images (the path the TN product replaces): equal up to summation order."""
    from haloop_amd import _lib, ops
    _lib.lend_scratch(256 << 20)
    mode = _lib.get_math_mode()
    _lib.set_math_mode('bf16')
    try:
        g = torch.Generator().manual_seed(5)
        K, M, N = 4096, 768, 768
        dy = torch.randn(K, M, generator=g).cuda().bfloat16()
        x = torch.randn(K, N, generator=g).cuda().bfloat16()
        got = ops.gemm_tn(dy, x)
        ref = ops.gemm_split(ops.split_image(dy.float(), transposed=True), ops.split_image(x.float(), transposed=True), M, N, K)
        torch.testing.assert_close(got, ref, rtol=0, atol=2e-3)
    finally:
        _lib.set_math_mode(mode)


def test_tn_refusals():
    from haloop_amd import _lib, ops
    a = torch.zeros(48, 128, device='cuda', dtype=torch.bfloat16)        # K % 32 != 0
    with pytest.raises(_lib.HaloError):
        ops.gemm_tn(a, a)
    a = torch.zeros(64, 12, device='cuda', dtype=torch.bfloat16)         # M % 8 != 0
    with pytest.raises(_lib.HaloError):
        ops.gemm_tn(a, a)


def test_two_slices_of_a_very_long_contraction():
    """256 .. 511 output tiles with K >= 32768 run as two K slices + a reduce (halo_pick_ksplit; the lm_head's input gradient): same result
    as the fp64 product of the same bf16 operand values."""
    from haloop_amd import _lib, ops
    _lib.lend_scratch(256 << 20)
    prev = _lib.get_math_mode()
    _lib.set_math_mode('bf16')
    try:
        g = torch.Generator().manual_seed(2)
        M, N, K = 4096, 1024, 32768                                   # 32 x 8 = 256 tiles, 1024 k-steps
        a = (torch.randn(M, K, generator=g) * 0.1).cuda()
        b = (torch.randn(N, K, generator=g) * 0.1).cuda()
        got = ops.gemm_split(ops.split_image(a), ops.split_image(b), M, N, K)
        rows = torch.randint(0, M, (32,), generator=g).cuda()
        want = a[rows].bfloat16().double() @ b.bfloat16().double().t()
        assert (got[rows].double() - want).abs().max().item() <= 5e-3
    finally:
        _lib.set_math_mode(prev)


@pytest.mark.parametrize('M,N,K', [(8192, 768, 768), (8192, 768, 3072), (8200, 640, 96), (4096, 1152, 1344)])
def test_half_height_tiles_equal_the_full_tiles(monkeypatch, M, N, K):
    """Single-pass products with 257 .. 511 full tiles run on 64 x 128 tiles (three workgroups per CU instead of one or two): the same
    k order per output element, so the same bits -- plain, with a residual addend, and with the A operand staged from row-major rows."""
    from haloop_amd import _lib, ops
    _lib.lend_scratch(256 << 20)
    prev = _lib.get_math_mode()
    _lib.set_math_mode('bf16')
    try:
        g = torch.Generator().manual_seed(M + N + K)
        a = torch.randn(M, K, generator=g).cuda()
        b = (torch.randn(N, K, generator=g) * K ** -0.5).cuda()
        r = torch.randn(M, N, generator=g).cuda()
        bias = torch.randn(N, generator=g).cuda()
        ai, bi, ab = ops.split_image(a), ops.split_image(b), ops.cast_bf16(a)
        rows_ok = K % 32 == 0
        out = {}
        for name, v in (('half', '1'), ('full', '0')):
            monkeypatch.setenv('HALO_GEMM_HALF_TILES', v)
            out[name] = (ops.gemm_split(ai, bi, M, N, K, bias1=bias), ops.gemm_split(ai, bi, M, N, K, residual=r),
                         ops.gemm_split_io((ab, None), bi, M, N, K, bias1=bias) if rows_ok else None,
                         ops.gemm_split_io((ab, None), bi, M, N, K, residual=r) if rows_ok else None)
        for x, y in zip(out['half'], out['full']):
            assert (x is None and y is None) or torch.equal(x, y)
        want = a[:64].bfloat16().double() @ b.bfloat16().double().t() + bias.double()
        assert (out['half'][0][:64].double() - want).abs().max().item() <= 2e-3
    finally:
        _lib.set_math_mode(prev)


def test_grouped_tn_products_equal_the_single_launches():
    """Four TN products over one contraction length in ONE launch on whole-K tiles (a GPT block's weight gradients) against the fp64 products
    of the same bf16 values and against the one-product launches (which cut K into slices: equal up to the summation order)."""
    from haloop_amd import _lib, ops
    _lib.lend_scratch(256 << 20)
    prev = _lib.get_math_mode()
    _lib.set_math_mode('bf16')
    try:
        g = torch.Generator().manual_seed(11)
        K = 8192
        shapes = [(768, 3072), (3072, 768), (768, 768), (2304, 768), (40, 8)]           # five: a launch of four and a launch of one
        pairs = [(torch.randn(K, M, generator=g).cuda().bfloat16(), torch.randn(K, N, generator=g).cuda().bfloat16()) for M, N in shapes]
        outs = ops.gemm_tn_group(pairs)
        assert [tuple(o.shape) for o in outs] == shapes
        for (a, b), o in zip(pairs, outs):
            rows = torch.randint(0, a.shape[1], (32,), generator=g).cuda()
            want = a.double()[:, rows].t() @ b.double()
            assert (o[rows].double() - want).abs().max().item() <= 2e-6 * K ** 0.5 * 16 + 1e-5
            torch.testing.assert_close(o, ops.gemm_tn(a, b), rtol=0, atol=2e-3)
    finally:
        _lib.set_math_mode(prev)


# ---- the 256-row tiles (csrc/gemm_tn_rows.hip): what ops.gemm_tn_group runs in single-pass bf16 arithmetic ----
@pytest.fixture
def bf16_mode():
    from haloop_amd import _lib
    _lib.lib(); _lib.lend_scratch(256 << 20)
    prev = _lib.get_math_mode()
    _lib.set_math_mode('bf16')
    yield
    _lib.set_math_mode(prev)


def _tn_rows_case(K, shapes, seed):
    g = torch.Generator().manual_seed(seed)
    pairs = [(torch.randn(K, m, generator=g).cuda().bfloat16(), torch.randn(K, n, generator=g).cuda().bfloat16()) for m, n in shapes]
    return pairs, g


# a GPT block's four weight gradients (216 tiles of 256 x 128), the lm_head's (256 x 256 tiles), ragged rows / columns that do not fill
# their last tile, one k-block, 8-wide operands, both tile widths forced
@pytest.mark.parametrize('K,shapes,tn', [(8192, [(2304, 768), (768, 768), (3072, 768), (768, 3072)], 0), (2048, [(50304, 768)], 0),
                                         (2048, [(50304, 768)], 4), (1024, [(776, 2304), (40, 8), (264, 136)], 4),
                                         (1024, [(776, 2304), (40, 8), (264, 136)], 8), (32, [(8, 8)], 0), (96, [(520, 264), (256, 128)], 8),
                                         (160, [(1000, 200)], 4)])
def test_tn_products_on_256_row_tiles(bf16_mode, monkeypatch, K, shapes, tn):
    from haloop_amd import ops
    if tn:
        monkeypatch.setenv('HALO_GEMM_TN_ROWS_TN', str(tn))
    monkeypatch.setenv('HALO_GEMM_TN_ROWS', '1')
    pairs, g = _tn_rows_case(K, shapes, K + len(shapes) + tn)
    got = ops.gemm_tn_group(pairs)
    monkeypatch.setenv('HALO_GEMM_TN_ROWS', '0')
    old = ops.gemm_tn_group(pairs)                                      # the 128 x 128 tiles: the same products, another summation order
    tol = 2e-6 * K ** 0.5 * 16 + 1e-5
    for (a, b), c, c_old in zip(pairs, got, old):
        M = a.shape[1]
        rows = torch.cat([torch.arange(0, min(M, 24)), torch.arange(max(0, M - 24), M), torch.randint(0, M, (48,), generator=g)]).cuda()
        want = a.double()[:, rows].t() @ b.double()
        assert (c[rows].double() - want).abs().max().item() <= tol
        assert (c - c_old).abs().max().item() <= 2 * tol


def test_tn_rows_tail_as_k_slices(bf16_mode, monkeypatch):
    """A group whose tiles do not fill their last round of the CUs (591 tiles of 256 x 256 on 256 CUs: two rounds and 79) runs the last
    tiles as K-slices into scratch slabs + a sum launch: the same products as whole-K tiles up to the summation order; without scratch the
    launch falls back to whole tiles."""
    from haloop_amd import _lib, ops
    monkeypatch.setenv('HALO_GEMM_TN_ROWS', '1')
    K = 2048
    pairs, g = _tn_rows_case(K, [(50304, 768)], 77)
    got = ops.gemm_tn_group(pairs)[0]
    monkeypatch.setenv('HALO_GEMM_TN_ROWS_TAIL', '0')
    whole = ops.gemm_tn_group(pairs)[0]
    tol = 2e-6 * K ** 0.5 * 16 + 1e-5
    assert not torch.equal(got, whole)                                  # (the tail really ran in slices: another summation order)
    assert (got - whole).abs().max().item() <= 2 * tol
    a, b = pairs[0]
    rows = torch.cat([torch.arange(50304 - 300, 50304), torch.randint(0, 50304, (64,), generator=g)]).cuda()       # the tail tiles' rows
    want = a.double()[:, rows].t() @ b.double()
    assert (got[rows].double() - want).abs().max().item() <= tol
    monkeypatch.delenv('HALO_GEMM_TN_ROWS_TAIL')
    _lib.lib().halo_set_scratch(None, 0)
    try:
        assert torch.equal(ops.gemm_tn_group(pairs)[0], whole)         # no scratch: whole tiles
    finally:
        _lib.lib().halo_set_scratch(_lib._scratch.data_ptr(), _lib._scratch.numel())


def test_tn_rows_refusals(bf16_mode):
    import ctypes as C
    from haloop_amd import _lib
    L = _lib.lib()
    one = lambda v: (C.c_int * 1)(v)
    assert L.halo_gemm_tn_rows_supported(1, one(64), one(64), 64) == 1
    assert L.halo_gemm_tn_rows_supported(1, one(60), one(64), 64) == 0      # M % 8
    assert L.halo_gemm_tn_rows_supported(1, one(64), one(64), 48) == 0      # K % 32
    assert L.halo_gemm_tn_rows_supported(5, one(64), one(64), 64) == 0      # more than four products
    assert L.halo_gemm_tn_rows_preferred(1, one(50304), one(768), 8192) == 1 and L.halo_gemm_tn_rows_preferred(1, one(3072), one(768), 8192) == 0
    _lib.set_math_mode('bf16x3')
    assert L.halo_gemm_tn_rows_supported(1, one(64), one(64), 64) == 0      # single-pass bf16 arithmetic only


def test_gpt_step_with_the_256_row_weight_gradients(bf16_mode, monkeypatch):
    """The GPT training step's gradients with the weight gradients on the 256-row tiles against the 128 x 128 tiles (same operands, another
    summation order): every gradient within fp32 accumulation noise."""
    from haloop_amd import attention
    torch.manual_seed(3)
    cfg = attention.GPTConfig(block_size=256, vocab_size=1024, n_layer=2, n_head=4, n_embd=256)
    model = attention.GPT(cfg).cuda().train()
    ids = torch.randint(1, 1024, (4, 256), device='cuda')
    tg = torch.roll(ids, -1, 1)

    def grads():
        for p in model.parameters():
            p.grad = None
        model.forward_all(ids, tg).backward()
        return [p.grad.clone() for p in model.parameters()]
    monkeypatch.setenv('HALO_GEMM_TN_ROWS', '1')
    new = grads()
    monkeypatch.setenv('HALO_GEMM_TN_ROWS', '0')
    old = grads()
    for a, b in zip(new, old):
        assert (a - b).abs().max().item() <= 1e-5 + 1e-4 * b.abs().max().item()
