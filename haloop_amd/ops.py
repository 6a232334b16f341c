"""Tensor-level wrappers over the C ABI (no autograd here).  Every function enqueues on torch's
current HIP stream and allocates its outputs/workspaces with torch (device memory plumbing only).
"""
import ctypes as C

import torch

from . import _lib
from ._lib import check, lib, ptr, ptr_array


def _stream():
    # the raw hipStream_t of torch's current stream: torch.cuda.current_stream() builds a Stream object per call (~4 us of
    # host time, half of what a small launch costs end to end); this is the accessor torch's own kernels launchers use
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())


def _f32c(t, name):
    if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
        raise ValueError(f'{name}: expected a contiguous float32 HIP tensor, got {t.dtype} {t.device} '
                         f'contiguous={t.is_contiguous()}')
    return t


class Dropout:
    """Value bundle (p, seed, offset, device step counter) for one forward/backward pair."""
    __slots__ = ('p', 'seed', 'offset', 'counter')

    def __init__(self, p=0.0, seed=0, offset=0, counter=None):
        self.p, self.seed, self.offset, self.counter = float(p), int(seed), int(offset), counter

    @property
    def counter_ptr(self):
        return ptr(self.counter)


NO_DROPOUT = Dropout()


def subsampled_length(T, ks=5, stride=4, pad=3):
    return (T + 2 * pad - ks) // stride + 1


def _gemm_flags(relu, gelu, accumulate):
    """gelu: False, True / 'tanh' (new_gelu) or 'erf' (nn.GELU())."""
    g = 0 if not gelu else (_lib.HALO_GEMM_GELU_ERF if gelu == 'erf' else _lib.HALO_GEMM_GELU)
    return (_lib.HALO_GEMM_RELU if relu else 0) | g | (_lib.HALO_GEMM_ACCUM if accumulate else 0)


def gemm(a, b, a_kcontig, b_kcontig, M, N, K, out=None, bias1=None, bias2=None, relu=False,
         drop=NO_DROPOUT, stream_id=0, gelu=False, accumulate=False):
    lda = a.stride(0) if a.dim() == 2 else a.shape[-1]          # row-strided views (column slices of packed buffers) are fine
    ldb = b.stride(0) if b.dim() == 2 else b.shape[-1]
    if out is None:
        out = torch.empty(M, N, device=a.device, dtype=torch.float32)
    check(lib().halo_gemm_f32(int(a_kcontig), int(b_kcontig), M, N, K, ptr(a), lda, ptr(b), ldb, ptr(out), N,
                              ptr(bias1), ptr(bias2), _gemm_flags(relu, gelu, accumulate), drop.p, drop.seed,
                              stream_id, drop.offset, drop.counter_ptr, _stream()), 'halo_gemm_f32')
    return out


def split_image(x2d, transposed=False):
    """bf16 hi/lo tiled image of logical X[rows][k]; x2d is [rows,k], or [k,rows] when ``transposed``."""
    if x2d.dtype != torch.float32 or not x2d.is_cuda or x2d.dim() != 2 or x2d.stride(1) != 1:
        raise ValueError('split_image: expected a 2-D float32 HIP tensor with a unit column stride')
    rows, k = (x2d.shape[1], x2d.shape[0]) if transposed else x2d.shape
    img = torch.empty(lib().halo_split_image_bytes(rows, k), device=x2d.device, dtype=torch.uint8)
    check(lib().halo_split_image(ptr(x2d), rows, k, x2d.stride(0), int(transposed), ptr(img), _stream()), 'halo_split_image')
    return img


PAIR_COPY, PAIR_GELU, PAIR_GELU_ERF, PAIR_GELU_BWD, PAIR_GELU_ERF_BWD = range(5)


def image_pair(x2d, op=PAIR_COPY, x2=None, rows_image=True, cols_image=True):
    """One read of x2d [rows, cols] (through ``op``) -> (split image of value [rows][cols], split image of value^T [cols][rows]):
    the two operand images a Linear's backward wants of the same matrix (dx = dy W and dW = dy^T x); either can be skipped (None).
    ops: PAIR_COPY value = x2d; PAIR_GELU(_ERF) value = gelu(x2d); PAIR_GELU(_ERF)_BWD value = x2d * gelu'(x2) (x2d = dy)."""
    for t in (x2d, x2):
        if t is not None and (t.dtype != torch.float32 or not t.is_cuda or t.dim() != 2 or t.stride(1) != 1):
            raise ValueError('image_pair: expected 2-D float32 HIP tensors with a unit column stride')
    if op in (PAIR_GELU_BWD, PAIR_GELU_ERF_BWD) and (x2 is None or x2.shape != x2d.shape):
        raise ValueError('image_pair: the GELU backward needs the pre-activation, shaped like dy')
    rows, cols = x2d.shape
    rm = torch.empty(lib().halo_split_image_bytes(rows, cols), device=x2d.device, dtype=torch.uint8) if rows_image else None
    tr = torch.empty(lib().halo_split_image_bytes(cols, rows), device=x2d.device, dtype=torch.uint8) if cols_image else None
    check(lib().halo_image_pair(ptr(x2d), ptr(x2), rows, cols, x2d.stride(0), x2.stride(0) if x2 is not None else 0, op, ptr(rm), ptr(tr),
                                _stream()), 'halo_image_pair')
    return rm, tr


def image_pairs(mats):
    """image_pair(PAIR_COPY) of several 2-D fp32 matrices in ceil(n / 6) launches -> [(image, image of the transpose), ...]."""
    import ctypes as C
    mats = list(mats)
    if not mats:
        return []
    for t in mats:
        if t.dtype != torch.float32 or not t.is_cuda or t.dim() != 2 or t.stride(1) != 1:
            raise ValueError('image_pairs: expected 2-D float32 HIP tensors with a unit column stride')
    n = len(mats)
    dev = mats[0].device
    rm = [torch.empty(lib().halo_split_image_bytes(t.shape[0], t.shape[1]), device=dev, dtype=torch.uint8) for t in mats]
    tr = [torch.empty(lib().halo_split_image_bytes(t.shape[1], t.shape[0]), device=dev, dtype=torch.uint8) for t in mats]
    vp = lambda xs: (C.c_void_p * n)(*[ptr(x) for x in xs])
    check(lib().halo_image_pairs(n, vp(mats), (C.c_int * n)(*[t.shape[0] for t in mats]), (C.c_int * n)(*[t.shape[1] for t in mats]),
                                 (C.c_long * n)(*[t.stride(0) for t in mats]), vp(rm), vp(tr), _stream()), 'halo_image_pairs')
    return list(zip(rm, tr))


def layernorm_image(x2d, weight, bias=None, eps=1e-5, want_y=False):
    """LayerNorm of the rows written directly as the split operand image of the Linear that follows (C % 32 == 0).
    -> (image, y fp32 or None)"""
    _f32c(x2d, 'x')
    rows, Cn = x2d.shape
    img = torch.empty(lib().halo_split_image_bytes(rows, Cn), device=x2d.device, dtype=torch.uint8)
    y = torch.empty_like(x2d) if want_y else None
    check(lib().halo_layernorm_image(ptr(x2d), ptr(weight), ptr(bias), ptr(y), ptr(img), rows, Cn, eps, _stream()), 'halo_layernorm_image')
    return img, y


def layernorm_bf16(x2d, weight, bias=None, eps=1e-5, want_image=False):
    """LayerNorm of the rows as row-major bf16 [rows, C]: the operand form gemm_split_io (a = (hi, None)) and gemm_tn read.
    ``want_image``: -> (rows, tiled operand image of the same values) from the one launch."""
    _f32c(x2d, 'x')
    rows, Cn = x2d.shape
    y = torch.empty(rows, Cn, device=x2d.device, dtype=torch.bfloat16)
    img = torch.empty(lib().halo_split_image_bytes(rows, Cn), device=x2d.device, dtype=torch.uint8) if want_image else None
    check(lib().halo_layernorm_bf16(ptr(x2d), ptr(weight), ptr(bias), None, ptr(y), ptr(img), rows, Cn, eps, _stream()), 'halo_layernorm_bf16')
    return (y, img) if want_image else y


def gelu_bf16(a, exact=False):
    """gelu(a) as bf16 (same shape)."""
    _f32c(a, 'a')
    y = torch.empty(a.shape, device=a.device, dtype=torch.bfloat16)
    check(lib().halo_gelu_bf16(ptr(a), ptr(y), a.numel(), int(exact), _stream()), 'halo_gelu_bf16')
    return y


def gelu_bwd_bf16(dy, a, exact=False):
    """dy * gelu'(a) as bf16."""
    _f32c(dy, 'dy'); _f32c(a, 'a')
    y = torch.empty(a.shape, device=a.device, dtype=torch.bfloat16)
    check(lib().halo_gelu_bwd_bf16(ptr(dy), ptr(a), ptr(y), a.numel(), int(exact), _stream()), 'halo_gelu_bwd_bf16')
    return y


def cast_bf16(x):
    _f32c(x, 'x')
    y = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16)
    check(lib().halo_cast_bf16(ptr(x), ptr(y), x.numel(), _stream()), 'halo_cast_bf16')
    return y


def gemm_split(a_img, b_img, M, N, K, out=None, bias1=None, bias2=None, relu=False, drop=NO_DROPOUT, stream_id=0,
               gelu=False, accumulate=False, residual=None):
    """C[M,N] = A[M,K] B[N,K]^T from split images (three bf16 MFMAs per product, fp32 accumulate).  ``residual`` [M, N]: the
    result is added to it and written to ``out`` (a fresh tensor by default); ``accumulate`` adds into ``out`` in place."""
    if out is None:
        out = torch.empty(M, N, device=a_img.device, dtype=torch.float32)
    if residual is not None:
        if accumulate:
            raise ValueError('gemm_split: residual= and accumulate= are alternatives')
        _f32c(residual, 'residual')
        if residual.shape != (M, N):
            raise ValueError('gemm_split: residual must be [M, N]')
        check(lib().halo_gemm_split_residual(ptr(a_img), ptr(b_img), M, N, K, ptr(out), N, ptr(residual), N, ptr(bias1), ptr(bias2),
                                             _gemm_flags(relu, gelu, False), drop.p, drop.seed, stream_id, drop.offset,
                                             drop.counter_ptr, _stream()), 'halo_gemm_split_residual')
        return out
    check(lib().halo_gemm_split(ptr(a_img), ptr(b_img), M, N, K, ptr(out), N, ptr(bias1), ptr(bias2),
                                _gemm_flags(relu, gelu, accumulate), drop.p, drop.seed, stream_id, drop.offset,
                                drop.counter_ptr, _stream()), 'halo_gemm_split')
    return out


def gemm_split_io(a, b_img, M, N, K, out=None, out_rowmajor=False, bias1=None, bias2=None, relu=False, gelu=False, accumulate=False,
                  residual=None):
    """gemm_split with row-major bf16 activations on either side (halo_gemm_split_io).  ``a``: an operand image (uint8 tensor) or a
    (hi, lo) pair of row-major bf16 [M, K] tensors (lo None in bf16 mode).  ``out_rowmajor``: return the result as a (hi, lo) pair of
    bf16 [M, N] tensors (lo None in bf16 mode) and write no fp32 unless ``out`` is given.  -> out, or (hi, lo)."""
    x3 = lib().halo_get_math_mode() != _lib.HALO_MATH_BF16
    a_img = a_hi = a_lo = None
    lda = 0
    if isinstance(a, (tuple, list)):
        a_hi, a_lo = a
        if a_hi.dtype != torch.bfloat16 or a_hi.shape != (M, K) or a_hi.stride(1) != 1 or (x3 and (a_lo is None or a_lo.shape != (M, K) or a_lo.stride() != a_hi.stride())):
            raise ValueError('gemm_split_io: a = (hi, lo) row-major bf16 [M, K] tensors with equal strides')
        lda = a_hi.stride(0)
    else:
        a_img = a
    dev = b_img.device
    o_hi = o_lo = None
    if out_rowmajor:
        o_hi = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        o_lo = torch.empty(M, N, device=dev, dtype=torch.bfloat16) if x3 else None
    elif out is None:
        out = torch.empty(M, N, device=dev, dtype=torch.float32)
    if residual is not None:
        _f32c(residual, 'residual')
    flags = _gemm_flags(relu, gelu, accumulate or residual is not None)
    check(lib().halo_gemm_split_io(ptr(a_img), ptr(a_hi), ptr(a_lo), lda, ptr(b_img), M, N, K, ptr(out), N if out is not None else 0,
                                   ptr(o_hi), ptr(o_lo), N, ptr(residual), N if residual is not None else 0, ptr(bias1), ptr(bias2),
                                   flags, _stream()), 'halo_gemm_split_io')
    return (o_hi, o_lo) if out_rowmajor else out


def gemm_tn(a, b, out=None, accumulate=False):
    """a [K, M]^T b [K, N] -> fp32 [M, N] for row-major bf16 a, b (the contraction index is their row): halo_gemm_tn_bf16, the weight
    gradient of a Linear from row-major bf16 activations / output gradients without a transposed operand image."""
    if a.dtype != torch.bfloat16 or b.dtype != torch.bfloat16 or a.dim() != 2 or b.dim() != 2 or a.shape[0] != b.shape[0]:
        raise ValueError('gemm_tn: a [K, M], b [K, N] bf16')
    if a.stride(1) != 1 or b.stride(1) != 1:
        raise ValueError('gemm_tn: row-major operands')
    K, M = a.shape
    N = b.shape[1]
    if out is None:
        out = torch.empty(M, N, device=a.device, dtype=torch.float32)
        accumulate = False
    check(lib().halo_gemm_tn_bf16(ptr(a), a.stride(0), ptr(b), b.stride(0), M, N, K, ptr(out), out.stride(0),
                                  _lib.HALO_GEMM_ACCUM if accumulate else 0, _stream()), 'halo_gemm_tn_bf16')
    return out


def gemm_tn_group(pairs):
    """[a_i [K, M_i]^T b_i [K, N_i] for (a_i, b_i) in pairs] -> list of fp32 [M_i, N_i], row-major bf16 operands with ONE contraction length
    K: up to four products per launch on whole-K tiles -- a GPT block's four weight gradients, the lm_head's.  In single-pass bf16
    arithmetic a group of at least two rounds of 256 x 256 tiles (the lm_head) runs on halo_gemm_tn_rows_group's 256-row tiles
    (csrc/gemm_tn_rows.hip), everything else on the 128 x 128 tiles of halo_gemm_tn_bf16_group; HALO_GEMM_TN_ROWS=1 / 0 force one."""
    import ctypes as C
    import os
    outs = []
    for i0 in range(0, len(pairs), 4):
        grp = pairs[i0:i0 + 4]
        n = len(grp)
        K = grp[0][0].shape[0]
        for a, b in grp:
            if a.dtype != torch.bfloat16 or b.dtype != torch.bfloat16 or a.dim() != 2 or b.dim() != 2 or a.shape[0] != K or b.shape[0] != K \
                    or a.stride(1) != 1 or b.stride(1) != 1:
                raise ValueError('gemm_tn_group: row-major bf16 a [K, M], b [K, N] with one K')
        cs = [torch.empty(a.shape[1], b.shape[1], device=a.device, dtype=torch.float32) for a, b in grp]
        vp = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
        Ms, Ns = (C.c_int * n)(*[a.shape[1] for a, _ in grp]), (C.c_int * n)(*[b.shape[1] for _, b in grp])
        force = os.environ.get('HALO_GEMM_TN_ROWS', '')
        if force != '0' and (lib().halo_gemm_tn_rows_supported if force == '1' else lib().halo_gemm_tn_rows_preferred)(n, Ms, Ns, K) \
                and all(a.stride(0) % 8 == 0 and b.stride(0) % 8 == 0 and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0 for a, b in grp):
            check(lib().halo_gemm_tn_rows_group(n, vp([a for a, _ in grp]), (C.c_long * n)(*[a.stride(0) for a, _ in grp]),
                                                vp([b for _, b in grp]), (C.c_long * n)(*[b.stride(0) for _, b in grp]), Ms, Ns, K,
                                                vp(cs), (C.c_long * n)(*[c.stride(0) for c in cs]), _stream()), 'halo_gemm_tn_rows_group')
            outs += cs
            continue
        check(lib().halo_gemm_tn_bf16_group(n, vp([a for a, _ in grp]), (C.c_long * n)(*[a.stride(0) for a, _ in grp]),
                                            vp([b for _, b in grp]), (C.c_long * n)(*[b.stride(0) for _, b in grp]),
                                            (C.c_int * n)(*[a.shape[1] for a, _ in grp]), (C.c_int * n)(*[b.shape[1] for _, b in grp]), K,
                                            vp(cs), (C.c_int * n)(*[c.stride(0) for c in cs]), 0, _stream()), 'halo_gemm_tn_bf16_group')
        outs += cs
    return outs


def gemm_split_ce(a_img, b_img, M, N, K, targets, ignore_index=0, bias=None, want_logits=False, want_lse=False):
    """Per-row cross-entropy of logits = A B^T (+ bias) straight from the split GEMM's epilogue: -> (loss [M], lse [M] or None,
    logits [M, N] or None).  Without want_logits nothing of size M x N is written."""
    tg = _i64c(targets.reshape(-1), 'targets')
    dev = a_img.device
    ws = torch.empty(lib().halo_gemm_split_ce_workspace_bytes(M, N), device=dev, dtype=torch.uint8)
    loss = torch.empty(M, device=dev, dtype=torch.float32)
    lse = torch.empty(M, device=dev, dtype=torch.float32) if want_lse else None
    logits = torch.empty(M, N, device=dev, dtype=torch.float32) if want_logits else None
    check(lib().halo_gemm_split_ce(ptr(a_img), ptr(b_img), M, N, K, ptr(logits), N, ptr(bias), ptr(tg), ignore_index, ptr(ws), ptr(loss),
                                   ptr(lse), _stream()), 'halo_gemm_split_ce')
    return loss, lse, logits


def gemm_rows_supported(M, N, K):
    """halo_gemm_rows takes this shape in the current arithmetic mode (single-pass bf16 only)."""
    return bool(lib().halo_gemm_rows_supported(M, N, K))


def gemm_rows(a, b_img, M, N, K, out=None, residual=None, out_bf16=False):
    """C [M, N] = A [M, K] B [N, K]^T on the 256-row tiles of halo_gemm_rows (single-pass bf16).  ``a``: row-major bf16 [M, K] or an operand
    image (uint8).  Result: fp32 ``out`` (fresh by default), ``residual`` [M, N] fp32 added when given; or, with ``out_bf16``, row-major
    bf16 [M, N]."""
    a_img = a_rm = None
    lda = 0
    if a.dtype == torch.bfloat16:
        if a.dim() != 2 or a.shape != (M, K) or a.stride(1) != 1:
            raise ValueError('gemm_rows: a = row-major bf16 [M, K]')
        a_rm, lda = a, a.stride(0)
    else:
        a_img = a
    dev = b_img.device
    ob = None
    if out_bf16:
        if residual is not None or out is not None:
            raise ValueError('gemm_rows: a bf16 result takes neither out= nor residual=')
        ob = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    elif out is None:
        out = torch.empty(M, N, device=dev, dtype=torch.float32)
    if residual is not None:
        _f32c(residual, 'residual')
        if residual.shape != (M, N):
            raise ValueError('gemm_rows: residual must be [M, N]')
    check(lib().halo_gemm_rows(ptr(a_img), ptr(a_rm), lda, ptr(b_img), M, N, K, ptr(out), N if out is not None else 0, ptr(residual),
                               N if residual is not None else 0, ptr(ob), N if ob is not None else 0, _stream()), 'halo_gemm_rows')
    return ob if out_bf16 else out


def gemm_rows_gelu(a, b_img, M, N, K, exact=False, keep_pre=False):
    """gelu(A B^T) as row-major bf16 [M, N] from halo_gemm_rows' epilogue; ``keep_pre``: -> (gelu, pre-activation), both bf16 -- the same
    bits as gemm_rows(out_bf16=True) followed by gelu_b16."""
    a_img = a_rm = None
    lda = 0
    if a.dtype == torch.bfloat16:
        a_rm, lda = a, a.stride(0)
    else:
        a_img = a
    g = torch.empty(M, N, device=b_img.device, dtype=torch.bfloat16)
    pre = torch.empty(M, N, device=b_img.device, dtype=torch.bfloat16) if keep_pre else None
    check(lib().halo_gemm_rows_gelu(ptr(a_img), ptr(a_rm), lda, ptr(b_img), M, N, K, ptr(g), ptr(pre), N, int(exact), _stream()), 'halo_gemm_rows_gelu')
    return (g, pre) if keep_pre else g


def gemm_rows_ce(a, b_img, M, N, K, targets, ignore_index=0, want_logits=False, want_lse=False):
    """Per-row cross-entropy of logits = A B^T from halo_gemm_rows_ce's epilogue: -> (loss [M], lse [M] or None, logits [M, N] as
    row-major bf16 or None).  ``a``: row-major bf16 [M, K] or an operand image."""
    tg = _i64c(targets.reshape(-1), 'targets')
    a_img = a_rm = None
    lda = 0
    if a.dtype == torch.bfloat16:
        a_rm, lda = a, a.stride(0)
    else:
        a_img = a
    dev = b_img.device
    ws = torch.empty(lib().halo_gemm_rows_ce_workspace_bytes(M, N), device=dev, dtype=torch.uint8)
    loss = torch.empty(M, device=dev, dtype=torch.float32)
    lse = torch.empty(M, device=dev, dtype=torch.float32) if want_lse else None
    logits = torch.empty(M, N, device=dev, dtype=torch.bfloat16) if want_logits else None
    check(lib().halo_gemm_rows_ce(ptr(a_img), ptr(a_rm), lda, ptr(b_img), M, N, K, ptr(tg), ignore_index, ptr(ws), ptr(loss), ptr(lse),
                                  ptr(logits), N if logits is not None else 0, _stream()), 'halo_gemm_rows_ce')
    return loss, lse, logits


def cross_entropy_bwd_bf16_(logits_bf16, targets, lse, grad_rows, ignore_index=0):
    """In place: bf16 logits [rows, V] -> (softmax - onehot) * grad_rows[row] as row-major bf16 (0 on ignored rows)."""
    if logits_bf16.dtype != torch.bfloat16 or logits_bf16.dim() != 2 or logits_bf16.stride(1) != 1:
        raise ValueError('cross_entropy_bwd_bf16_: row-major bf16 logits')
    tg = _i64c(targets.reshape(-1), 'targets')
    rows, V = logits_bf16.shape
    g = grad_rows.reshape(-1)
    _f32c(g, 'grad_rows'); _f32c(lse, 'lse')
    stride = 0 if g.numel() == 1 else 1
    check(lib().halo_cross_entropy_bwd_bf16(ptr(logits_bf16), ptr(tg), ptr(lse), ptr(g), stride, rows, V, logits_bf16.stride(0), ignore_index,
                                            _stream()), 'halo_cross_entropy_bwd_bf16')
    return logits_bf16


def gelu_b16(a, exact=False):
    """gelu(a), bf16 -> bf16."""
    y = torch.empty_like(a)
    check(lib().halo_gelu_b16(ptr(a), ptr(y), a.numel(), int(exact), _stream()), 'halo_gelu_b16')
    return y


def gelu_bwd_b16(dy, a, exact=False):
    """dy * gelu'(a), bf16 x bf16 -> bf16."""
    y = torch.empty_like(a)
    check(lib().halo_gelu_bwd_b16(ptr(dy), ptr(a), ptr(y), a.numel(), int(exact), _stream()), 'halo_gelu_bwd_b16')
    return y


def dropout_fwd(x, drop, stream_id):
    _f32c(x, 'x')
    y = torch.empty_like(x)
    check(lib().halo_dropout_fwd(ptr(x), ptr(y), x.numel(), drop.p, drop.seed, stream_id, drop.offset,
                                 drop.counter_ptr, _stream()), 'halo_dropout_fwd')
    return y


def subsample_fwd(x, w, bias, drop, ks=5, stride=4, pad=3):
    """x [B,T,F] -> y [T',B,C] time-major, col (saved im2col image)."""
    _f32c(x, 'inputs'); _f32c(w, 'subsample.weight'); _f32c(bias, 'subsample.bias')
    B, T, F = x.shape
    Cc = w.shape[0]
    Tp = subsampled_length(T, ks, stride, pad)
    y = torch.empty(Tp, B, Cc, device=x.device, dtype=torch.float32)
    col = torch.empty(Tp * B, F * ks, device=x.device, dtype=torch.float32)
    check(lib().halo_subsample_fwd(ptr(x), ptr(w), ptr(bias), ptr(y), ptr(col), B, T, F, Cc, ks, stride, pad,
                                   drop.p, drop.seed, drop.offset, drop.counter_ptr, _stream()), 'halo_subsample_fwd')
    return y, col


def subsample_bwd(dy, y, col, B, T, F, Cc, p_drop, dw=None, dbias=None, ks=5, stride=4, pad=3, slabs=1):
    """``slabs`` > 1: ``dy`` is [slabs, T', B, C], the gradient left as that many K-slices by ``lstm_bwd(..., dx_slabs=)``; they are added
    while they are read."""
    _f32c(dy, 'dy')
    if slabs > 1 and dy.numel() < slabs * y.numel():
        raise ValueError('haloop_amd.ops.subsample_bwd: dy holds fewer than `slabs` matrices')
    dpre = torch.empty_like(y)
    if dw is None:
        dw = torch.empty(Cc, F, ks, device=y.device, dtype=torch.float32)
    if dbias is None:
        dbias = torch.empty(Cc, device=y.device, dtype=torch.float32)
    check(lib().halo_subsample_bwd_slabs(ptr(dy), int(slabs), ptr(y), ptr(col), ptr(dpre), ptr(dw), ptr(dbias), B, T, F, Cc, ks, stride,
                                         pad, p_drop, _stream()), 'halo_subsample_bwd_slabs')
    return dw, dbias


def lstm_fwd(x_tm, w_ih, w_hh, b_ih, b_hh, h0=None, c0=None, y=None, y_strides=None, y_relu=False,
             want_state=False, drop=NO_DROPOUT, expect_backward=True, reserve=None, weights_stamp=0):
    """x_tm [T,B,in] time-major.  Returns (y, hn, cn, reserve).
    expect_backward=False (inference): the two-layer launch packs only the forward's weight images (include/halo.h); with a
    caller-owned ``reserve`` (ops.lstm_reserve) and a non-zero ``weights_stamp`` that changes whenever the weights do, it keeps the
    images the previous call left there (halo_set_lstm_weights_stamp).

    y defaults to a time-major [T,B,H] tensor; pass a preallocated ``y`` with ``y_strides`` =
    (stride_t, stride_b) in elements to have the last layer write e.g. batch-first."""
    _f32c(x_tm, 'x')
    T, B, in0 = x_tm.shape
    L = len(w_hh)
    H = w_hh[0].shape[1]
    for t in list(w_ih) + list(w_hh) + list(b_ih) + list(b_hh):
        _f32c(t, 'lstm parameter')
    dev = x_tm.device
    if y is None:
        y = torch.empty(T, B, H, device=dev, dtype=torch.float32)
        y_strides = (B * H, H)
    need = (lib().halo_lstm_reserve_bytes(T, B, in0, H, L) + 3) // 4
    if reserve is None:
        reserve = torch.empty(need, device=dev, dtype=torch.float32)
    elif reserve.dtype != torch.float32 or reserve.numel() < need or not reserve.is_contiguous():
        raise ValueError('lstm_fwd: the reserve is too small for this shape (ops.lstm_reserve)')
    hn = torch.empty(L, B, H, device=dev, dtype=torch.float32) if want_state else None
    cn = torch.empty(L, B, H, device=dev, dtype=torch.float32) if want_state else None
    if h0 is not None:
        _f32c(h0, 'h0'); _f32c(c0, 'c0')
    a_ih, a_hh, a_bi, a_bh = ptr_array(w_ih), ptr_array(w_hh), ptr_array(b_ih), ptr_array(b_hh)
    if not expect_backward:
        lib().halo_set_lstm_expect_backward(0)
        lib().halo_set_lstm_weights_stamp(int(weights_stamp))
    try:
        check(lib().halo_lstm_fwd(ptr(x_tm), a_ih, a_hh, a_bi, a_bh, ptr(h0), ptr(c0), ptr(y), y_strides[0], y_strides[1],
                                  int(y_relu), ptr(hn), ptr(cn), ptr(reserve), T, B, in0, H, L, drop.p, drop.seed,
                                  drop.offset, drop.counter_ptr, _stream()), 'halo_lstm_fwd')
    finally:
        if not expect_backward:
            lib().halo_set_lstm_expect_backward(1)
            lib().halo_set_lstm_weights_stamp(0)
    return y, hn, cn, reserve


def lstm_reserve(T, B, in0, H, L, device):
    """A reserve buffer the caller owns (lstm_fwd(reserve=...))."""
    return torch.empty((lib().halo_lstm_reserve_bytes(T, B, in0, H, L) + 3) // 4, device=device, dtype=torch.float32)


def lstm_bwd_workspace(x_tm, w_hh):
    T, B, in0 = x_tm.shape
    return torch.empty((lib().halo_lstm_bwd_workspace_bytes(T, B, in0, w_hh[0].shape[1], len(w_hh)) + 3) // 4,
                       device=x_tm.device, dtype=torch.float32)


def lstm_bwd(x_tm, w_ih, w_hh, dy, y_strides, y_relu, reserve, dhn=None, dcn=None, want_dx=False,
             grads=None, drop=NO_DROPOUT, layers=None, workspace=None, dx=None, dx_slabs=1):
    """Returns (dx or None, grads dict of lists dw_ih/dw_hh/db_ih/db_hh).  ``grads`` may carry
    preallocated outputs.  ``layers=(lo, hi)`` runs only that range (top down); a split backward
    passes the same ``workspace`` (ops.lstm_bwd_workspace) to both calls.
    ``dx_slabs`` > 1: ``dx`` has room for that many [T, B, in] matrices and the call may leave the input gradient there as unreduced
    K-slices; ``lstm_dx_slabs_left()`` right after the call says how many (1: dx[0] is the gradient), for ``subsample_bwd(slabs=)``."""
    T, B, in0 = x_tm.shape
    L = len(w_hh)
    H = w_hh[0].shape[1]
    dev = x_tm.device
    if grads is None:
        grads = {
            'dw_ih': [torch.empty_like(w) for w in w_ih],
            'dw_hh': [torch.empty_like(w) for w in w_hh],
            'db_ih': [torch.empty(4 * H, device=dev, dtype=torch.float32) for _ in range(L)],
            'db_hh': [torch.empty(4 * H, device=dev, dtype=torch.float32) for _ in range(L)],
        }
    ws = workspace if workspace is not None else lstm_bwd_workspace(x_tm, w_hh)
    lo, hi = layers if layers is not None else (0, L)
    if dx is None and want_dx and lo == 0:
        dx = torch.empty(T, B, in0, device=dev, dtype=torch.float32)
    a_ih, a_hh = ptr_array(w_ih), ptr_array(w_hh)
    g_ih, g_hh = ptr_array(grads['dw_ih']), ptr_array(grads['dw_hh'])
    g_bi, g_bh = ptr_array(grads['db_ih']), ptr_array(grads['db_hh'])
    if dx_slabs > 1:
        if dx is None or dx.numel() < dx_slabs * T * B * in0:
            raise ValueError('haloop_amd.ops.lstm_bwd: dx_slabs needs a dx buffer of that many [T, B, in] matrices')
        check(lib().halo_set_lstm_dx_slabs(int(dx_slabs)), 'halo_set_lstm_dx_slabs')
    try:
        check(lib().halo_lstm_bwd(ptr(x_tm), a_ih, a_hh, ptr(dy), y_strides[0], y_strides[1], int(y_relu), ptr(dhn),
                                  ptr(dcn), ptr(reserve), ptr(ws), ptr(dx), g_ih, g_hh, g_bi, g_bh, T, B, in0, H, L, lo, hi,
                                  drop.p, drop.seed, drop.offset, drop.counter_ptr, _stream()), 'halo_lstm_bwd')
    finally:
        if dx_slabs > 1:
            lib().halo_set_lstm_dx_slabs(1)
    return dx, grads


class defer_small_jobs:
    """``with ops.defer_small_jobs():`` -- the small fixed-order reductions at the end of a backward pass (the CTC head's partial sums, the
    two-layer LSTM launch's bias-gradient partials) are queued on the context instead of launched, and ride in the tail blocks of the conv
    backward's reduce launch (``subsample_bwd``); whatever is still queued at the end of the block runs in one launch of its own.  The
    buffers those reductions read must live until then (``ctc_head_bwd(workspace=)``, the LSTM backward's workspace)."""

    def __enter__(self):
        self.begin()
        return self

    def __exit__(self, *exc):
        self.end()
        return False

    @staticmethod
    def begin():
        check(lib().halo_set_defer_small_jobs(1), 'halo_set_defer_small_jobs')

    @staticmethod
    def end():
        try:
            check(lib().halo_flush_small_jobs(_stream()), 'halo_flush_small_jobs')
        finally:
            lib().halo_set_defer_small_jobs(0)


def lstm_dx_slabs_left():
    """How many K-slices the last ``lstm_bwd`` of this thread's context left in its dx buffer (1: the gradient itself)."""
    return int(lib().halo_lstm_dx_slabs_left())


def log_softmax_fwd(x2d):
    _f32c(x2d, 'logits')
    y = torch.empty_like(x2d)
    check(lib().halo_log_softmax_fwd(ptr(x2d), ptr(y), x2d.shape[0], x2d.shape[1], _stream()), 'halo_log_softmax_fwd')
    return y


def log_softmax_bwd(dy2d, y2d):
    _f32c(dy2d, 'dy'); _f32c(y2d, 'y')
    dx = torch.empty_like(y2d)
    check(lib().halo_log_softmax_bwd(ptr(dy2d), ptr(y2d), ptr(dx), y2d.shape[0], y2d.shape[1], _stream()),
          'halo_log_softmax_bwd')
    return dx


def colsum(x2d, out=None):
    _f32c(x2d, 'x')
    if out is None:
        out = torch.empty(x2d.shape[1], device=x2d.device, dtype=torch.float32)
    check(lib().halo_colsum(ptr(x2d), x2d.shape[0], x2d.shape[1], x2d.shape[1], ptr(out), _stream()), 'halo_colsum')
    return out


def _i64c(t, name):
    if t.dtype != torch.int64:
        t = t.to(torch.int64)
    if not t.is_cuda:
        raise ValueError(f'{name}: expected a HIP tensor')
    return t.contiguous()


def ctc_fwd(lp, time_major, targets, input_lengths, target_lengths, flags=0):
    """lp: [T,N,C] (time_major) or [N,T,C]; any strides with a unit class stride.  -> (nll [N], alpha)."""
    if lp.dtype != torch.float32 or lp.stride(-1) != 1:
        raise ValueError('log-probs must be float32 with unit class stride')
    if time_major:
        T, N, Cn = lp.shape
        st, sn = lp.stride(0), lp.stride(1)
    else:
        N, T, Cn = lp.shape
        sn, st = lp.stride(0), lp.stride(1)
    targets = _i64c(targets, 'targets')
    if targets.dim() != 2:
        raise ValueError('targets must be [N,S] (padded); concatenated targets are not supported')
    S = targets.shape[1]
    tl = _i64c(target_lengths, 'target_lengths')
    il = None if input_lengths is None else _i64c(input_lengths, 'input_lengths')
    alpha = torch.empty(N, T, 2 * S + 1, device=lp.device, dtype=torch.float32)
    nll = torch.empty(N, device=lp.device, dtype=torch.float32)
    check(lib().halo_ctc_fwd(ptr(lp), st, sn, T, N, Cn, ptr(targets), targets.stride(0), S, ptr(il), ptr(tl), flags,
                             ptr(alpha), ptr(nll), _stream()), 'halo_ctc_fwd')
    return nll, alpha, (targets, il, tl)


def ctc_bwd(lp, time_major, saved, alpha, nll, grad_out):
    targets, il, tl = saved
    if time_major:
        T, N, Cn = lp.shape
        st, sn = lp.stride(0), lp.stride(1)
    else:
        N, T, Cn = lp.shape
        sn, st = lp.stride(0), lp.stride(1)
    S = targets.shape[1]
    grad = torch.empty(lp.shape, device=lp.device, dtype=torch.float32)
    gst, gsn = (grad.stride(0), grad.stride(1)) if time_major else (grad.stride(1), grad.stride(0))
    beta = torch.empty_like(alpha)
    grad_out = grad_out.to(torch.float32).contiguous()
    check(lib().halo_ctc_bwd(ptr(lp), st, sn, T, N, Cn, ptr(targets), targets.stride(0), S, ptr(il), ptr(tl),
                             ptr(alpha), ptr(nll), ptr(grad_out), ptr(beta), ptr(grad), gst, gsn, _stream()),
          'halo_ctc_bwd')
    return grad


def ctc_prepare(input_lengths, target_lengths, ks=5, stride=4, pad=3):
    """-> (feature_lengths int64 [N], grad_out f32 [N]) in one launch."""
    il, tl = _i64c(input_lengths, 'input_lengths'), _i64c(target_lengths, 'target_lengths')
    n = il.numel()
    flen = torch.empty(n, device=il.device, dtype=torch.int64)
    gout = torch.empty(n, device=il.device, dtype=torch.float32)
    check(lib().halo_ctc_prepare(ptr(il), ptr(tl), n, ks, stride, pad, ptr(flen), ptr(gout), _stream()), 'halo_ctc_prepare')
    return flen, gout


def ctc_mean_loss(nll, target_lengths, out):
    tl = _i64c(target_lengths, 'target_lengths')
    check(lib().halo_ctc_mean_loss(ptr(nll), ptr(tl), nll.numel(), ptr(out), _stream()), 'halo_ctc_mean_loss')
    return out


def ctc_head_supported(T, H, V, S):
    return bool(lib().halo_ctc_head_supported(T, H, V, S))


def ctc_head_fwd(feats, weight, bias, drop, stream_id, input_lengths, targets, target_lengths, loss, ticket, ks=5, stride=4, pad=3):
    """The CTC head's forward in one launch (include/halo.h): feats [B,T,H] -> (lp, alpha, nll, feature_lengths, grad_out); the mean
    loss goes to ``loss`` (device scalar); ``ticket``: a zero-initialised device int32 the caller keeps."""
    _f32c(feats, 'features')
    B, T, H = feats.shape
    V = weight.shape[0]
    tg = _i64c(targets, 'targets')
    il, tl = _i64c(input_lengths, 'input_lengths'), _i64c(target_lengths, 'target_lengths')
    S = tg.shape[1]
    dev = feats.device
    lp = torch.empty(B, T, V, device=dev, dtype=torch.float32)
    alpha = torch.empty(B, T, 2 * S + 1, device=dev, dtype=torch.float32)
    nll = torch.empty(B, device=dev, dtype=torch.float32)
    flen = torch.empty(B, device=dev, dtype=torch.int64)
    grad_out = torch.empty(B, device=dev, dtype=torch.float32)
    check(lib().halo_ctc_head_fwd(ptr(feats), ptr(weight), ptr(bias), drop.p, drop.seed, stream_id, drop.offset, drop.counter_ptr,
                                  ptr(il), ks, stride, pad, ptr(tg), tg.stride(0), S, ptr(tl), ptr(lp), ptr(alpha), ptr(nll), ptr(flen),
                                  ptr(grad_out), ptr(loss), ptr(ticket), B, T, H, V, _stream()), 'halo_ctc_head_fwd')
    return lp, alpha, nll, flen, grad_out, (tg, tl)


def ctc_head_train_workspace(B, H, V, device):
    return torch.empty(lib().halo_ctc_head_train_workspace_bytes(B, H, V), device=device, dtype=torch.uint8)


def ctc_head_train_ticket(B, H, device):
    """The zero-initialised words ``ctc_head_train`` keeps between calls (loss ticket, launch count, the slices' tagged partial logits)."""
    return torch.zeros(lib().halo_ctc_head_train_ticket_words(B, H), device=device, dtype=torch.int32)


def ctc_head_train(feats, weight, bias, drop, stream_id, input_lengths, targets, target_lengths, loss, ticket, dweight, dbias,
                   workspace=None, want_lp=False, ks=5, stride=4, pad=3):
    """The CTC head of a training step, forward and backward, in ONE launch (include/halo.h: halo_ctc_head_train; not in the exact-f32
    mode).  feats [B,T,H] -> (d features [B,T,H], nll [B], feature_lengths [B], lp or None); the mean loss goes to ``loss``, the classifier's
    gradients to dweight / dbias (deferred like ``ctc_head_bwd``'s).  ``ticket``: ``ctc_head_train_ticket(B, H, device)``, the caller's to keep between calls."""
    _f32c(feats, 'features')
    B, T, H = feats.shape
    V = weight.shape[0]
    tg = _i64c(targets, 'targets')
    il, tl = _i64c(input_lengths, 'input_lengths'), _i64c(target_lengths, 'target_lengths')
    need = lib().halo_ctc_head_train_ticket_words(B, H)
    if ticket.numel() < need:
        raise ValueError(f'ctc_head_train: ticket must hold {need} words (ctc_head_train_ticket), has {ticket.numel()}')
    dev = feats.device
    lp = torch.empty(B, T, V, device=dev, dtype=torch.float32) if want_lp else None
    nll = torch.empty(B, device=dev, dtype=torch.float32)
    flen = torch.empty(B, device=dev, dtype=torch.int64)
    dfeats = torch.empty_like(feats)
    ws = workspace if workspace is not None else ctc_head_train_workspace(B, H, V, dev)
    check(lib().halo_ctc_head_train(ptr(feats), ptr(weight), ptr(bias), drop.p, drop.seed, stream_id, drop.offset, drop.counter_ptr,
                                    ptr(il), ks, stride, pad, ptr(tg), tg.stride(0), tg.shape[1], ptr(tl), ptr(lp) if want_lp else None,
                                    ptr(nll), ptr(flen), ptr(loss), ptr(ticket), ptr(dfeats), ptr(dweight), ptr(dbias), ptr(ws),
                                    B, T, H, V, _stream()), 'halo_ctc_head_train')
    return dfeats, nll, flen, lp


def ctc_head_workspace(B, H, V, device):
    return torch.empty(lib().halo_ctc_head_workspace_bytes(B, H, V), device=device, dtype=torch.uint8)


def ctc_head_bwd(feats, weight, drop, stream_id, flen, tg, tl, lp, alpha, nll, grad_out, dweight, dbias, workspace=None):
    """The CTC head's backward (two launches): returns d features [B,T,H]; the classifier's gradients go to dweight / dbias.
    Inside ``defer_small_jobs()`` the second launch (the fixed-order sum of the per-utterance partials) is queued instead and dweight /
    dbias are complete after the block's flush; the caller then passes a ``workspace`` (``ctc_head_workspace``) that lives that long."""
    B, T, H = feats.shape
    V = weight.shape[0]
    dfeats = torch.empty_like(feats)
    ws = workspace if workspace is not None else ctc_head_workspace(B, H, V, feats.device)
    check(lib().halo_ctc_head_bwd(ptr(feats), ptr(weight), drop.p, drop.seed, stream_id, drop.offset, drop.counter_ptr, ptr(flen), ptr(tg),
                                  tg.stride(0), tg.shape[1], ptr(tl), ptr(lp), ptr(alpha), ptr(nll), ptr(grad_out), ptr(dfeats),
                                  ptr(dweight), ptr(dbias), ptr(ws), B, T, H, V, _stream()), 'halo_ctc_head_bwd')
    return dfeats


def ctc_greedy(lp):
    _f32c(lp, 'log-probs')
    N, T, Cn = lp.shape
    dev = lp.device
    ali = torch.empty(N, T, device=dev, dtype=torch.int64)
    scores = torch.empty(N, T, device=dev, dtype=torch.float32)
    hyp = torch.zeros(N, T, device=dev, dtype=torch.int64)
    hyp_len = torch.empty(N, device=dev, dtype=torch.int64)
    check(lib().halo_ctc_greedy(ptr(lp), N, T, Cn, ptr(ali), ptr(scores), ptr(hyp), ptr(hyp_len), _stream()),
          'halo_ctc_greedy')
    return ali, scores, hyp, hyp_len


def ctc_head_greedy(feats, weight, bias, want_lp=False):
    """feats [B,T,H] -> (alignments, scores, hyp (zero padded), hyp_len[, log-probs]): classifier + log_softmax + greedy collapse in one launch."""
    _f32c(feats, 'features')
    B, T, H = feats.shape
    V = weight.shape[0]
    dev = feats.device
    ali = torch.empty(B, T, device=dev, dtype=torch.int64)
    scores = torch.empty(B, T, device=dev, dtype=torch.float32)
    hyp = torch.empty(B, T, device=dev, dtype=torch.int64)
    hyp_len = torch.empty(B, device=dev, dtype=torch.int64)
    lp = torch.empty(B, T, V, device=dev, dtype=torch.float32) if want_lp else None
    check(lib().halo_ctc_head_greedy(ptr(feats), ptr(weight), ptr(bias), ptr(lp), ptr(ali), ptr(scores), ptr(hyp), ptr(hyp_len), B, T, H, V,
                                     _stream()), 'halo_ctc_head_greedy')
    return (ali, scores, hyp, hyp_len, lp) if want_lp else (ali, scores, hyp, hyp_len)


def ctc_beam(em, beam, log_domain=True):
    """em [N,T,V] -> (seqs [N,beam,T] int64, lens [N,beam] int32, scores [N,beam])."""
    _f32c(em, 'emissions')
    N, T, V = em.shape
    dev = em.device
    seqs = torch.empty(N, beam, T, device=dev, dtype=torch.int64)
    lens = torch.empty(N, beam, device=dev, dtype=torch.int32)
    scores = torch.empty(N, beam, device=dev, dtype=torch.float32)
    ws = torch.empty(lib().halo_ctc_beam_workspace_bytes(N, T, V, beam), device=dev, dtype=torch.uint8)
    check(lib().halo_ctc_beam(ptr(em), N, T, V, beam, int(log_domain), ptr(seqs), ptr(lens), ptr(scores), ptr(ws),
                              _stream()), 'halo_ctc_beam')
    return seqs, lens, scores


def logaddexp_aten(a, b):
    """torch.logaddexp(a, b) with ATen's CPU arithmetic bit for bit (1-D contiguous float32; see halo_set_beam_vector_chunk)."""
    _f32c(a, 'a'); _f32c(b, 'b')
    out = torch.empty_like(a)
    check(lib().halo_logaddexp_aten(ptr(a), ptr(b), ptr(out), a.numel(), _stream()), 'halo_logaddexp_aten')
    return out


def topk(values2d, k):
    """Row-wise top-k in torch.topk's CPU order (ties included). -> (values [rows,k], indices [rows,k] int64)"""
    _f32c(values2d, 'values')
    rows, n = values2d.shape
    dev = values2d.device
    vals = torch.empty(rows, k, device=dev, dtype=torch.float32)
    idx = torch.empty(rows, k, device=dev, dtype=torch.int64)
    ws = torch.empty(rows * n * 8, device=dev, dtype=torch.uint8)
    check(lib().halo_topk_f32(ptr(values2d), rows, n, k, ptr(vals), ptr(idx), ptr(ws), _stream()), 'halo_topk_f32')
    return vals, idx


def sumsq_partials(flat, partials=None):
    if partials is None:
        partials = torch.empty(_lib.HALO_SUMSQ_PARTS, device=flat.device, dtype=torch.float32)
    check(lib().halo_sumsq(ptr(flat), flat.numel(), ptr(partials), _stream()), 'halo_sumsq')
    return partials


def _range_arrays(ranges):
    n = len(ranges)
    return n, (C.c_size_t * n)(*[int(r[0]) for r in ranges]), (C.c_size_t * n)(*[int(r[1]) for r in ranges])


def sumsq_ranges(flat, ranges, partials):
    """partials[0:HALO_SUMSQ_PARTS] <- squared-norm partials of the concatenation of flat[lo:hi] for (lo, hi) in ranges (<= 8, multiples of 4)."""
    n, b, e = _range_arrays(ranges)
    check(lib().halo_sumsq_ranges(ptr(flat), n, b, e, ptr(partials), _stream()), 'halo_sumsq_ranges')
    return partials


def pack_ranges_bf16(flat, ranges, dst):
    """dst (bfloat16, contiguous) <- the concatenation of flat[lo:hi] for (lo, hi) in ranges, rounded to nearest-even."""
    n, b, e = _range_arrays(ranges)
    check(lib().halo_pack_ranges_bf16(ptr(flat), n, b, e, ptr(dst), _stream()), 'halo_pack_ranges_bf16')
    return dst


def expand_ranges_bf16(stage, spans, chunks, world, skip_rank, flat):
    """flat[span k] <- the gathered bf16 records of every rank but skip_rank (include/halo.h halo_expand_ranges_bf16)."""
    n = len(spans)
    b = (C.c_size_t * n)(*[int(s) for s in spans])
    c = (C.c_size_t * n)(*[int(x) for x in chunks])
    check(lib().halo_expand_ranges_bf16(ptr(stage), n, b, c, int(world), int(skip_rank), ptr(flat), _stream()), 'halo_expand_ranges_bf16')
    return flat


GRAD_SUMSQ_ALL = 31        # halo_grad_sumsq_state: both layers' matrices and biases and the conv front end have contributed


def collect_grad_sumsq(partials):
    """The backward launches that follow write the squared-norm partials of the clipped gradients they store into ``partials`` (float32;
    None: off) -- see ``grad_sumsq_state``."""
    check(lib().halo_set_grad_sumsq(ptr(partials), 0 if partials is None else partials.numel()), 'halo_set_grad_sumsq')


def grad_sumsq_state():
    """(slots written, producer bits) since ``collect_grad_sumsq``: the partials are the whole clipped norm when bits == GRAD_SUMSQ_ALL
    and the clipped parameters are the conv front end + a 2-layer LSTM."""
    n, bits = C.c_int(0), C.c_uint(0)
    check(lib().halo_grad_sumsq_state(C.byref(n), C.byref(bits)), 'halo_grad_sumsq_state')
    return int(n.value), int(bits.value)


def clip_coef(partials, count, max_norm, coef, norm, applied_steps=None):
    """applied_steps: optional device int32 counter advanced when the norm is finite (the Adam step count of applied updates)."""
    check(lib().halo_clip_coef_step(ptr(partials), count, float(max_norm), ptr(coef), ptr(norm), ptr(applied_steps), _stream()),
          'halo_clip_coef_step')


def adamw(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_scale=None):
    """In-place AdamW on flat views.  Pass views that share the parameter's version counter (``param.detach().view(-1)``,
    not ``param.data``): the update bumps it so that cached operand images of the weight (haloop_amd/_linear.py) are rebuilt."""
    _lib.bump_weights_epoch()
    check(lib().halo_adamw(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), lr, beta1, beta2, eps, weight_decay, step,
                           ptr(grad_scale), _stream()), 'halo_adamw')
    torch.autograd.graph.increment_version(p)


def adamw_ranges(p, g, m, v, ranges, lr, beta1, beta2, eps, step, counter=None):
    """One-launch AdamW over ``ranges`` = [(begin, end, weight_decay, grad_scale tensor or None), ...] of flat buffers.
    ``step``: the 1-based update count as a python int, or a device int32 tensor holding it (no host scalar: graph-capturable).
    ``lr``: a python float, or (with a device ``step``) a one-element float32 device tensor the launch reads it from."""
    _lib.bump_weights_epoch()
    n = len(ranges)
    begin = (C.c_size_t * n)(*[r[0] for r in ranges])
    end = (C.c_size_t * n)(*[r[1] for r in ranges])
    wd = (C.c_float * n)(*[float(r[2]) for r in ranges])
    gs = (C.c_void_p * n)(*[ptr(r[3]) for r in ranges])
    if torch.is_tensor(step):
        lr_dev = lr if torch.is_tensor(lr) else None
        check(lib().halo_adamw_ranges_dev(ptr(p), ptr(g), ptr(m), ptr(v), n, begin, end, wd, gs, 0.0 if lr_dev is not None else lr,
                                          ptr(lr_dev), beta1, beta2, eps, ptr(step), ptr(counter), _stream()), 'halo_adamw_ranges_dev')
    else:
        check(lib().halo_adamw_ranges(ptr(p), ptr(g), ptr(m), ptr(v), n, begin, end, wd, gs, lr, beta1, beta2, eps, step, ptr(counter),
                                      _stream()), 'halo_adamw_ranges')
    torch.autograd.graph.increment_version(p)


class AdamWMulti:
    """torch.optim.AdamW arithmetic over a whole parameter list in ONE launch per 256 tensors (the reference's fused AdamW,
    ha/attention_loop.py:141-147).  ``params``: the nn.Parameters (contiguous fp32 on the HIP device); ``weight_decays``: one value
    per parameter.  The optimizer state lives here; ``step()`` reads each parameter's current ``.grad``."""

    def __init__(self, params, weight_decays, lr, betas=(0.9, 0.999), eps=1e-8):
        import numpy as np
        self.params = list(params)
        self.wds = [float(w) for w in weight_decays]
        self.param_groups = [{'lr': float(lr), 'params': self.params}]
        self.betas, self.eps = betas, float(eps)
        for p in self.params:
            if p.dtype != torch.float32 or not p.is_contiguous() or not p.is_cuda:
                raise ValueError('AdamWMulti: parameters must be contiguous float32 tensors on the HIP device')
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]
        self.t = 0
        dev = self.params[0].device
        chunk, group = lib().halo_adamw_multi_chunk(), lib().halo_adamw_multi_max_tensors()
        assert lib().halo_adamw_multi_tensor_bytes() == 40
        rec = np.dtype([('p', '<u8'), ('m', '<u8'), ('v', '<u8'), ('n', '<u8'), ('decay', '<f4'), ('pad', '<i4')])
        self.groups = []                                        # (first parameter, count, device tensor table, device chunk table, chunks)
        for first in range(0, len(self.params), group):
            idx = range(first, min(first + group, len(self.params)))
            table = np.zeros(len(idx), dtype=rec)
            for k, i in enumerate(idx):
                p = self.params[i]
                table[k] = (p.data_ptr(), self.m[i].data_ptr(), self.v[i].data_ptr(), p.numel(), self.wds[i], 0)
            pairs = [(k, c) for k, i in enumerate(idx) for c in range((self.params[i].numel() + chunk - 1) // chunk)]
            self.groups.append((first, len(idx), torch.from_numpy(table.view(np.uint8)).to(dev),
                                torch.tensor(pairs, dtype=torch.int32).view(-1).to(dev), len(pairs)))

    @property
    def lr(self):
        return self.param_groups[0]['lr']

    @lr.setter
    def lr(self, value):
        self.param_groups[0]['lr'] = float(value)

    def step(self, grad_scale=None):
        _lib.bump_weights_epoch()
        self.t += 1
        for first, count, table, chunks, n_chunks in self.groups:
            ptrs = (C.c_void_p * count)()
            for k in range(count):
                p = self.params[first + k]
                g = p.grad
                if g is None or g.dtype != torch.float32 or not g.is_contiguous() or g.device != p.device or g.shape != p.shape:
                    raise ValueError('AdamWMulti.step: every parameter needs a contiguous float32 .grad of its shape on its device')
                ptrs[k] = g.data_ptr()
            check(lib().halo_adamw_multi(ptr(table), ptr(chunks), n_chunks, ptrs, count, self.lr, self.betas[0], self.betas[1], self.eps,
                                         self.t, ptr(grad_scale), _stream()), 'halo_adamw_multi')
        for p in self.params:
            torch.autograd.graph.increment_version(p)          # cached operand images of the weights are rebuilt


def scale_add_(y, x, alpha, beta, guard=None):
    """y <- alpha*y + beta*x in place (flat fp32 buffers); alpha == 0 never reads y.  guard: optional device scalar; when it is not
    finite, x contributes nothing (a micro-batch with a NaN/Inf loss is dropped, ha/loop.py:167-174)."""
    check(lib().halo_scale_add_guarded(ptr(y), ptr(x), float(alpha), float(beta), y.numel(), ptr(guard), _stream()),
          'halo_scale_add_guarded')
    return y


def cast_f32_to_bf16_(dst_bf16, src_f32):
    check(lib().halo_cast_f32_bf16(ptr(src_f32), ptr(dst_bf16), src_f32.numel(), _stream()), 'halo_cast_f32_bf16')
    return dst_bf16


def cast_bf16_to_f32_(dst_f32, src_bf16, scale=1.0):
    check(lib().halo_cast_bf16_f32(ptr(src_bf16), ptr(dst_f32), float(scale), dst_f32.numel(), _stream()), 'halo_cast_bf16_f32')
    return dst_f32


def counter_inc(counter):
    check(lib().halo_counter_inc(ptr(counter), _stream()), 'halo_counter_inc')


# ---- GPT forward operators ------------------------------------------------------------------------
def embed_fwd(ids, wte, wpe, pos0=0):
    ids = _i64c(ids, 'input_ids')
    Bn, T = ids.shape
    C = wte.shape[1]
    x = torch.empty(Bn * T, C, device=wte.device, dtype=torch.float32)
    check(lib().halo_embed_fwd(ptr(ids), ptr(wte), ptr(wpe), ptr(x), Bn * T, T, C, pos0, wte.shape[0], _stream()), 'halo_embed_fwd')
    return x


def layernorm_fwd(x2d, weight, bias=None, eps=1e-5):
    _f32c(x2d, 'x')
    y = torch.empty_like(x2d)
    check(lib().halo_layernorm_fwd(ptr(x2d), ptr(weight), ptr(bias), ptr(y), x2d.shape[0], x2d.shape[1], eps, _stream()),
          'halo_layernorm_fwd')
    return y


def attention_causal_fwd(qkv2d, B, T, n_head):
    _f32c(qkv2d, 'qkv')
    C = qkv2d.shape[1] // 3
    y = torch.empty(B * T, C, device=qkv2d.device, dtype=torch.float32)
    check(lib().halo_attention_causal_fwd(ptr(qkv2d), ptr(y), B, T, n_head, C, _stream()), 'halo_attention_causal_fwd')
    return y


def cross_entropy_fwd(logits2d, targets, ignore_index=0):
    _f32c(logits2d, 'logits')
    tg = _i64c(targets.reshape(-1), 'targets')
    loss = torch.empty(logits2d.shape[0], device=logits2d.device, dtype=torch.float32)
    check(lib().halo_cross_entropy_fwd(ptr(logits2d), ptr(tg), ptr(loss), logits2d.shape[0], logits2d.shape[1],
                                       logits2d.shape[1], ignore_index, _stream()), 'halo_cross_entropy_fwd')
    return loss


# ---- attention / rotary / KV-cache operators (GPT and the enc-dec ASR path) ------------------------------
def attention_fwd(q, k, v, N, heads, head_dim, Tq, Tk, causal=False, key_lengths=None, want_lse=False, want_entropy=False,
                  drop=NO_DROPOUT, stream_id=0):
    """q: rows [N*Tq, >=C] (a column slice of a packed GEMM output is fine), k/v: rows [N*Tk, ...] sharing one row stride.
    -> y [N*Tq, C] (+ lse / entropy [N, heads, Tq] when asked).  ``drop``: dropout on the attention probabilities."""
    C = heads * head_dim
    for t in (q, k, v):
        if t.dtype != torch.float32 or not t.is_cuda or t.stride(-1) != 1:
            raise ValueError('attention operands must be float32 HIP tensors with a unit column stride')
    if k.stride(0) != v.stride(0):
        raise ValueError('k and v must share a row stride')
    dev = q.device
    y = torch.empty(N * Tq, C, device=dev, dtype=torch.float32)
    lse = torch.empty(N, heads, Tq, device=dev, dtype=torch.float32) if want_lse else None
    ent = torch.empty(N, heads, Tq, device=dev, dtype=torch.float32) if want_entropy else None
    if key_lengths is not None:
        key_lengths = key_lengths.to(device=dev, dtype=torch.int32).contiguous()
    check(lib().halo_attention_fwd_strided(ptr(q), q.stride(0), q.stride(0) * Tq, head_dim, ptr(k), ptr(v), k.stride(0),
                                           k.stride(0) * Tk, head_dim, ptr(y), C, C * Tq, ptr(lse), ptr(ent), N, heads, head_dim, Tq, Tk,
                                           int(causal), ptr(key_lengths), drop.p, drop.seed, stream_id, drop.offset, drop.counter_ptr,
                                           _stream()), 'halo_attention_fwd')
    return y, lse, ent


class RopeTable:
    """cos/sin tables of rotate_interleaved (ha/transformer.py:16-31) for positions [0, T)."""

    def __init__(self, T, head_dim, device, base=10000.0):
        self.T, self.head_dim = T, head_dim
        self.cos = torch.empty(T, head_dim // 2, device=device, dtype=torch.float32)
        self.sin = torch.empty(T, head_dim // 2, device=device, dtype=torch.float32)
        check(lib().halo_rope_table(ptr(self.cos), ptr(self.sin), T, head_dim, float(base), _stream()), 'halo_rope_table')


def rope_(x2d, T, heads, head_dim, table, t0=0, inverse=False):
    """Rotate in place the heads*head_dim leading columns of the rows of x2d (row r is at position t0 + r % T)."""
    if x2d.dtype != torch.float32 or not x2d.is_cuda or x2d.stride(-1) != 1:
        raise ValueError('rope: expected a float32 HIP tensor with a unit column stride')
    check(lib().halo_rope_interleaved(ptr(x2d), x2d.stride(0), x2d.shape[0], T, heads, head_dim, t0, ptr(table.cos), ptr(table.sin),
                                      table.T, int(inverse), _stream()), 'halo_rope_interleaved')
    return x2d


def attention_cached_fwd(q, cache_k, cache_v, Tq, n_keys, causal=True):
    """q: rows [N*Tq, C] (a column slice is fine); cache_{k,v}: [N, heads, Tc, head_dim] fp32 holding n_keys valid keys.
    -> y [N*Tq, C]: attend_cached of ha/attention.py:64-93 (queries are the LAST Tq of the n_keys positions)."""
    N, heads, Tc, hd = cache_k.shape
    C = heads * hd
    y = torch.empty(N * Tq, C, device=q.device, dtype=torch.float32)
    check(lib().halo_attention_fwd_strided(ptr(q), q.stride(0), q.stride(0) * Tq, hd, ptr(cache_k), ptr(cache_v), hd, heads * Tc * hd,
                                           Tc * hd, ptr(y), C, C * Tq, None, None, N, heads, hd, Tq, n_keys, int(causal), None,
                                           0.0, 0, 0, 0, None, _stream()), 'halo_attention_fwd_strided')
    return y


def kv_cache_store(src2d, v_offset, cache_k, cache_v, N, S, heads, head_dim, t0):
    """cache_{k,v} [N, heads, Tc, head_dim] float16 (or float32) <- rows [N*S] of src2d (k at column 0, v at column v_offset)."""
    if cache_k.dtype == torch.float32:
        check(lib().halo_kv_cache_store_f32(ptr(src2d), src2d.stride(0), v_offset, ptr(cache_k), ptr(cache_v), N, S, heads, head_dim,
                                            cache_k.shape[2], t0, _stream()), 'halo_kv_cache_store_f32')
        return
    check(lib().halo_kv_cache_store(ptr(src2d), src2d.stride(0), v_offset, ptr(cache_k), ptr(cache_v), N, S, heads, head_dim,
                                    cache_k.shape[2], t0, _stream()), 'halo_kv_cache_store')


def attention_decode(q2d, cache_k, cache_v, n_keys, key_lengths=None, table=None, out=None):
    N, heads, Tc, hd = cache_k.shape
    if out is None:
        out = torch.empty(N, heads * hd, device=q2d.device, dtype=torch.float32)
    check(lib().halo_attention_decode(ptr(q2d), q2d.stride(0), ptr(cache_k), ptr(cache_v), ptr(out), out.stride(0), N, heads, hd, Tc,
                                      n_keys, ptr(key_lengths), ptr(table.cos) if table else None, ptr(table.sin) if table else None,
                                      _stream()), 'halo_attention_decode')
    return out


def attention_decode_step(q2d, k2d, v2d, cache_k, cache_v, n_keys, table=None):
    """Self-attention of one decode step in one launch: stores k2d / v2d (fp16) at cache position n_keys - 1, rotates q and the
    cached keys (table), attends over n_keys positions.  q2d / k2d / v2d: column slices of one packed row buffer."""
    N, heads, Tc, hd = cache_k.shape
    if not (q2d.stride(0) == k2d.stride(0) == v2d.stride(0)):
        raise ValueError('q, k, v must share a row stride')
    out = torch.empty(N, heads * hd, device=q2d.device, dtype=torch.float32)
    check(lib().halo_attention_decode_step(ptr(q2d), ptr(k2d), ptr(v2d), q2d.stride(0), ptr(cache_k), ptr(cache_v), ptr(out),
                                           out.stride(0), N, heads, hd, Tc, n_keys, ptr(table.cos) if table else None,
                                           ptr(table.sin) if table else None, _stream()), 'halo_attention_decode_step')
    return out


def logprob_max(logits2d, want_entropy=False):
    _f32c(logits2d, 'logits')
    rows, V = logits2d.shape
    dev = logits2d.device
    val = torch.empty(rows, device=dev, dtype=torch.float32)
    idx = torch.empty(rows, device=dev, dtype=torch.int64)
    ne = torch.empty(rows, device=dev, dtype=torch.float32) if want_entropy else None
    check(lib().halo_logprob_max(ptr(logits2d), V, rows, V, ptr(val), ptr(idx), ptr(ne), _stream()), 'halo_logprob_max')
    return val, idx, ne


def greedy_update(val, idx, negent, tokens, t, plen, etx, alive, out_len, log_probs, sum_ent):
    check(lib().halo_greedy_update(ptr(val), ptr(idx), ptr(negent), ptr(tokens), tokens.stride(0), t, plen, etx, ptr(alive),
                                   ptr(out_len), ptr(log_probs), ptr(sum_ent), tokens.shape[0], _stream()), 'halo_greedy_update')


# ---- star-CTC and transducer lattices (csrc/lattice.hip) -------------------------------------------------------
def _check_lengths(name, lengths, hi):
    if lengths.numel() and (int(lengths.min()) < 0 or int(lengths.max()) > hi):
        raise ValueError(f'{name} out of range [0, {hi}]')


def star_ctc_fwd(em, targets, emission_lengths, target_lengths, star_penalty, keep=False):
    """em [T, N, C] fp32 log-probabilities (unit column stride) -> (losses [N], workspace or None)."""
    T, N, C = em.shape
    S = targets.shape[1]
    if targets.shape[0] != N or S < 1 or C < 2:
        raise ValueError('star_ctc: targets must be [N, S >= 1] and C >= 2')
    if targets.numel() and (int(targets.min()) < 0 or int(targets.max()) >= C):
        raise ValueError('star_ctc: target label out of range')
    if emission_lengths.numel() and int(emission_lengths.min()) < 1:
        raise ValueError('star_ctc: emission_lengths must be >= 1')
    _check_lengths('star_ctc: emission_lengths', emission_lengths, T)
    _check_lengths('star_ctc: target_lengths', target_lengths, S)
    losses = torch.empty(N, device=em.device, dtype=torch.float32)
    ws = torch.empty(lib().halo_star_ctc_workspace_bytes(T, N, S), device=em.device, dtype=torch.uint8) if keep else None
    check(lib().halo_star_ctc_fwd(ptr(em), em.stride(0), em.stride(1), T, N, C, ptr(targets), S, ptr(emission_lengths), ptr(target_lengths),
                                  star_penalty, ptr(ws), ptr(losses), _stream()), 'halo_star_ctc_fwd')
    return losses, ws


def star_ctc_bwd(em, targets, emission_lengths, target_lengths, star_penalty, workspace, losses, grad_losses):
    T, N, C = em.shape
    if workspace is None:
        raise ValueError('star_ctc backward: the forward did not keep its lattice')
    grad = torch.empty_like(em)
    check(lib().halo_star_ctc_bwd(ptr(em), em.stride(0), em.stride(1), T, N, C, ptr(targets), targets.shape[1], ptr(emission_lengths),
                                  ptr(target_lengths), star_penalty, ptr(workspace), ptr(losses), ptr(grad_losses), ptr(grad), _stream()),
          'halo_star_ctc_bwd')
    return grad


def transducer_fwd(joint, targets, joint_lengths, target_lengths, keep=False):
    """joint [N, T, U+1, K] fp32 contiguous log-probabilities -> (losses [N], workspace or None)."""
    N, T, U1, K = joint.shape
    if targets.numel() and (int(targets.min()) < 0 or int(targets.max()) >= K):
        raise ValueError('transducer: target label out of range')
    if joint_lengths.numel() and int(joint_lengths.min()) < 1:
        raise ValueError('transducer: joint_lengths must be >= 1')
    _check_lengths('transducer: joint_lengths', joint_lengths, T)
    _check_lengths('transducer: target_lengths', target_lengths, U1 - 1)
    losses = torch.empty(N, device=joint.device, dtype=torch.float32)
    ws = torch.empty(lib().halo_transducer_workspace_bytes(N, T, U1), device=joint.device, dtype=torch.uint8) if keep else None
    check(lib().halo_transducer_fwd(ptr(joint), N, T, U1, K, ptr(targets), ptr(joint_lengths), ptr(target_lengths), ptr(ws), ptr(losses),
                                    _stream()), 'halo_transducer_fwd')
    return losses, ws


def transducer_bwd(joint, targets, joint_lengths, target_lengths, workspace, losses, grad_losses):
    N, T, U1, K = joint.shape
    if workspace is None:
        raise ValueError('transducer backward: the forward did not keep its lattice')
    grad = torch.empty_like(joint)
    check(lib().halo_transducer_bwd(ptr(joint), N, T, U1, K, ptr(targets), ptr(joint_lengths), ptr(target_lengths), ptr(workspace),
                                    ptr(losses), ptr(grad_losses), ptr(grad), _stream()), 'halo_transducer_bwd')
    return grad


# ---- fused launches of a greedy decode step (csrc/decode.hip) -------------------------------------------------
def decode_linear_supported(k, layernorm):
    return bool(lib().halo_decode_linear_supported(k, int(layernorm)))


def decode_image(weight):
    """Decode image (split-bf16 fragments in MFMA operand order) of an nn.Linear weight [n_out, k]."""
    _f32c(weight, 'weight')
    n_out, k = weight.shape
    img = torch.empty(lib().halo_decode_image_bytes(n_out, k), device=weight.device, dtype=torch.uint8)
    check(lib().halo_decode_image(ptr(weight), n_out, k, k, ptr(img), _stream()), 'halo_decode_image')
    return img


def decode_linear(x, image, n_out, out, ln_weight=None, eps=1e-5, accumulate=False, gelu=False, x_side=None, side_in=None, side_out=None):
    """out (+)= act(layer_norm?(x) W^T) with W given by its decode image; x [rows, k] and out [rows, >= n_out] fp32 (row strides kept).
    The residual stream as a pair (halo_decode_linear_pair): ``x_side`` (LayerNorm variants) is added to the input rows; ``side_out``
    (accumulating, no LayerNorm, k % 256 == 0) runs the product as two K-slices: out = (out + side_in) + the first half, side_out = the
    second -- the next launch reads out + side_out."""
    rows, k = x.shape
    flags = (_lib.HALO_GEMM_ACCUM if accumulate else 0) | (_lib.HALO_GEMM_GELU_ERF if gelu else 0)
    if x_side is None and side_out is None and side_in is None:
        check(lib().halo_decode_linear(ptr(x), x.stride(0), rows, k, ptr(ln_weight), eps, ptr(image), n_out, ptr(out), out.stride(0), flags,
                                       _stream()), 'halo_decode_linear')
        return out
    for t in (x_side,):
        if t is not None and (t.shape != x.shape or t.stride(0) != x.stride(0)):
            raise ValueError('decode_linear: x_side must have the shape and row stride of x')
    for t in (side_in, side_out):
        if t is not None and (t.shape != out.shape or t.stride(0) != out.stride(0)):
            raise ValueError('decode_linear: side_in / side_out must have the shape and row stride of out')
    check(lib().halo_decode_linear_pair(ptr(x), ptr(x_side), x.stride(0), rows, k, ptr(ln_weight), eps, ptr(image), n_out, ptr(out), ptr(side_in),
                                        ptr(side_out), out.stride(0), flags, _stream()), 'halo_decode_linear_pair')
    return out


def decode_attention_pair(a, mem_k, mem_v, memory_lengths, time_k, time_v, n_keys, table, out):
    """a [N, 4C] = cross query | self q | k | v -> out [N, 2C] = cross-attention output | self-attention output (one launch)."""
    N, heads, S, hd = mem_k.shape
    check(lib().halo_decode_attention_pair(ptr(a), a.stride(0), N, heads, hd, ptr(mem_k), ptr(mem_v), S, ptr(memory_lengths), ptr(time_k),
                                           ptr(time_v), time_k.shape[2], n_keys, ptr(table.cos) if table else None,
                                           ptr(table.sin) if table else None, ptr(out), out.stride(0), _stream()),
          'halo_decode_attention_pair')
    return out


def decode_memory_caches(kv, caches):
    """caches [L, 2, N, heads, S, head_dim] float16 <- kv rows [N*S, L*2C] (layer l: keys at columns l*2C, values at l*2C + C)."""
    L, _, N, heads, S, hd = caches.shape
    check(lib().halo_decode_memory_caches(ptr(kv), kv.stride(0), L, ptr(caches), N, S, heads, hd, _stream()), 'halo_decode_memory_caches')


def decode_token(logits, tokens, t, plen, etx, alive, out_len, log_probs, sum_ent, wte=None, y_next=None):
    """alive: uint8 [2, N], double-buffered -- step t reads plane t & 1 and writes plane (t + 1) & 1."""
    N, V = logits.shape
    if alive.shape != (2, N) or not alive.is_contiguous():
        raise ValueError('decode_token: alive must be a contiguous [2, N] uint8 tensor')
    check(lib().halo_decode_token(ptr(logits), logits.stride(0), N, V, ptr(tokens), tokens.stride(0), t, plen, etx, ptr(alive),
                                  ptr(out_len), ptr(log_probs), ptr(sum_ent), ptr(wte), wte.shape[0] if wte is not None else 0,
                                  wte.shape[1] if wte is not None else 0, ptr(y_next), _stream()), 'halo_decode_token')


# ---- channels-last conv front-end (ha/conv.py) -------------------------------------------------------------
def conv_out_length(T, ks, stride, pad):
    return (T + 2 * pad - ks) // stride + 1


def im2col_cl(x3d, ks, stride, pad):
    _f32c(x3d, 'x')
    N, T, Cin = x3d.shape
    To = conv_out_length(T, ks, stride, pad)
    col = torch.empty(N * To, Cin * ks, device=x3d.device, dtype=torch.float32)
    check(lib().halo_im2col_cl(ptr(x3d), ptr(col), N, T, Cin, ks, stride, pad, _stream()), 'halo_im2col_cl')
    return col, To


def col2im_cl(dcol, N, T, Cin, ks, stride, pad):
    """The fold: gradient of im2col_cl. dcol [N*T', Cin*ks] -> dx [N, T, Cin]."""
    _f32c(dcol, 'dcol')
    dx = torch.empty(N, T, Cin, device=dcol.device, dtype=torch.float32)
    check(lib().halo_col2im_cl(ptr(dcol), ptr(dx), N, T, Cin, ks, stride, pad, _stream()), 'halo_col2im_cl')
    return dx


def dwconv1d_cl(x3d, weight, bias, stride, pad):
    _f32c(x3d, 'x')
    N, T, Cn = x3d.shape
    ks = weight.shape[-1]
    To = conv_out_length(T, ks, stride, pad)
    y = torch.empty(N, To, Cn, device=x3d.device, dtype=torch.float32)
    check(lib().halo_dwconv1d_cl(ptr(x3d), ptr(weight), ptr(bias), ptr(y), N, T, Cn, ks, stride, pad, _stream()), 'halo_dwconv1d_cl')
    return y


# ---- backward operators of the GPT / transformer training step ----------------------------------------------
def attention_bwd(q, k, v, y, dy, lse, dq, dk, dv, N, heads, head_dim, Tq, Tk, causal=False, key_lengths=None, drop=NO_DROPOUT,
                  stream_id=0):
    """Gradients of attention_fwd written into the caller's dq / dk / dv views (row layouts like q / k / v)."""
    C = heads * head_dim
    _f32c(y, 'y'); _f32c(dy, 'dy')
    if k.stride(0) != v.stride(0) or dk.stride(0) != dv.stride(0):
        raise ValueError('k/v (and dk/dv) must share a row stride')
    delta = torch.empty(N, heads, Tq, device=q.device, dtype=torch.float32)
    check(lib().halo_attention_bwd(ptr(q), q.stride(0), q.stride(0) * Tq, ptr(k), ptr(v), k.stride(0), k.stride(0) * Tk, ptr(y), ptr(dy),
                                   C, C * Tq, ptr(lse), ptr(delta), ptr(dq), dq.stride(0), dq.stride(0) * Tq, ptr(dk), ptr(dv),
                                   dk.stride(0), dk.stride(0) * Tk, N, heads, head_dim, Tq, Tk, int(causal), ptr(key_lengths),
                                   drop.p, drop.seed, stream_id, drop.offset, drop.counter_ptr, _stream()), 'halo_attention_bwd')


def attention_masked(q, k, v, mask):
    """ha/transformer.py:413-430 attend for any boolean mask (True = hidden; broadcastable to (N, heads, T, S)) and any head dimension:
    q (N, heads, T, hd), k / v (N, heads, S, hd) -> (y like q, entropy [N, heads, T]).  The slow general path (halo_attention_masked)."""
    N, H, T, hd = q.shape
    S = k.shape[-2]
    qc, kc, vc = (t.float().contiguous() for t in (q, k, v))
    y = torch.empty_like(qc)
    ent = torch.empty(N, H, T, device=q.device, dtype=torch.float32)
    m8, sn, sh, st = None, 0, 0, 0
    if mask is not None:
        m8 = torch.broadcast_to(mask.to(torch.uint8), (N, H, T, S))
        if m8.stride(3) not in (1, 0) or (m8.stride(3) == 0 and S > 1):
            m8 = m8.contiguous()
        sn, sh, st = m8.stride(0), m8.stride(1), m8.stride(2)
    check(lib().halo_attention_masked(ptr(qc), ptr(kc), ptr(vc), ptr(m8), sn, sh, st, ptr(y), ptr(ent), N, H, T, S, hd, _stream()),
          'halo_attention_masked')
    return y, ent


def attention_fwd_bf16(q, k, v, N, heads, head_dim, Tq, Tk, causal=False, key_lengths=None, drop=NO_DROPOUT, stream_id=0):
    """attention_fwd (training: with lse) that also returns y as row-major bf16 -> (y, lse, y_bf16)."""
    C = heads * head_dim
    dev = q.device
    y = torch.empty(N * Tq, C, device=dev, dtype=torch.float32)
    yb = torch.empty(N * Tq, C, device=dev, dtype=torch.bfloat16)
    lse = torch.empty(N, heads, Tq, device=dev, dtype=torch.float32)
    if key_lengths is not None:
        key_lengths = key_lengths.to(device=dev, dtype=torch.int32).contiguous()
    check(lib().halo_attention_fwd_bf16(ptr(q), q.stride(0), q.stride(0) * Tq, ptr(k), ptr(v), k.stride(0), k.stride(0) * Tk, ptr(y), C,
                                        C * Tq, ptr(yb), C, C * Tq, ptr(lse), N, heads, head_dim, Tq, Tk, int(causal), ptr(key_lengths),
                                        drop.p, drop.seed, stream_id, drop.offset, drop.counter_ptr, _stream()), 'halo_attention_fwd_bf16')
    return y, lse, yb


def attention_fwd_b16(q, k, v, N, heads, head_dim, Tq, Tk, causal=False, want_y=False, want_lse=False):
    """Attention forward from row-major bf16 q / k / v views [N * T, heads * head_dim] (halo_attention_fwd_b16: bf16 arithmetic, head_dim 64)
    -> (y fp32 or None, lse or None, y as row-major bf16)."""
    C = heads * head_dim
    dev = q.device
    if q.dtype != torch.bfloat16 or k.dtype != torch.bfloat16 or v.dtype != torch.bfloat16:
        raise ValueError('attention_fwd_b16: bf16 q, k, v')
    y = torch.empty(N * Tq, C, device=dev, dtype=torch.float32) if want_y else None
    yb = torch.empty(N * Tq, C, device=dev, dtype=torch.bfloat16)
    lse = torch.empty(N, heads, Tq, device=dev, dtype=torch.float32) if want_lse else None
    check(lib().halo_attention_fwd_b16(ptr(q), q.stride(0), q.stride(0) * Tq, ptr(k), ptr(v), k.stride(0), k.stride(0) * Tk, ptr(y), C, C * Tq,
                                       ptr(yb), C, C * Tq, ptr(lse), N, heads, head_dim, Tq, Tk, int(causal), _stream()), 'halo_attention_fwd_b16')
    return y, lse, yb


def attention_bwd_b16(q, k, v, yb, dyb, lse, dq, dk, dv, N, heads, head_dim, Tq, Tk, causal=False):
    """Backward of attention_fwd_b16: every operand row-major bf16 (q, k, v views of the c_attn rows; yb the forward's bf16 output; dyb the
    output gradient), dq / dk / dv bf16 views written in place."""
    for t in (q, k, v, yb, dyb, dq, dk, dv):
        if t.dtype != torch.bfloat16:
            raise ValueError('attention_bwd_b16: bf16 operands')
    delta = torch.empty(N * heads * Tq, device=q.device, dtype=torch.float32)
    check(lib().halo_attention_bwd_b16(ptr(q), q.stride(0), q.stride(0) * Tq, ptr(k), ptr(v), k.stride(0), k.stride(0) * Tk, ptr(yb), ptr(dyb),
                                       yb.stride(0), yb.stride(0) * Tq, ptr(lse), ptr(delta), ptr(dq), ptr(dk), ptr(dv), dq.stride(0),
                                       dq.stride(0) * Tq, N, heads, head_dim, Tq, Tk, int(causal), _stream()), 'halo_attention_bwd_b16')


def attention_bwd_bf16(q, k, v, y, dy, lse, dq, dk, dv, N, heads, head_dim, Tq, Tk, causal=False, key_lengths=None, drop=NO_DROPOUT,
                       stream_id=0):
    """attention_bwd with dq / dk / dv bf16 views of one row stride (the column blocks of a packed [rows, 3C] bf16 buffer)."""
    C = heads * head_dim
    _f32c(y, 'y'); _f32c(dy, 'dy')
    if not (dq.dtype == dk.dtype == dv.dtype == torch.bfloat16) or not (dq.stride(0) == dk.stride(0) == dv.stride(0)) or Tq != Tk:
        raise ValueError('attention_bwd_bf16: bf16 dq / dk / dv with one row stride (self-attention)')
    delta = torch.empty(N, heads, Tq, device=q.device, dtype=torch.float32)
    check(lib().halo_attention_bwd_bf16(ptr(q), q.stride(0), q.stride(0) * Tq, ptr(k), ptr(v), k.stride(0), k.stride(0) * Tk, ptr(y), ptr(dy),
                                        C, C * Tq, ptr(lse), ptr(delta), ptr(dq), ptr(dk), ptr(dv), dq.stride(0), dq.stride(0) * Tq, N, heads,
                                        head_dim, Tq, Tk, int(causal), ptr(key_lengths), drop.p, drop.seed, stream_id, drop.offset,
                                        drop.counter_ptr, _stream()), 'halo_attention_bwd_bf16')


def layernorm_bwd(dy, x2d, weight, dres=None, has_bias=False, eps=1e-5, want_bf16=False):
    """-> (dx = dres + dLN, dweight, dbias or None[, dx as row-major bf16 with ``want_bf16``]).  dy: fp32, or row-major bf16 (the
    input-gradient product's bf16 result)."""
    _f32c(x2d, 'x')
    rows, C = x2d.shape
    if dy.dtype == torch.bfloat16:
        dev = x2d.device
        dx = torch.empty_like(x2d)
        dw = torch.empty(C, device=dev, dtype=torch.float32)
        db = torch.empty(C, device=dev, dtype=torch.float32) if has_bias else None
        ws = torch.empty(lib().halo_layernorm_bwd_workspace_bytes(rows, C), device=dev, dtype=torch.uint8)
        dxb = torch.empty(rows, C, device=dev, dtype=torch.bfloat16) if want_bf16 else None
        check(lib().halo_layernorm_bwd_b16(ptr(dy), ptr(x2d), ptr(weight), ptr(dres), ptr(dx), ptr(dxb), ptr(dw), ptr(db), ptr(ws), rows, C,
                                           eps, _stream()), 'halo_layernorm_bwd_b16')
        return (dx, dw, db, dxb) if want_bf16 else (dx, dw, db)
    _f32c(dy, 'dy')
    dev = x2d.device
    dx = torch.empty_like(x2d)
    dw = torch.empty(C, device=dev, dtype=torch.float32)
    db = torch.empty(C, device=dev, dtype=torch.float32) if has_bias else None
    ws = torch.empty(lib().halo_layernorm_bwd_workspace_bytes(rows, C), device=dev, dtype=torch.uint8)
    if want_bf16:
        dxb = torch.empty(rows, C, device=dev, dtype=torch.bfloat16)
        check(lib().halo_layernorm_bwd_bf16(ptr(dy), ptr(x2d), ptr(weight), ptr(dres), ptr(dx), ptr(dxb), ptr(dw), ptr(db), ptr(ws), rows, C,
                                            eps, _stream()), 'halo_layernorm_bwd_bf16')
        return dx, dw, db, dxb
    check(lib().halo_layernorm_bwd(ptr(dy), ptr(x2d), ptr(weight), ptr(dres), ptr(dx), ptr(dw), ptr(db), ptr(ws), rows, C, eps,
                                   _stream()), 'halo_layernorm_bwd')
    return dx, dw, db


def gelu_fwd(a, exact=False):
    _f32c(a, 'a')
    y = torch.empty_like(a)
    check(lib().halo_gelu_fwd(ptr(a), ptr(y), a.numel(), int(exact), _stream()), 'halo_gelu_fwd')
    return y


def gelu_bwd(dy, a, exact=False, out=None):
    _f32c(dy, 'dy'); _f32c(a, 'a')
    da = out if out is not None else torch.empty_like(a)
    check(lib().halo_gelu_bwd(ptr(dy), ptr(a), ptr(da), a.numel(), int(exact), _stream()), 'halo_gelu_bwd')
    return da


def cross_entropy_fwd_lse(logits2d, targets, ignore_index=0):
    _f32c(logits2d, 'logits')
    tg = _i64c(targets.reshape(-1), 'targets')
    rows, V = logits2d.shape
    loss = torch.empty(rows, device=logits2d.device, dtype=torch.float32)
    lse = torch.empty(rows, device=logits2d.device, dtype=torch.float32)
    check(lib().halo_cross_entropy_fwd_lse(ptr(logits2d), ptr(tg), ptr(loss), ptr(lse), rows, V, V, ignore_index, _stream()),
          'halo_cross_entropy_fwd_lse')
    return loss, lse


def cross_entropy_bwd_(logits2d, targets, lse, grad_rows, ignore_index=0):
    """logits2d <- d loss / d logits in place; grad_rows [rows] (or a 1-element tensor broadcast to all rows)."""
    _f32c(logits2d, 'logits'); _f32c(grad_rows, 'grad')
    tg = _i64c(targets.reshape(-1), 'targets')
    rows, V = logits2d.shape
    stride = 0 if grad_rows.numel() == 1 else 1
    check(lib().halo_cross_entropy_bwd(ptr(logits2d), ptr(tg), ptr(lse), ptr(grad_rows), stride, rows, V, V, ignore_index, _stream()),
          'halo_cross_entropy_bwd')
    return logits2d


def cross_entropy_bwd_images(logits2d, targets, lse, grad_rows, ignore_index=0):
    """d loss / d logits (as cross_entropy_bwd_ computes it) written as the two split images the lm_head's backward
    products read -- (image of [rows][V], image of [V][rows]) -- without materialising it in fp32; logits2d is not modified."""
    _f32c(logits2d, 'logits'); _f32c(grad_rows, 'grad')
    tg = _i64c(targets.reshape(-1), 'targets')
    rows, V = logits2d.shape
    stride = 0 if grad_rows.numel() == 1 else 1
    rm = torch.empty(lib().halo_split_image_bytes(rows, V), device=logits2d.device, dtype=torch.uint8)
    tr = torch.empty(lib().halo_split_image_bytes(V, rows), device=logits2d.device, dtype=torch.uint8)
    check(lib().halo_cross_entropy_bwd_images(ptr(logits2d), ptr(tg), ptr(lse), ptr(grad_rows), stride, rows, V, V, ignore_index,
                                              ptr(rm), ptr(tr), _stream()), 'halo_cross_entropy_bwd_images')
    return rm, tr


def embed_bwd(ids, dx2d, dwte, dwpe, pos0=0, accumulate_wpe=False):
    ids = _i64c(ids, 'input_ids')
    Bn, T = ids.shape
    vocab = dwte.shape[0] if dwte is not None else 1
    check(lib().halo_embed_bwd(ptr(ids), ptr(dx2d), ptr(dwte), ptr(dwpe), Bn, T, dx2d.shape[1], pos0, vocab,
                               int(accumulate_wpe), _stream()), 'halo_embed_bwd')


def dwconv1d_cl_bwd(dy3d, x3d, weight, stride, pad, want_dx=True, has_bias=True):
    """-> (dx or None, dweight [C, ks], dbias or None)"""
    _f32c(dy3d, 'dy'); _f32c(x3d, 'x')
    N, T, Cn = x3d.shape
    ks = weight.shape[-1]
    dev = x3d.device
    dx = torch.empty_like(x3d) if want_dx else None
    dw = torch.empty(Cn, ks, device=dev, dtype=torch.float32)
    db = torch.empty(Cn, device=dev, dtype=torch.float32) if has_bias else None
    ws = torch.empty(lib().halo_dwconv1d_cl_bwd_workspace_bytes(Cn, ks), device=dev, dtype=torch.uint8)
    check(lib().halo_dwconv1d_cl_bwd(ptr(dy3d), ptr(x3d), ptr(weight), ptr(dx), ptr(dw), ptr(db), ptr(ws), N, T, Cn, ks, stride, pad,
                                     _stream()), 'halo_dwconv1d_cl_bwd')
    return dx, dw, db


def add_rows_bcast_(x2d, p2d, T):
    """x2d[n] += p2d[n % T] in place."""
    _f32c(x2d, 'x'); _f32c(p2d, 'p')
    check(lib().halo_add_rows_bcast(ptr(x2d), ptr(p2d), x2d.shape[0], T, x2d.shape[1], _stream()), 'halo_add_rows_bcast')
    return x2d
