"""Linear layers of the attention paths on the HIP GEMMs: y = x W^T (+ b) with the activation / residual
add fused into the GEMM epilogue.  Weights are split (bf16 hi/lo) and tiled once per parameter version;
several nn.Linear weights that read the same input (q|k|v, k|v) are concatenated into one GEMM."""
import torch

from . import _lib, ops


SMALL_M = int(__import__('os').environ.get('HALO_SMALL_M', '64'))


_PAIRS = [False]
# Inside a GraphedTrainStep run (train.py) this holds a number unique to that run: cached weight operands then count as current
# only for the run that built them, so every captured step records the image-building launches (into buffers of the graph's
# own pool) and a replay re-derives the images from the weights as the optimizer left them.  None: eager calls, cache by version.
_RUN_EPOCH = [None]
_EPOCHS = [0]


class graphed_run:
    """One forward + backward that is, or will be, replayed from a HIP graph: weight operand images are rebuilt inside it."""

    def __enter__(self):
        self.prev = _RUN_EPOCH[0]
        _EPOCHS[0] += 1
        _RUN_EPOCH[0] = _EPOCHS[0]

    def __exit__(self, *exc):
        _RUN_EPOCH[0] = self.prev
        return False


def _stamp(weights):
    return (_RUN_EPOCH[0],) + tuple((w._version, w.data_ptr()) for w in weights)


class training_images:
    """Inside this context a weight's forward image is built together with its transposed image (one read, one launch): the
    training forward wraps itself in it because the backward will ask for the transposed one."""

    def __enter__(self):
        self.prev = _PAIRS[0]
        _PAIRS[0] = True

    def __exit__(self, *exc):
        _PAIRS[0] = self.prev
        return False


class WeightImages:
    """Per-module cache of GEMM-ready weight operands, rebuilt only when a weight changes."""

    def __init__(self):
        self._cache = {}

    def _lookup(self, kind, weights, build):
        key = (kind,) + tuple(id(w) for w in weights)
        stamp = _stamp(weights)
        hit = self._cache.get(key)
        if hit is None or hit[0] != stamp:
            hit = (stamp, build())
            self._cache[key] = hit
        return hit[1]

    def dense(self, weights):
        """The (concatenated) fp32 weight [sum N_i, K]; conv weights [N, C, ks] are flattened to [N, C*ks]."""
        if len(weights) == 1 and weights[0].dim() == 2 and weights[0].is_contiguous():
            return weights[0].detach()
        return self._lookup('dense', weights,
                            lambda: torch.cat([w.detach().reshape(w.shape[0], -1) for w in weights], dim=0).contiguous())

    def _pair(self, weights):
        """(image of W, image of W^T) from one read of the (concatenated) weight: a Linear's forward reads the first, the input
        gradient of its backward the second, and both go stale together when the optimizer steps."""
        # bf16 mode writes only the hi part of an image: keep the images of different modes apart
        return self._lookup('pair:' + _lib.get_math_mode(), weights, lambda: ops.image_pair(self.dense(weights).contiguous()))

    def prefetch_pairs(self, weight_sets):
        """Build the (image, transposed image) pairs of several Linears that lack a current one in ceil(n / 6) launches instead of one
        launch each: a training step calls it once with every block's weights, which all went stale at the optimizer step."""
        kind = 'pair:' + _lib.get_math_mode()
        todo = [tuple(ws) for ws in weight_sets if not self._has(kind, tuple(ws))]
        if not todo:
            return
        pairs = ops.image_pairs([self.dense(ws).contiguous() for ws in todo])
        for ws, pr in zip(todo, pairs):
            self._cache[(kind,) + tuple(id(w) for w in ws)] = (_stamp(ws), pr)

    def split(self, weights):
        """The split/tiled image of the (concatenated) weight."""
        if _PAIRS[0] or self._has('pair:' + _lib.get_math_mode(), weights):
            return self._pair(weights)[0]
        return self._lookup('split:' + _lib.get_math_mode(), weights, lambda: ops.split_image(self.dense(weights).contiguous()))

    def split_t(self, weights):
        """The image of W^T (logical [in, sum out]) of the (concatenated) weight: the B operand of dx = dy W."""
        return self._pair(weights)[1]

    def _has(self, kind, weights):
        hit = self._cache.get((kind,) + tuple(id(w) for w in weights))
        return hit is not None and hit[0] == _stamp(weights)


def linear(images, x2d, weights, bias=None, out=None, gelu=False, accumulate=False, drop=ops.NO_DROPOUT, stream_id=0, a_image=None,
           shape=None, residual=None, a_rowmajor=None):
    """x2d [M, K] times the row-concatenation of ``weights`` (each [N_i, K]) -> [M, sum N_i].
    ``weights`` must be the long-lived nn.Parameter objects themselves (the cache is keyed on their identity
    and version), not views made per call.  ``drop``: inverted dropout on the result (before the residual add).
    ``a_image``: the split image of x2d when the caller already has it (normed_image, forward_images), used if the split GEMM
    runs; x2d may then be None with ``shape`` = its shape.  ``residual`` [M, sum N_i]: returns residual + result in a new tensor
    (the residual connection without copying the stream first); ``accumulate`` adds into ``out`` in place.
    ``a_rowmajor``: x2d as the (hi, lo) row-major bf16 pair a previous Linear wrote (ln_linear(..., out_rowmajor=True)); no operand
    image of it is built (inference paths: no dropout)."""
    if isinstance(weights, torch.Tensor):
        weights = (weights,)
    M, K = x2d.shape if x2d is not None else shape
    N = sum(w.shape[0] for w in weights)
    # a handful of rows (one decode step): the operand-image pass would cost more than the product; the exact-f32 kernel
    # reads x and W where they lie
    if a_rowmajor is not None:
        return ops.gemm_split_io(a_rowmajor, images.split(weights), M, N, K, out=out, bias1=bias, accumulate=accumulate, residual=residual)
    if _lib.get_math_mode() != 'f32' and K >= 64 and N >= 64 and M > SMALL_M:
        return ops.gemm_split(a_image if a_image is not None else ops.split_image(x2d), images.split(weights), M, N, K, out=out,
                              bias1=bias, gelu=gelu, accumulate=accumulate, drop=drop, stream_id=stream_id, residual=residual)
    if residual is not None:
        out, accumulate = residual.clone(), True
    return ops.gemm(x2d, images.dense(weights), True, True, M, N, K, out=out, bias1=bias, gelu=gelu, accumulate=accumulate,
                    drop=drop, stream_id=stream_id)


def use_split(M, N, K):
    return _lib.get_math_mode() != 'f32' and K >= 64 and N >= 64


def linear_dx(images, dy2d, weights, out=None, accumulate=False, dy_image=None, shape=None):
    """dx [M, in] = dy [M, sum out] W [sum out, in]  (the input gradient of y = x W^T; ``weights`` as in linear()).
    ``dy_image``: the split image of dy when the caller already has it (grad_images); then dy2d may be None with ``shape`` = dy's."""
    if isinstance(weights, torch.Tensor):
        weights = (weights,)
    M, K = dy2d.shape if dy2d is not None else shape
    N = weights[0][0].numel()
    if use_split(M, N, K):
        return ops.gemm_split(dy_image if dy_image is not None else ops.split_image(dy2d), images.split_t(weights), M, N, K, out=out,
                              accumulate=accumulate)
    return ops.gemm(dy2d, images.dense(weights), True, False, M, N, K, out=out, accumulate=accumulate)


def linear_dw(dy2d, x2d, out=None, accumulate=False, dy_image_t=None, x_image_t=None, shapes=None):
    """dW [out, in] = dy^T [out, M] x [M, in]  (the weight gradient of y = x W^T).  ``dy_image_t`` / ``x_image_t``: the split
    images of dy^T / x^T when the caller already has them; the fp32 tensor may then be None with ``shapes`` = (dy.shape, x.shape)."""
    K, M = dy2d.shape if dy2d is not None else shapes[0]
    N = x2d.shape[1] if x2d is not None else shapes[1][1]
    if use_split(M, N, K):
        return ops.gemm_split(dy_image_t if dy_image_t is not None else ops.split_image(dy2d, transposed=True),
                              x_image_t if x_image_t is not None else ops.split_image(x2d, transposed=True), M, N, K, out=out,
                              accumulate=accumulate)
    return ops.gemm(dy2d, x2d, False, False, M, N, K, out=out, accumulate=accumulate)


def forward_images(x2d, n_out, op=ops.PAIR_COPY):
    """(image of v, image of v^T) for v = op(x2d) [M, K], the input of a Linear with n_out outputs: the forward GEMM reads the first,
    the weight-gradient GEMM of the backward the second, and v itself is never written in fp32.  (None, None) when the split GEMM
    would not run (the caller then materialises v as before)."""
    M, K = x2d.shape
    if use_split(M, n_out, K) and M > SMALL_M and use_split(n_out, K, M):
        return ops.image_pair(x2d, op)
    return None, None


def grad_images(dy2d, n_in, op=ops.PAIR_COPY, x2=None):
    """(image of dy, image of dy^T) from ONE read of dy [M, out] -- what linear_dx and linear_dw of the same Linear (n_in inputs)
    each want -- or (None, None) when the split GEMM would not run for it (the callers then pass the fp32 tensor as before)."""
    M, K = dy2d.shape
    if use_split(M, n_in, K) and use_split(K, n_in, M):
        return ops.image_pair(dy2d, op, x2)
    return None, None


class DropSites:
    """Dropout of one training forward: (p, seed, offset) from the module's DropoutStream plus a running stream id, one per
    dropout site in forward order (the backward replays the same ids).  p == 0 hands out NO_DROPOUT."""
    BASE = 64

    def __init__(self, drop):
        self.drop, self.n = drop, 0

    def next(self):
        self.n += 1
        return self.drop, self.BASE + self.n - 1


NO_SITES = DropSites(ops.NO_DROPOUT)


def drop_rows(x2d, site):
    """x2d * mask of a dropout site (elementwise kernel); the same call gives the backward of that site."""
    drop, sid = site
    return ops.dropout_fwd(x2d, drop, sid) if drop.p > 0 else x2d


def rowmajor_ok(M, N, K):
    """Whether a Linear [M, K] -> [M, N] can hand its result on as row-major bf16 to a Linear that contracts over N (gemm_split_io)."""
    # halo_gemm_split_io runs without split-K: both products must fill the chip with whole-K tiles (the attention-ASR blocks at
    # 1280 rows do not: 40 tiles, 1.2 -> 1.8 ms per encoder pass when forced)
    tiles = lambda r, c: ((r + 127) // 128) * ((c + 127) // 128)
    return use_split(M, N, K) and M > SMALL_M and K % 32 == 0 and N % 32 == 0 and tiles(M, N) >= 200 and tiles(M, K) >= 200


def ln_linear(images, x2d, ln_weight, ln_bias, weights, bias=None, gelu=False, want_normed=False, eps=1e-5, out_rowmajor=False):
    """linear(layer_norm(x2d), weights) with the normalisation written straight into the GEMM's operand image when the split
    GEMM will run (saves the pass that re-reads the normalised rows to split them).  -> (out, normed fp32 rows or None).
    ``out_rowmajor`` (callers check rowmajor_ok first): out is the (hi, lo) row-major bf16 pair for linear(..., a_rowmajor=)."""
    if isinstance(weights, torch.Tensor):
        weights = (weights,)
    M, K = x2d.shape
    N = sum(w.shape[0] for w in weights)
    if use_split(M, N, K) and M > SMALL_M and K % 32 == 0:
        a_img, h = ops.layernorm_image(x2d, ln_weight, ln_bias, eps, want_y=want_normed)
        if out_rowmajor:
            return ops.gemm_split_io(a_img, images.split(weights), M, N, K, out_rowmajor=True, bias1=bias, gelu=gelu), h
        return ops.gemm_split(a_img, images.split(weights), M, N, K, bias1=bias, gelu=gelu), h
    h = ops.layernorm_fwd(x2d, ln_weight, ln_bias, eps)
    return linear(images, h, weights, bias=bias, gelu=gelu), (h if want_normed else None)


def normed_image(x2d, ln_weight, ln_bias=None, n_out=64, want_normed=True, eps=1e-5):
    """layer_norm(x2d) for Linear layers with n_out outputs: (normed fp32 rows or None, their split image or None).  The image
    comes straight out of the LayerNorm kernel when the split GEMM will consume it; otherwise only the fp32 rows exist."""
    M, K = x2d.shape
    if use_split(M, n_out, K) and M > SMALL_M and K % 32 == 0:
        img, h = ops.layernorm_image(x2d, ln_weight, ln_bias, eps, want_y=want_normed)
        return h, img
    return ops.layernorm_fwd(x2d, ln_weight, ln_bias, eps), None
