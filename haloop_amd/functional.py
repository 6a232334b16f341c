"""torch.autograd glue: each Function is one forward/backward pair of C-ABI calls."""
import torch

from . import ops
from .ops import NO_DROPOUT


def _lstm_lists(lstm_params, L):
    return ([lstm_params[4 * l + 0] for l in range(L)], [lstm_params[4 * l + 1] for l in range(L)],
            [lstm_params[4 * l + 2] for l in range(L)], [lstm_params[4 * l + 3] for l in range(L)])


class _Encoder(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, conv_w, conv_b, drop, L, *lstm_params):
        B, T, F = x.shape
        w_ih, w_hh, b_ih, b_hh = _lstm_lists(lstm_params, L)
        H = w_hh[0].shape[1]
        x = x.contiguous()
        y_sub, col = ops.subsample_fwd(x, conv_w, conv_b, drop)
        Tp = y_sub.shape[0]
        feats = torch.empty(B, Tp, H, device=x.device, dtype=torch.float32)
        _, _, _, reserve = ops.lstm_fwd(y_sub, w_ih, w_hh, b_ih, b_hh, y=feats, y_strides=(H, Tp * H), y_relu=True,
                                        drop=drop, expect_backward=any(ctx.needs_input_grad))
        ctx.save_for_backward(y_sub, col, reserve, *lstm_params)
        ctx.meta = (B, T, F, conv_w.shape[0], L, H, Tp, drop)
        return feats

    @staticmethod
    def backward(ctx, dfeats):
        B, T, F, Cc, L, H, Tp, drop = ctx.meta
        y_sub, col, reserve, *lstm_params = ctx.saved_tensors
        w_ih, w_hh, _, _ = _lstm_lists(lstm_params, L)
        dfeats = dfeats.contiguous()
        dy_sub, g = ops.lstm_bwd(y_sub, w_ih, w_hh, dfeats, (H, Tp * H), True, reserve, want_dx=True, drop=drop)
        dw, db = ops.subsample_bwd(dy_sub, y_sub, col, B, T, F, Cc, drop.p)
        lstm_grads = []
        for l in range(L):
            lstm_grads += [g['dw_ih'][l], g['dw_hh'][l], g['db_ih'][l], g['db_hh'][l]]
        return (None, dw, db, None, None, *lstm_grads)


def encoder_forward(x, conv_w, conv_b, lstm_params, num_layers, drop=NO_DROPOUT):
    return _Encoder.apply(x, conv_w, conv_b, drop, num_layers, *lstm_params)


class _LSTM(torch.autograd.Function):
    """Time-major LSTM with explicit initial/final state (ha/rnn.py:50: nn.LSTM(emb, state))."""

    @staticmethod
    def forward(ctx, x_tm, h0, c0, drop, L, *lstm_params):
        w_ih, w_hh, b_ih, b_hh = _lstm_lists(lstm_params, L)
        x_tm = x_tm.contiguous()
        y, hn, cn, reserve = ops.lstm_fwd(x_tm, w_ih, w_hh, b_ih, b_hh, h0=h0.contiguous(), c0=c0.contiguous(),
                                          want_state=True, drop=drop, expect_backward=any(ctx.needs_input_grad))
        ctx.save_for_backward(x_tm, reserve, *lstm_params)
        ctx.meta = (L, w_hh[0].shape[1], drop, x_tm.requires_grad)
        return y, hn, cn

    @staticmethod
    def backward(ctx, dy, dhn, dcn):
        L, H, drop, want_dx = ctx.meta
        x_tm, reserve, *lstm_params = ctx.saved_tensors
        w_ih, w_hh, _, _ = _lstm_lists(lstm_params, L)
        T, B, _ = x_tm.shape
        dy = dy.contiguous()
        dx, g = ops.lstm_bwd(x_tm, w_ih, w_hh, dy, (B * H, H), False, reserve, dhn=dhn.contiguous(),
                             dcn=dcn.contiguous(), want_dx=True, drop=drop)
        lstm_grads = []
        for l in range(L):
            lstm_grads += [g['dw_ih'][l], g['dw_hh'][l], g['db_ih'][l], g['db_hh'][l]]
        # gradients w.r.t. the incoming state are not produced: callers detach it (ha/rnn.py:75-77)
        return (dx, None, None, None, None, *lstm_grads)


def lstm_forward(x_tm, state, lstm_params, num_layers, drop=NO_DROPOUT):
    h0, c0 = state
    return _LSTM.apply(x_tm, h0, c0, drop, num_layers, *lstm_params)


class _Linear(torch.autograd.Function):
    """y = x W^T + b on the f32 MFMA GEMM (ha/recognizer.py:45, ha/rnn.py:51)."""

    @staticmethod
    def forward(ctx, x2d, w, b):
        x2d = x2d.contiguous()
        M, K = x2d.shape
        N = w.shape[0]
        y = ops.gemm(x2d, w, True, True, M, N, K, bias1=b)
        ctx.save_for_backward(x2d, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        x2d, w = ctx.saved_tensors
        dy = dy.contiguous()
        M, K = x2d.shape
        N = w.shape[0]
        dx = ops.gemm(dy, w, True, False, M, K, N)          # [M,N] x [N,K]
        dw = ops.gemm(dy, x2d, False, False, N, K, M)       # dy^T [N,M] x x [M,K]
        db = ops.colsum(dy)
        return dx, dw, db


class _Embedding(torch.autograd.Function):
    """nn.Embedding lookup (ha/rnn.py:38,45) on the HIP gather; the backward scatter-adds rows with float atomics."""

    @staticmethod
    def forward(ctx, ids, weight):
        flat = ids.reshape(1, -1)
        ctx.save_for_backward(flat)
        ctx.shape = weight.shape
        return ops.embed_fwd(flat, weight, None).view(*ids.shape, weight.shape[1])

    @staticmethod
    def backward(ctx, dy):
        (flat,) = ctx.saved_tensors
        dw = torch.zeros(ctx.shape, device=dy.device, dtype=torch.float32)
        ops.embed_bwd(flat, dy.reshape(-1, ctx.shape[1]).contiguous().float(), dw, None)
        return None, dw


def embedding(ids, weight):
    return _Embedding.apply(ids, weight)


def linear(x, w, b):
    shp = x.shape
    y = _Linear.apply(x.reshape(-1, shp[-1]), w, b)
    return y.view(*shp[:-1], w.shape[0])


class _DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, drop, stream_id):
        x = x.contiguous()
        ctx.meta = (drop, stream_id)
        return ops.dropout_fwd(x, drop, stream_id)

    @staticmethod
    def backward(ctx, dy):
        drop, stream_id = ctx.meta
        return ops.dropout_fwd(dy.contiguous(), drop, stream_id), None, None


def dropout(x, drop, stream_id):
    if drop.p <= 0.0:
        return x
    return _DropoutFn.apply(x, drop, stream_id)


class _LogSoftmax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x2d):
        y = ops.log_softmax_fwd(x2d.contiguous())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return ops.log_softmax_bwd(dy.contiguous(), y)


def log_softmax(x):
    shp = x.shape
    return _LogSoftmax.apply(x.reshape(-1, shp[-1])).view(shp)


class _CTC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, lp, targets, input_lengths, target_lengths, time_major):
        nll, alpha, saved = ops.ctc_fwd(lp, time_major, targets, input_lengths, target_lengths, flags=0)
        ctx.save_for_backward(lp, alpha, nll, saved[0], saved[2], *([saved[1]] if saved[1] is not None else []))
        ctx.time_major = time_major
        return nll

    @staticmethod
    def backward(ctx, grad_out):
        lp, alpha, nll, targets, tl, *rest = ctx.saved_tensors
        il = rest[0] if rest else None
        grad = ops.ctc_bwd(lp, ctx.time_major, (targets, il, tl), alpha, nll, grad_out)
        return grad, None, None, None, None


def ctc_loss(log_probs, targets, input_lengths, target_lengths, reduction='mean', time_major=True):
    """F.ctc_loss(blank=0, zero_infinity=False) on HIP (ha/recognizer.py:71).

    log_probs [T,N,C] (time_major) or [N,T,C]; any batch/time strides, so the caller's
    ``permute(1, 0, 2)`` view is consumed without a copy."""
    nll = _CTC.apply(log_probs, targets, input_lengths, target_lengths, time_major)
    if reduction == 'none':
        return nll
    if reduction == 'sum':
        return nll.sum()
    if reduction == 'mean':
        tl = target_lengths.to(device=nll.device, dtype=nll.dtype).clamp_min(1)
        return (nll / tl).mean()
    raise ValueError(f'unknown reduction {reduction!r}')
