"""Philox4x32-10 in numpy -- the oracle's copy of the dropout stream used by the HIP kernels.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference draws its dropout masks from torch's global RNG (``nn.Dropout`` in
rnn.py:8, ``nn.LSTM(dropout=0.2)`` rnn.py:11, recognizer.py:41); those streams cannot be
reproduced bit-wise on another device ("Dropout RNG").  The HIP path
therefore defines its own counter-based stream (haloop_amd/csrc/philox.h) and this file
restates it so that training-mode parity tests can feed the *same* masks to the CPU
restatement.

Mask definition (must match csrc/philox.h):
    ctr = (lo32(e >> 2), hi32(e >> 2), stream_id, offset)      e = flat element index
    key = (lo32(seed), hi32(seed))
    r   = philox4x32_10(ctr, key)[e & 3]
    keep(e) = r >= uint32(p * 2**32)          scale = float32(1) / (float32(1) - float32(p))
"""
import numpy as np

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = 0x9E3779B9
_W1 = 0xBB67AE85
_MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over numpy uint32 arrays c0..c3; k0, k1 python ints. Returns 4 uint32 arrays."""
    c0 = c0.astype(np.uint64); c1 = c1.astype(np.uint64)
    c2 = c2.astype(np.uint64); c3 = c3.astype(np.uint64)
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = _M0 * c0
        p1 = _M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & _MASK32
        hi1, lo1 = p1 >> np.uint64(32), p1 & _MASK32
        n0 = hi1 ^ c1 ^ np.uint64(k0)
        n2 = hi0 ^ c3 ^ np.uint64(k1)
        c0, c1, c2, c3 = n0, lo1, n2, lo0
        k0 = (k0 + _W0) & 0xFFFFFFFF
        k1 = (k1 + _W1) & 0xFFFFFFFF
    return (c0.astype(np.uint32), c1.astype(np.uint32), c2.astype(np.uint32), c3.astype(np.uint32))


def dropout_threshold(p):
    return np.uint32(min(int(np.float32(p) * np.float64(4294967296.0)), 0xFFFFFFFF))


def dropout_scale(p):
    return np.float32(1.0) / (np.float32(1.0) - np.float32(p))


def dropout_mask(n, p, seed, stream_id, offset):
    """float32 array of length n: 0 where dropped, 1/(1-p) where kept."""
    if p <= 0.0:
        return np.ones(n, dtype=np.float32)
    e = np.arange(n, dtype=np.uint64)
    q = e >> np.uint64(2)
    c0 = (q & _MASK32).astype(np.uint32)
    c1 = (q >> np.uint64(32)).astype(np.uint32)
    c2 = np.full(n, stream_id, dtype=np.uint32)
    c3 = np.full(n, offset, dtype=np.uint32)
    r = philox4x32_10(c0, c1, c2, c3, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    sel = (e & np.uint64(3)).astype(np.int64)
    rr = np.choose(sel, r)
    keep = rr >= dropout_threshold(p)
    return np.where(keep, dropout_scale(p), np.float32(0.0)).astype(np.float32)


def attention_dropout_mask(N, H, Tq, Tk, p, seed, stream_id, offset):
    """Multipliers [N, H, Tq, Tk] of the attention-probability dropout of halo_attention_fwd (include/halo.h):
    probability (n, h, i, j) uses stream element ((((n*H + h)*Tq + i) * ceil(Tk/64) + j/64) * 64 + 4*(j%16) + (j%64)/16."""
    KT = (Tk + 63) // 64
    flat = dropout_mask(N * H * Tq * KT * 64, p, seed, stream_id, offset).reshape(N, H, Tq, KT, 64)
    j = np.arange(Tk)
    return flat[:, :, :, j // 64, 4 * (j % 16) + (j % 64) // 16]
