"""Randomised shape sweep of round 5's kernels behind GUARD BANDS (run once on the GPU box; not part of the test suite):
halo_gemm_rows (fp32 / + residual / bf16 results, forced and chosen tile widths), halo_gemm_rows_ce, halo_attention_fwd_b16 /
halo_attention_bwd_b16, halo_gemm_tn_bf16_group on random ragged shapes, every output allocated inside a larger buffer filled with a sentinel that must survive,
results against fp64 references of the same bf16 operand values.

    python tools/fuzz_round5.py [cases] [seed]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from haloop_amd import _lib, ops                                                       # noqa: E402

DEV = 'cuda'
GUARD = 4096                                                                            # elements on either side
SENT = 12345.0


class Guarded:
    """A tensor of `shape` (row-major) in the middle of a sentinel-filled buffer."""

    def __init__(self, shape, dtype):
        n = 1
        for s in shape:
            n *= s
        self.buf = torch.full((n + 2 * GUARD,), SENT, device=DEV, dtype=dtype)
        self.t = self.buf[GUARD:GUARD + n].view(*shape)
        self.n = n

    def intact(self):
        return bool((self.buf[:GUARD] == SENT).all() and (self.buf[GUARD + self.n:] == SENT).all())


def ptr(t):
    return t.data_ptr() if t is not None else None


def fuzz_gemm_rows(g, cases):
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    bad = 0
    for c in range(cases):
        M = int(torch.randint(1, 2500, (1,), generator=g))
        N = int(torch.randint(1, 300, (1,), generator=g)) * 8
        K = int(torch.randint(1, 48, (1,), generator=g)) * 32
        tn = [0, 3, 6, 9][int(torch.randint(0, 4, (1,), generator=g))]
        if tn:
            os.environ['HALO_GEMM_ROWS_TN'] = str(tn)
        else:
            os.environ.pop('HALO_GEMM_ROWS_TN', None)
        if not ops.gemm_rows_supported(M, N, K):
            continue
        a = torch.randn(M, K, generator=g).to(DEV).bfloat16()
        w = torch.randn(N, K, generator=g).to(DEV)
        img = ops.split_image(w)
        want = a.double() @ w.bfloat16().double().t()
        tol = 2e-6 * K ** 0.5 * 16 + 1e-5
        r = torch.randn(M, N, generator=g).to(DEV)
        out32, outr, outb = Guarded((M, N), torch.float32), Guarded((M, N), torch.float32), Guarded((M, N), torch.bfloat16)
        _lib.check(L.halo_gemm_rows(None, ptr(a), K, ptr(img), M, N, K, ptr(out32.t), N, None, 0, None, 0, st), 'gemm_rows f32')
        _lib.check(L.halo_gemm_rows(None, ptr(a), K, ptr(img), M, N, K, ptr(outr.t), N, ptr(r), N, None, 0, st), 'gemm_rows resid')
        _lib.check(L.halo_gemm_rows(None, ptr(a), K, ptr(img), M, N, K, None, 0, None, 0, ptr(outb.t), N, st), 'gemm_rows bf16')
        torch.cuda.synchronize()
        e32 = (out32.t.double() - want).abs().max().item()
        er = (outr.t.double() - want - r.double()).abs().max().item()
        okb = torch.equal(outb.t, out32.t.bfloat16())
        ok = e32 <= tol and er <= tol + 1e-6 and okb and out32.intact() and outr.intact() and outb.intact()
        if not ok:
            bad += 1
            print(f'gemm_rows MISMATCH M {M} N {N} K {K} tn {tn}: err {e32:.3e} / {er:.3e} (tol {tol:.3e}) bf16 {okb} guards '
                  f'{out32.intact()} {outr.intact()} {outb.intact()}', flush=True)
        # the cross-entropy epilogue on the same operands
        tgt = torch.randint(0, N, (M,), generator=g).to(DEV)
        ws = torch.empty(L.halo_gemm_rows_ce_workspace_bytes(M, N), device=DEV, dtype=torch.uint8)
        loss, lse, logit = Guarded((M,), torch.float32), Guarded((M,), torch.float32), Guarded((M, N), torch.bfloat16)
        _lib.check(L.halo_gemm_rows_ce(None, ptr(a), K, ptr(img), M, N, K, ptr(tgt), -100, ptr(ws), ptr(loss.t), ptr(lse.t), ptr(logit.t), N, st),
                   'gemm_rows_ce')
        torch.cuda.synchronize()
        lse_ref = torch.logsumexp(want, dim=1)
        loss_ref = lse_ref - want[torch.arange(M), tgt]
        el = (lse.t.double() - lse_ref).abs().max().item()
        eo = (loss.t.double() - loss_ref).abs().max().item()
        okl = torch.equal(logit.t, out32.t.bfloat16())
        ok = el <= tol + 2e-5 and eo <= 2 * tol + 2e-5 and okl and loss.intact() and lse.intact() and logit.intact()
        if not ok:
            bad += 1
            print(f'gemm_rows_ce MISMATCH M {M} V {N} K {K} tn {tn}: lse {el:.3e} loss {eo:.3e} logits {okl} guards {loss.intact()} '
                  f'{lse.intact()} {logit.intact()}', flush=True)
    os.environ.pop('HALO_GEMM_ROWS_TN', None)
    return bad


def fuzz_attention(g, cases):
    bad = 0
    for c in range(cases):
        N = int(torch.randint(1, 4, (1,), generator=g))
        heads = int(torch.randint(1, 5, (1,), generator=g))
        T = int(torch.randint(1, 400, (1,), generator=g))
        causal = bool(torch.randint(0, 2, (1,), generator=g))
        hd, C = 64, heads * 64
        qkv = Guarded((N * T, 3 * C), torch.bfloat16)
        qkv.t.copy_((torch.randn(N * T, 3 * C, generator=g) * 0.7).to(DEV).bfloat16())
        q, k, v = qkv.t[:, :C], qkv.t[:, C:2 * C], qkv.t[:, 2 * C:]
        y, lse, yb = ops.attention_fwd_b16(q, k, v, N, heads, hd, T, T, causal=causal, want_y=True, want_lse=True)
        qd, kd, vd = (t.double().view(N, T, heads, hd).transpose(1, 2).detach().requires_grad_(True) for t in (q, k, v))
        sc = qd @ kd.transpose(-1, -2) / hd ** 0.5
        if causal:
            sc = sc.masked_fill(torch.ones(T, T, device=DEV, dtype=torch.bool).triu(1), float('-inf'))
        yr = (sc.softmax(-1) @ vd)
        yref = yr.transpose(1, 2).reshape(N * T, C)
        ef = (y.double() - yref).abs().max().item()
        dyb = (torch.randn(N * T, C, generator=g) * 0.5).to(DEV).bfloat16()
        yref.backward(dyb.double())
        d = Guarded((N * T, 3 * C), torch.bfloat16)
        ops.attention_bwd_b16(q, k, v, yb, dyb, lse, d.t[:, :C], d.t[:, C:2 * C], d.t[:, 2 * C:], N, heads, hd, T, T, causal=causal)
        torch.cuda.synchronize()
        cosmin = 1.0
        for got, ref in ((d.t[:, :C], qd.grad), (d.t[:, C:2 * C], kd.grad), (d.t[:, 2 * C:], vd.grad)):
            ref2 = ref.transpose(1, 2).reshape(N * T, C)
            den = (got.double().norm() * ref2.norm()).item()
            cosmin = min(cosmin, ((got.double() * ref2).sum().item() / den) if den > 0 else 1.0)
        ok = ef <= 2e-2 and cosmin >= 0.999 and d.intact() and qkv.intact() and bool(torch.isfinite(d.t.float()).all())
        if not ok:
            bad += 1
            print(f'attention MISMATCH N {N} heads {heads} T {T} causal {causal}: fwd err {ef:.3e} min cosine {cosmin:.5f} guards {d.intact()} '
                  f'{qkv.intact()}', flush=True)
    return bad


def fuzz_tn_group(g, cases):
    """halo_gemm_tn_bf16_group: 1-4 products a_i [K, M_i]^T b_i [K, N_i] of one K per launch, results inside guard bands."""
    import ctypes as C
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    bad = 0
    for c in range(cases):
        n = int(torch.randint(1, 5, (1,), generator=g))
        K = int(torch.randint(1, 40, (1,), generator=g)) * 32
        Ms = [int(torch.randint(1, 60 if c % 8 else 700, (1,), generator=g)) * 8 for _ in range(n)]      # (every eighth group: many tiles -> K-slices)
        Ns = [int(torch.randint(1, 60, (1,), generator=g)) * 8 for _ in range(n)]
        As = [torch.randn(K, m, generator=g).to(DEV).bfloat16() for m in Ms]
        Bs = [torch.randn(K, nn, generator=g).to(DEV).bfloat16() for nn in Ns]
        Cs = [Guarded((m, nn), torch.float32) for m, nn in zip(Ms, Ns)]
        vp = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
        rc = L.halo_gemm_tn_bf16_group(n, vp(As), (C.c_long * n)(*Ms), vp(Bs), (C.c_long * n)(*Ns), (C.c_int * n)(*Ms), (C.c_int * n)(*Ns), K,
                                       vp([c_.t for c_ in Cs]), (C.c_int * n)(*Ns), 0, st)
        _lib.check(rc, 'halo_gemm_tn_bf16_group')
        # the same products on the 256-row tiles of csrc/gemm_tn_rows.hip (a random tile width; under-filled rounds as K-slices)
        Rs = [Guarded((m, nn), torch.float32) for m, nn in zip(Ms, Ns)]
        os.environ['HALO_GEMM_TN_ROWS_TN'] = ['4', '8'][int(torch.randint(0, 2, (1,), generator=g))]
        rc = L.halo_gemm_tn_rows_group(n, vp(As), (C.c_long * n)(*Ms), vp(Bs), (C.c_long * n)(*Ns), (C.c_int * n)(*Ms), (C.c_int * n)(*Ns), K,
                                       vp([c_.t for c_ in Rs]), (C.c_long * n)(*Ns), st)
        _lib.check(rc, 'halo_gemm_tn_rows_group')
        torch.cuda.synchronize()
        for a, b, cc in zip(As, Bs, Rs):
            want = a.double().t() @ b.double()
            err = (cc.t.double() - want).abs().max().item()
            if err > 2e-6 * K ** 0.5 * 16 + 1e-5 or not cc.intact():
                bad += 1
                print(f'gemm_tn_rows MISMATCH n {n} K {K} M {a.shape[1]} N {b.shape[1]} tn {os.environ["HALO_GEMM_TN_ROWS_TN"]}: err {err:.3e} guard {cc.intact()}', flush=True)
        for a, b, cc in zip(As, Bs, Cs):
            want = a.double().t() @ b.double()
            err = (cc.t.double() - want).abs().max().item()
            tol = 2e-6 * K ** 0.5 * 16 + 1e-5
            if err > tol or not cc.intact():
                bad += 1
                print(f'gemm_tn_group MISMATCH n {n} K {K} M {a.shape[1]} N {b.shape[1]}: err {err:.3e} (tol {tol:.3e}) guard {cc.intact()}', flush=True)
    os.environ.pop('HALO_GEMM_TN_ROWS_TN', None)
    return bad


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    _lib.lib(); _lib.lend_scratch(256 << 20)
    _lib.set_math_mode('bf16')
    g = torch.Generator().manual_seed(seed)
    b1 = fuzz_gemm_rows(g, cases)
    print(f'gemm_rows / gemm_rows_ce: {cases} random shapes, {b1} failures', flush=True)
    b2 = fuzz_attention(g, max(cases // 2, 1))
    print(f'attention fwd / bwd from bf16 rows: {max(cases // 2, 1)} random shapes, {b2} failures', flush=True)
    b3 = fuzz_tn_group(g, max(cases // 2, 1))
    print(f'grouped TN weight-gradient products: {max(cases // 2, 1)} random groups, {b3} failures', flush=True)
    raise SystemExit(1 if b1 + b2 + b3 else 0)


if __name__ == '__main__':
    main()
