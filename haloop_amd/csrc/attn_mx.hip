// Softmax attention forward / backward on the bf16 matrix cores of gfx950 (v_mfma_f32_16x16x32_bf16), same
// masks, outputs and 64-key tiles as the exact-f32 kernels of attn.hip.  Operands are fp32 in HBM; on their way into
// LDS / registers they are split x = hi + lo (two bf16) and every product is accumulated in fp32 as
//     a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi        (PASSES = 3, ~2^-16 relative error per product)
// or just a_hi*b_hi (PASSES = 1, plain bf16 operands).  Three 16-cycle bf16 MFMAs replace eight 32-cycle f32
// MFMAs per 32-deep product: 5.3x the matrix rate at fp32-grade accuracy.
//
// LDS images: a 64-row tile is kept ROW-major [row][HD] bf16 (hi image, then lo image), row stride HD*2+16 bytes.
//   - products that contract over the row's own axis (Q.K^T, dO.V^T, K.Q^T, V.dO^T) read a B fragment as one
//     ds_read_b128 per lane (8 consecutive dims of one row);
//   - products that contract over the ROW index (P.V, dS.K, P^T.dO, dS^T.Q) read the same image with
//     ds_read_b64_tr_b16, the hardware transpose read: no second, transposed copy of V / K / Q / dO is staged.
// Score tiles are computed transposed so that the lane-fixed index of the MFMA D layout is the row a workgroup owns
// (the query in the forward / dQ sweeps, the key in the dK/dV sweep): softmax statistics are lane scalars and the
// probabilities / score gradients are, as they sit in registers, the B operand of the second product.
#include "halo_common.h"
#include "attn_args.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;

template <int HD, int PASSES = 3>
struct Img {
    // row stride in bytes (multiple of 16).  HD = 64: 160 -- with the hardware's lane groups (MI355X_MICROARCH.md, LDS) the 16 lanes of a
    // ds_read_b128 group land on 16 different 16-byte slots (row * 10 + chunk mod 16) and the 32 lanes of a ds_read_b64_tr_b16 group
    // on 64 different banks (row * 40 mod 64 in steps of 8); 144 left 40 % of the LDS cycles as bank conflicts
    // (single-pass kernels only: the three-pass dK/dV sweep holds four hi|lo image pairs and loses a workgroup per CU to the wider rows)
    static constexpr int ROWB = (HD == 64 && PASSES == 1) ? 160 : HD * 2 + 16;
    static constexpr int BYTES = 64 * ROWB;            // one image (hi or lo) of a 64-row tile
    static constexpr int UNITS = 64 * (HD / 4) / 256;  // float4 units per thread and tile
    static constexpr int KSTEPS = HD / 32;             // 32-deep MFMA steps across the head dimension
};

__device__ __forceinline__ void split8(const float (&x)[8], bf16x8 &hi, bf16x8 &lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 h = (__bf16)x[j];
        hi[j] = h;
        lo[j] = (__bf16)(x[j] - (float)h);
    }
}

template <int PASSES>
__device__ __forceinline__ f32x4 mma(f32x4 acc, const bf16x8 &ah, const bf16x8 &al, const bf16x8 &bh, const bf16x8 &bl) {
    if (PASSES == 3) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);
    }
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
}

// 8 fp32 values of one row (16-byte aligned) -> scaled hi/lo fragments
__device__ __forceinline__ void load_split8(const float *p, float scale, bf16x8 &hi, bf16x8 &lo) {
    const f32x4 a = *reinterpret_cast<const f32x4 *>(p), b = *reinterpret_cast<const f32x4 *>(p + 4);
    const float x[8] = {a[0] * scale, a[1] * scale, a[2] * scale, a[3] * scale, b[0] * scale, b[1] * scale, b[2] * scale, b[3] * scale};
    split8(x, hi, lo);
}

template <int HD>
__device__ __forceinline__ void fetch_tile(f32x4 *reg, const float *base, long rs, int row0, int n_rows) {
#pragma unroll
    for (int i = 0; i < Img<HD>::UNITS; ++i) {
        const int u = threadIdx.x + 256 * i;
        const int row = min(row0 + u / (HD / 4), n_rows - 1), d4 = (u % (HD / 4)) * 4;
        reg[i] = *reinterpret_cast<const f32x4 *>(base + (long)row * rs + d4);
    }
}

// the same units from precomputed element offsets (row * rs + d4 of this thread's units inside a 64-row tile) for tiles that lie
// wholly inside the operand: one 64-bit add per 16-byte load instead of a clamp, a multiply and an add chain
template <int HD>
__device__ __forceinline__ void fetch_tile_inner(f32x4 *reg, const float *tile_base, const int *uoff) {
#pragma unroll
    for (int i = 0; i < Img<HD>::UNITS; ++i) reg[i] = *reinterpret_cast<const f32x4 *>(tile_base + uoff[i]);
}

// registers (fetch_tile layout) -> hi | lo row-major bf16 images
template <int HD, int PASSES>
__device__ __forceinline__ void stage_tile(char *img, const f32x4 *reg) {
#pragma unroll
    for (int i = 0; i < Img<HD, PASSES>::UNITS; ++i) {
        const int u = threadIdx.x + 256 * i;
        const int row = u / (HD / 4), d4 = (u % (HD / 4)) * 4;
        bf16x4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            h[e] = (__bf16)reg[i][e];
            l[e] = (__bf16)(reg[i][e] - (float)h[e]);
        }
        *reinterpret_cast<bf16x4 *>(img + row * Img<HD, PASSES>::ROWB + d4 * 2) = h;
        if (PASSES == 3) *reinterpret_cast<bf16x4 *>(img + Img<HD, PASSES>::BYTES + row * Img<HD, PASSES>::ROWB + d4 * 2) = l;
    }
}

// B (or A) fragment whose k runs along the row: 8 consecutive elements of `row` starting at element k0
template <int HD, int PASSES>
__device__ __forceinline__ bf16x8 row_frag(const char *img, int row, int k0) {
    return *reinterpret_cast<const bf16x8 *>(img + row * Img<HD, PASSES>::ROWB + k0 * 2);
}

// ---- forward ---------------------------------------------------------------------------------------------
// The score tile is computed TRANSPOSED, S^T = K Q^T, so that in the MFMA D layout a lane holds 16 keys of ONE query
// (query = lane & 15): the running max / sum / rescale are lane-local scalars, the only cross-lane traffic per tile
// is one max over the four 16-lane rows (two v_permlane*_swap), and the probabilities are already laid out as the B
// operand of O^T = V^T P^T -- no trip through LDS.  The contraction slot (lq, e) of a 32-deep P.V step stands for key
// 16*(2kk + e/4) + 4*lq + e%4; the transpose read fetches V's rows in the same order.

// A fragment of V^T (or any row-major image read across its rows): rows rowA .. rowA+3 and rowB .. rowB+3 of column
// col0 + (lane & 15), through the hardware transpose read.  ds_read_b64_tr_b16: within a 16-lane group, lane 4q+p
// addresses row q, columns 4p..4p+3 of a 4 x 16 block and lane i receives column i of the 4 rows.  Needs EXEC all
// ones: call it from wave-uniform code only.
template <int HD, int PASSES>
__device__ __forceinline__ bf16x8 tr_frag2(const char *img, int rowA, int rowB, int col0, int lr) {
    const int off = (lr >> 2) * Img<HD, PASSES>::ROWB + (col0 + 4 * (lr & 3)) * 2;
    const bf16x4 x0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(img + rowA * Img<HD, PASSES>::ROWB + off));
    const bf16x4 x1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(img + rowB * Img<HD, PASSES>::ROWB + off));
    return bf16x8{x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
}

// QB: 16-query blocks per wave (a workgroup covers 64 * QB queries).  With QB = 2 every K fragment and every V^T fragment read
// from LDS feeds two MFMAs, and a staged tile (and its two barriers) serves twice the queries.
// DROP: the Philox dropout of the probabilities is compiled in only for the launches that use it
template <int HD, int PASSES, int QB, bool DROP>
__global__ __launch_bounds__(256, QB == 1 ? 3 : 2) void attention_fwd_mx_kernel(AttnArgs a) {
    using I = Img<HD, PASSES>;
    constexpr int WQ = 16 * QB, TQ = 64 * QB;                  // queries per wave / per workgroup
    __shared__ __attribute__((aligned(16))) char Kimg[(PASSES == 3 ? 2 : 1) * I::BYTES];      // hi | lo images (hi only in single-pass mode)
    __shared__ __attribute__((aligned(16))) char Vimg[(PASSES == 3 ? 2 : 1) * I::BYTES];
    // One query tile per workgroup, dispatched LONGEST FIRST when causal: grid (heads * N, n_tiles), blockIdx.y = rank,
    // tile = n_tiles - 1 - rank.  Tile x walks x + 1 key tiles (two per 128 queries), so the hardware's in-order dispatch turns
    // into a longest-job-first schedule: CUs that drew long tiles are topped up with short ones as they finish, and every CU ends
    // near the mean (pairing long and short tiles inside one workgroup left 384 equal jobs for 512 slots at B=8, T=1024: the CUs that
    // held two set the time, 36 tile-steps against a mean of 27).
    const int n_tiles_x = (a.Tq + TQ - 1) / TQ;
    const int rank = blockIdx.y, h = (int)blockIdx.x % a.heads, b = (int)blockIdx.x / a.heads;      // grid (heads * N, n_tiles): x runs fastest
    {
        const int qt = a.causal ? n_tiles_x - 1 - rank : rank;
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
        const int Tq = a.Tq, Tk = a.Tk;
        const float *qb = a.q + (long)b * a.q_bs + (long)h * a.q_hs;
        const float *kb = a.k + (long)b * a.kv_bs + (long)h * a.kv_hs;
        const float *vb = a.v + (long)b * a.kv_bs + (long)h * a.kv_hs;
        const int q0 = qt * TQ + wave * WQ;
        const int klim = a.key_len ? max(0, min(Tk, a.key_len[b])) : Tk;
        const int coff = Tk - Tq;

        int qrow[QB];                                              // this lane's query in each block
        bf16x8 qh[QB][I::KSTEPS], ql[QB][I::KSTEPS];               // B[k = 32ks + 8lq + e][col = query lr], pre-scaled
        f32x4 o[QB][HD / 16];                                      // O^T[dim = 16m + 4lq + r][query lr]
        // Softmax reference point: the scores leave the MFMAs already relative to a per-query reference m_ref (its negation is the
        // accumulator's initial value, so no subtraction and no zeroing per tile), and m_ref is only moved when a tile's maximum exceeds
        // it by more than REBASE (2^8): the output and the normaliser are rescaled on those tiles alone -- the first one and, on real
        // score distributions, hardly any other -- instead of on every tile.  Any reference gives the same softmax; p <= 2^8 is exact
        // headroom for the fp32 sums and for the bf16 probabilities.
        constexpr float REBASE = 8.0f;
        float mref[QB], lsum[QB];                                  // lsum: this lane's share (its 16 keys per tile) of the normaliser
        f32x4 negm[QB];                                            // -m_ref in all four slots (0 while no key has been seen)
    #pragma unroll
        for (int g = 0; g < QB; ++g) {
            qrow[g] = q0 + 16 * g + lr;
            const float *qp = qb + (long)min(qrow[g], Tq - 1) * a.q_rs;
    #pragma unroll
            for (int ks = 0; ks < I::KSTEPS; ++ks) load_split8(qp + 32 * ks + 8 * lq, a.scale * LOG2E, qh[g][ks], ql[g][ks]);   // scores in log2 units: exp is one v_exp_f32
    #pragma unroll
            for (int m = 0; m < HD / 16; ++m) o[g][m] = f32x4{0.f, 0.f, 0.f, 0.f};
            mref[g] = -INFINITY; lsum[g] = 0.f;
            negm[g] = f32x4{0.f, 0.f, 0.f, 0.f};
        }

        int n_ktiles = (klim + 63) / 64;
        if (a.causal) n_ktiles = min(n_ktiles, max(0, (min(qt * TQ + TQ - 1, Tq - 1) + coff) / 64 + 1));
        f32x4 kreg[I::UNITS], vreg[I::UNITS];
        int uoff[I::UNITS];
    #pragma unroll
        for (int i = 0; i < I::UNITS; ++i) {
            const int u = threadIdx.x + 256 * i;
            uoff[i] = (u / (HD / 4)) * (int)a.kv_rs + (u % (HD / 4)) * 4;
        }
        auto fetch_kv = [&](int t) {
            if ((t + 1) * 64 <= Tk) {
                fetch_tile_inner<HD>(kreg, kb + (long)t * 64 * a.kv_rs, uoff);
                fetch_tile_inner<HD>(vreg, vb + (long)t * 64 * a.kv_rs, uoff);
            } else {
                fetch_tile<HD>(kreg, kb, a.kv_rs, t * 64, Tk);
                fetch_tile<HD>(vreg, vb, a.kv_rs, t * 64, Tk);
            }
        };
        if (n_ktiles > 0) fetch_kv(0);
        for (int kt = 0; kt < n_ktiles; ++kt) {
            __syncthreads();
            stage_tile<HD, PASSES>(Kimg, kreg);
            stage_tile<HD, PASSES>(Vimg, vreg);
            __syncthreads();
            if (kt + 1 < n_ktiles) fetch_kv(kt + 1);
            f32x4 sacc[QB][4];                                     // S^T[key = 64kt + 16n + 4lq + r][query lr] - m_ref[query]
    #pragma unroll
            for (int n = 0; n < 4; ++n) {
    #pragma unroll
                for (int ks = 0; ks < I::KSTEPS; ++ks) {
                    const bf16x8 kh = row_frag<HD, PASSES>(Kimg, 16 * n + lr, 32 * ks + 8 * lq);
                    const bf16x8 kl = PASSES == 3 ? row_frag<HD, PASSES>(Kimg + I::BYTES, 16 * n + lr, 32 * ks + 8 * lq) : kh;
    #pragma unroll
                    for (int g = 0; g < QB; ++g) sacc[g][n] = mma<PASSES>(ks == 0 ? negm[g] : sacc[g][n], kh, kl, qh[g][ks], ql[g][ks]);
                }
            }
            bf16x8 ph[QB][2], pl[QB][2];                           // probabilities as the B operand of the two 32-slot P.V steps
    #pragma unroll
            for (int g = 0; g < QB; ++g) {
                // masks only where the tile can need them (wave-uniform): the key-length edge and the causal diagonal
                if (kt * 64 + 63 >= klim || (a.causal && kt * 64 + 63 > q0 + 16 * g + coff)) {
                    const int kmax = a.causal ? min(klim - 1, qrow[g] + coff) : klim - 1;    // last visible key of this query
                    const int th = kmax - (kt * 64 + 4 * lq);      // slot 16n + r of this lane is masked iff 16n + r > th
    #pragma unroll
                    for (int n = 0; n < 4; ++n)
    #pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (16 * n + r > th) sacc[g][n][r] = -INFINITY;
                }
                float mx = fmaxf(fmaxf(sacc[g][0][0], sacc[g][0][1]), fmaxf(sacc[g][0][2], sacc[g][0][3]));
    #pragma unroll
                for (int n = 1; n < 4; ++n) mx = fmaxf(mx, fmaxf(fmaxf(sacc[g][n][0], sacc[g][n][1]), fmaxf(sacc[g][n][2], sacc[g][n][3])));
                mx = rows4_max(mx);                                // this tile's maximum of the query, relative to m_ref
                // move the reference?  wave-uniform: when any query of the block has no reference yet or outgrew it by 2^REBASE
                if (__any((mref[g] == -INFINITY && mx > -INFINITY) || mx > REBASE)) {
                    const float shift = mref[g] == -INFINITY ? (mx == -INFINITY ? 0.f : mx) : fmaxf(mx, 0.f);
                    const float mnew = mref[g] == -INFINITY ? (mx == -INFINITY ? -INFINITY : mx) : mref[g] + shift;
                    const float alpha = __builtin_amdgcn_exp2f(-shift);            // queries that keep their reference: 2^0
                    if (mref[g] != -INFINITY) {                    // (nothing accumulated yet otherwise)
                        lsum[g] *= alpha;
    #pragma unroll
                        for (int m = 0; m < HD / 16; ++m)
    #pragma unroll
                            for (int r = 0; r < 4; ++r) o[g][m][r] *= alpha;
                    }
    #pragma unroll
                    for (int n = 0; n < 4; ++n)
    #pragma unroll
                        for (int r = 0; r < 4; ++r) sacc[g][n][r] -= shift;
                    mref[g] = mnew;
                    const float nm = mnew == -INFINITY ? 0.f : -mnew;
                    negm[g] = f32x4{nm, nm, nm, nm};
                }
                float ps = 0.f;
    #pragma unroll
                for (int n = 0; n < 4; ++n)
    #pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        sacc[g][n][r] = __builtin_amdgcn_exp2f(sacc[g][n][r]);
                        ps += sacc[g][n][r];
                    }
                lsum[g] += ps;                                     // the normaliser keeps the undropped sum
                if (DROP && a.use_drop) {
                    const uint64_t base = attn_drop_tile_base(b, a.heads, h, Tq, min(qrow[g], Tq - 1), (Tk + 63) / 64, kt);
    #pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const f32x4 dm = dropout_mult4(a.drop, base + 4 * (4 * lq + r));
    #pragma unroll
                        for (int n = 0; n < 4; ++n) sacc[g][n][r] *= dm[n];
                    }
                }
    #pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const float pf[8] = {sacc[g][2 * kk][0], sacc[g][2 * kk][1], sacc[g][2 * kk][2], sacc[g][2 * kk][3],
                                         sacc[g][2 * kk + 1][0], sacc[g][2 * kk + 1][1], sacc[g][2 * kk + 1][2], sacc[g][2 * kk + 1][3]};
                    split8(pf, ph[g][kk], pl[g][kk]);
                }
            }
            // O^T += V^T P^T : two 32-slot steps; V^T fragments come from the row-major image through the transpose read
    #pragma unroll
            for (int kk = 0; kk < 2; ++kk)
    #pragma unroll
                for (int m = 0; m < HD / 16; ++m) {
                    const bf16x8 vh = tr_frag2<HD, PASSES>(Vimg, 32 * kk + 4 * lq, 32 * kk + 16 + 4 * lq, 16 * m, lr);
                    const bf16x8 vl = PASSES == 3 ? tr_frag2<HD, PASSES>(Vimg + I::BYTES, 32 * kk + 4 * lq, 32 * kk + 16 + 4 * lq, 16 * m, lr) : vh;
    #pragma unroll
                    for (int g = 0; g < QB; ++g) o[g][m] = mma<PASSES>(o[g][m], vh, vl, ph[g][kk], pl[g][kk]);
                }
        }
    #pragma unroll
        for (int g = 0; g < QB; ++g) {
            const float lrow = rows4_sum(lsum[g]);
            if (qrow[g] < Tq) {
                const float inv = 1.0f / lrow;
                float *yp = a.y + (long)b * a.y_bs + (long)qrow[g] * a.y_rs + (long)h * HD + 4 * lq;
    #pragma unroll
                for (int m = 0; m < HD / 16; ++m)
                    *reinterpret_cast<f32x4 *>(yp + 16 * m) = f32x4{o[g][m][0] * inv, o[g][m][1] * inv, o[g][m][2] * inv, o[g][m][3] * inv};
                if (a.y_bf) {
                    __bf16 *yb = a.y_bf + (long)b * a.ybf_bs + (long)qrow[g] * a.ybf_rs + (long)h * HD + 4 * lq;
    #pragma unroll
                    for (int m = 0; m < HD / 16; ++m)
                        *reinterpret_cast<bf16x4 *>(yb + 16 * m) =
                            bf16x4{(__bf16)(o[g][m][0] * inv), (__bf16)(o[g][m][1] * inv), (__bf16)(o[g][m][2] * inv), (__bf16)(o[g][m][3] * inv)};
                }
                if (a.lse && lq == 0) a.lse[((long)b * a.heads + h) * Tq + qrow[g]] = mref[g] * LN2 + logf(lrow);
            }
        }
    }
}

// ---- backward, dQ sweep: one workgroup = 64 query rows, walks the key tiles ----------------------------------
// Same transposed formulation as the forward: S^T = K Q^T and dP^T = V dO^T leave a lane with 16 keys of ONE query, so
// lse / delta are lane scalars and dS^T is already the B operand of dQ^T = K^T dS^T (K^T through the transpose read).
template <int HD, int PASSES, bool DROP>
__global__ __launch_bounds__(256, 3) void attention_bwd_dq_mx_kernel(AttnBwdArgs a) {
    using I = Img<HD, PASSES>;
    __shared__ __attribute__((aligned(16))) char Kimg[(PASSES == 3 ? 2 : 1) * I::BYTES];      // hi | lo images (hi only in single-pass mode)
    __shared__ __attribute__((aligned(16))) char Vimg[(PASSES == 3 ? 2 : 1) * I::BYTES];
    // causal: tile n-1-x (long) then tile x (short): n + 1 key tiles per workgroup, whichever x (see the dK/dV sweep)
    const int n_tiles_x = (a.Tq + 63) / 64, h = blockIdx.y, b = blockIdx.z;
    for (int pass = 0; pass < 2; ++pass) {
        const int qt = a.causal ? (pass == 0 ? n_tiles_x - 1 - (int)blockIdx.x : (int)blockIdx.x) : (int)blockIdx.x;
        if (pass == 1 && (!a.causal || qt == n_tiles_x - 1 - (int)blockIdx.x)) break;
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
        const int Tq = a.Tq, Tk = a.Tk;
        const float *qb = a.q + (long)b * a.q_bs + (long)h * HD;
        const float *dyb = a.dy + (long)b * a.dy_bs + (long)h * HD;
        const float *kb = a.k + (long)b * a.kv_bs + (long)h * HD;
        const float *vb = a.v + (long)b * a.kv_bs + (long)h * HD;
        const int q0 = qt * 64 + wave * 16;
        const int qrow = q0 + lr, qsafe = min(qrow, Tq - 1);       // this lane's query
        const int klim = a.key_len ? max(0, min(Tk, a.key_len[b])) : Tk;
        const int coff = Tk - Tq;
        bf16x8 qh[I::KSTEPS], ql[I::KSTEPS], doh[I::KSTEPS], dol[I::KSTEPS];      // B[k = dims][col = query lr]
    #pragma unroll
        for (int ks = 0; ks < I::KSTEPS; ++ks) {
            load_split8(qb + (long)qsafe * a.q_rs + 32 * ks + 8 * lq, a.scale * LOG2E, qh[ks], ql[ks]);   // scores in log2 units: P = exp2(S' - lse')
            load_split8(dyb + (long)qsafe * a.dy_rs + 32 * ks + 8 * lq, 1.0f, doh[ks], dol[ks]);
        }
        const long stat = ((long)b * a.heads + h) * Tq + qsafe;
        float delta;
        if (a.y) {          // delta = sum_d dy[q][d] * y[q][d]: this lane's 8 * KSTEPS dims, then the four lanes that share the query
            const float *yb = a.y + (long)b * a.dy_bs + (long)h * HD + (long)qsafe * a.dy_rs;
            const float *db = dyb + (long)qsafe * a.dy_rs;
            float part = 0.f;
    #pragma unroll
            for (int ks = 0; ks < I::KSTEPS; ++ks) {
                const int d0 = 32 * ks + 8 * lq;
                const f32x4 y0 = *reinterpret_cast<const f32x4 *>(yb + d0), y1 = *reinterpret_cast<const f32x4 *>(yb + d0 + 4);
                const f32x4 g0 = *reinterpret_cast<const f32x4 *>(db + d0), g1 = *reinterpret_cast<const f32x4 *>(db + d0 + 4);
    #pragma unroll
                for (int e = 0; e < 4; ++e) part += g0[e] * y0[e] + g1[e] * y1[e];
            }
            part += __shfl_xor(part, 16, 64);
            part += __shfl_xor(part, 32, 64);
            delta = part;
            if (lq == 0 && qrow < Tq) a.delta_w[stat] = delta;
        } else {
            delta = a.delta[stat];
        }
        const float lse = a.lse[stat] * LOG2E;
        f32x4 dq[HD / 16];                                         // dQ^T[dim = 16m + 4lq + r][query lr]
    #pragma unroll
        for (int m = 0; m < HD / 16; ++m) dq[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        int n_ktiles = (klim + 63) / 64;
        if (a.causal) n_ktiles = min(n_ktiles, max(0, (min(qt * 64 + 63, Tq - 1) + coff) / 64 + 1));
        f32x4 kreg[I::UNITS], vreg[I::UNITS];
        if (n_ktiles > 0) { fetch_tile<HD>(kreg, kb, a.kv_rs, 0, Tk); fetch_tile<HD>(vreg, vb, a.kv_rs, 0, Tk); }
        for (int kt = 0; kt < n_ktiles; ++kt) {
            __syncthreads();
            stage_tile<HD, PASSES>(Kimg, kreg);
            stage_tile<HD, PASSES>(Vimg, vreg);
            __syncthreads();
            if (kt + 1 < n_ktiles) { fetch_tile<HD>(kreg, kb, a.kv_rs, (kt + 1) * 64, Tk); fetch_tile<HD>(vreg, vb, a.kv_rs, (kt + 1) * 64, Tk); }
            f32x4 sacc[4], pacc[4];                                // S^T, dP^T [key = 64kt + 16n + 4lq + r][query lr]
    #pragma unroll
            for (int n = 0; n < 4; ++n) {
                sacc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
                pacc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    #pragma unroll
                for (int ks = 0; ks < I::KSTEPS; ++ks) {
                    const bf16x8 kh = row_frag<HD, PASSES>(Kimg, 16 * n + lr, 32 * ks + 8 * lq);
                    const bf16x8 kl = PASSES == 3 ? row_frag<HD, PASSES>(Kimg + I::BYTES, 16 * n + lr, 32 * ks + 8 * lq) : kh;
                    const bf16x8 vh = row_frag<HD, PASSES>(Vimg, 16 * n + lr, 32 * ks + 8 * lq);
                    const bf16x8 vl = PASSES == 3 ? row_frag<HD, PASSES>(Vimg + I::BYTES, 16 * n + lr, 32 * ks + 8 * lq) : vh;
                    sacc[n] = mma<PASSES>(sacc[n], kh, kl, qh[ks], ql[ks]);
                    pacc[n] = mma<PASSES>(pacc[n], vh, vl, doh[ks], dol[ks]);
                }
            }
            // dS^T = P^T (dP^T . dropout - delta), in place in sacc
            const bool edge = kt * 64 + 63 >= klim || (a.causal && kt * 64 + 63 > q0 + coff);      // wave-uniform
            const int kmax = a.causal ? min(klim - 1, qrow + coff) : klim - 1;                     // last visible key of this query
    #pragma unroll
            for (int r = 0; r < 4; ++r) {
                f32x4 dm = f32x4{1.f, 1.f, 1.f, 1.f};
                if (DROP && a.use_drop)
                    dm = dropout_mult4(a.drop, attn_drop_tile_base(b, a.heads, h, Tq, qsafe, (Tk + 63) / 64, kt) + 4 * (4 * lq + r));
    #pragma unroll
                for (int n = 0; n < 4; ++n) {
                    float p = __builtin_amdgcn_exp2f(sacc[n][r] - lse);
                    if (edge && kt * 64 + 16 * n + 4 * lq + r > kmax) p = 0.f;
                    sacc[n][r] = p * (pacc[n][r] * dm[n] - delta);
                }
            }
    #pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const float df[8] = {sacc[2 * kk][0], sacc[2 * kk][1], sacc[2 * kk][2], sacc[2 * kk][3],
                                     sacc[2 * kk + 1][0], sacc[2 * kk + 1][1], sacc[2 * kk + 1][2], sacc[2 * kk + 1][3]};
                bf16x8 dh, dl;
                split8(df, dh, dl);
    #pragma unroll
                for (int m = 0; m < HD / 16; ++m) {
                    const bf16x8 kh = tr_frag2<HD, PASSES>(Kimg, 32 * kk + 4 * lq, 32 * kk + 16 + 4 * lq, 16 * m, lr);
                    const bf16x8 kl = PASSES == 3 ? tr_frag2<HD, PASSES>(Kimg + I::BYTES, 32 * kk + 4 * lq, 32 * kk + 16 + 4 * lq, 16 * m, lr) : kh;
                    dq[m] = mma<PASSES>(dq[m], kh, kl, dh, dl);
                }
            }
        }
        if (qrow < Tq && a.dq_bf) {
            __bf16 *dp = a.dq_bf + (long)b * a.dqb_bs + (long)qrow * a.dqb_rs + (long)h * HD + 4 * lq;
    #pragma unroll
            for (int m = 0; m < HD / 16; ++m)
                *reinterpret_cast<bf16x4 *>(dp + 16 * m) =
                    bf16x4{(__bf16)(dq[m][0] * a.scale), (__bf16)(dq[m][1] * a.scale), (__bf16)(dq[m][2] * a.scale), (__bf16)(dq[m][3] * a.scale)};
        } else if (qrow < Tq) {
            float *dp = a.dq + (long)b * a.dq_bs + (long)qrow * a.dq_rs + (long)h * HD + 4 * lq;
    #pragma unroll
            for (int m = 0; m < HD / 16; ++m)
                *reinterpret_cast<f32x4 *>(dp + 16 * m) = f32x4{dq[m][0] * a.scale, dq[m][1] * a.scale, dq[m][2] * a.scale, dq[m][3] * a.scale};
        }
    }
}

// The dQ sweep with QB 16-query blocks per wave (a workgroup covers 64 QB queries) and one query tile per workgroup dispatched longest
// first when causal -- what the forward gained from (every K / V / K^T fragment read from LDS feeds QB MFMAs; a staged key tile and its
// two barriers serve QB times the queries).  grid (heads * N, n_tiles).
template <int HD, int PASSES, bool DROP, int QB>
__global__ __launch_bounds__(256, QB == 1 ? 3 : 2) void attention_bwd_dq_mx_qb_kernel(AttnBwdArgs a) {
    using I = Img<HD, PASSES>;
    constexpr int WQ = 16 * QB, TQ = 64 * QB;
    __shared__ __attribute__((aligned(16))) char Kimg[(PASSES == 3 ? 2 : 1) * I::BYTES];
    __shared__ __attribute__((aligned(16))) char Vimg[(PASSES == 3 ? 2 : 1) * I::BYTES];
    const int n_tiles_x = (a.Tq + TQ - 1) / TQ;
    const int rank = blockIdx.y, h = (int)blockIdx.x % a.heads, b = (int)blockIdx.x / a.heads;
    const int qt = a.causal ? n_tiles_x - 1 - rank : rank;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
    const int Tq = a.Tq, Tk = a.Tk;
    const float *qb = a.q + (long)b * a.q_bs + (long)h * HD;
    const float *dyb = a.dy + (long)b * a.dy_bs + (long)h * HD;
    const float *kb = a.k + (long)b * a.kv_bs + (long)h * HD;
    const float *vb = a.v + (long)b * a.kv_bs + (long)h * HD;
    const int q0 = qt * TQ + wave * WQ;
    const int klim = a.key_len ? max(0, min(Tk, a.key_len[b])) : Tk;
    const int coff = Tk - Tq;
    int qrow[QB], qsafe[QB];
    bf16x8 qh[QB][I::KSTEPS], ql[QB][I::KSTEPS], doh[QB][I::KSTEPS], dol[QB][I::KSTEPS];
    float delta[QB], lse[QB];
    f32x4 dq[QB][HD / 16];
#pragma unroll
    for (int g = 0; g < QB; ++g) {
        qrow[g] = q0 + 16 * g + lr;
        qsafe[g] = min(qrow[g], Tq - 1);
#pragma unroll
        for (int ks = 0; ks < I::KSTEPS; ++ks) {
            load_split8(qb + (long)qsafe[g] * a.q_rs + 32 * ks + 8 * lq, a.scale * LOG2E, qh[g][ks], ql[g][ks]);
            load_split8(dyb + (long)qsafe[g] * a.dy_rs + 32 * ks + 8 * lq, 1.0f, doh[g][ks], dol[g][ks]);
        }
        const long stat = ((long)b * a.heads + h) * Tq + qsafe[g];
        if (a.y) {
            const float *yb = a.y + (long)b * a.dy_bs + (long)h * HD + (long)qsafe[g] * a.dy_rs;
            const float *db = dyb + (long)qsafe[g] * a.dy_rs;
            float part = 0.f;
#pragma unroll
            for (int ks = 0; ks < I::KSTEPS; ++ks) {
                const int d0 = 32 * ks + 8 * lq;
                const f32x4 y0 = *reinterpret_cast<const f32x4 *>(yb + d0), y1 = *reinterpret_cast<const f32x4 *>(yb + d0 + 4);
                const f32x4 g0 = *reinterpret_cast<const f32x4 *>(db + d0), g1 = *reinterpret_cast<const f32x4 *>(db + d0 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) part += g0[e] * y0[e] + g1[e] * y1[e];
            }
            part += __shfl_xor(part, 16, 64);
            part += __shfl_xor(part, 32, 64);
            delta[g] = part;
            if (lq == 0 && qrow[g] < Tq) a.delta_w[stat] = part;
        } else {
            delta[g] = a.delta[stat];
        }
        lse[g] = a.lse[stat] * LOG2E;
#pragma unroll
        for (int m = 0; m < HD / 16; ++m) dq[g][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    int n_ktiles = (klim + 63) / 64;
    if (a.causal) n_ktiles = min(n_ktiles, max(0, (min(qt * TQ + TQ - 1, Tq - 1) + coff) / 64 + 1));
    f32x4 kreg[I::UNITS], vreg[I::UNITS];
    if (n_ktiles > 0) { fetch_tile<HD>(kreg, kb, a.kv_rs, 0, Tk); fetch_tile<HD>(vreg, vb, a.kv_rs, 0, Tk); }
    for (int kt = 0; kt < n_ktiles; ++kt) {
        __syncthreads();
        stage_tile<HD, PASSES>(Kimg, kreg);
        stage_tile<HD, PASSES>(Vimg, vreg);
        __syncthreads();
        if (kt + 1 < n_ktiles) { fetch_tile<HD>(kreg, kb, a.kv_rs, (kt + 1) * 64, Tk); fetch_tile<HD>(vreg, vb, a.kv_rs, (kt + 1) * 64, Tk); }
        f32x4 sacc[QB][4], pacc[QB][4];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
#pragma unroll
            for (int g = 0; g < QB; ++g) { sacc[g][n] = f32x4{0.f, 0.f, 0.f, 0.f}; pacc[g][n] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int ks = 0; ks < I::KSTEPS; ++ks) {
                const bf16x8 kh = row_frag<HD, PASSES>(Kimg, 16 * n + lr, 32 * ks + 8 * lq);
                const bf16x8 kl = PASSES == 3 ? row_frag<HD, PASSES>(Kimg + I::BYTES, 16 * n + lr, 32 * ks + 8 * lq) : kh;
                const bf16x8 vh = row_frag<HD, PASSES>(Vimg, 16 * n + lr, 32 * ks + 8 * lq);
                const bf16x8 vl = PASSES == 3 ? row_frag<HD, PASSES>(Vimg + I::BYTES, 16 * n + lr, 32 * ks + 8 * lq) : vh;
#pragma unroll
                for (int g = 0; g < QB; ++g) {
                    sacc[g][n] = mma<PASSES>(sacc[g][n], kh, kl, qh[g][ks], ql[g][ks]);
                    pacc[g][n] = mma<PASSES>(pacc[g][n], vh, vl, doh[g][ks], dol[g][ks]);
                }
            }
        }
        // dS^T = P^T (dP^T . dropout - delta), in place in sacc
        const bool edge = kt * 64 + 63 >= klim || (a.causal && kt * 64 + 63 > q0 + coff);      // wave-uniform (q0: the wave's first query)
#pragma unroll
        for (int g = 0; g < QB; ++g) {
            const int kmax = a.causal ? min(klim - 1, qrow[g] + coff) : klim - 1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                f32x4 dm = f32x4{1.f, 1.f, 1.f, 1.f};
                if (DROP && a.use_drop)
                    dm = dropout_mult4(a.drop, attn_drop_tile_base(b, a.heads, h, Tq, qsafe[g], (Tk + 63) / 64, kt) + 4 * (4 * lq + r));
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    float p = __builtin_amdgcn_exp2f(sacc[g][n][r] - lse[g]);
                    if (edge && kt * 64 + 16 * n + 4 * lq + r > kmax) p = 0.f;
                    sacc[g][n][r] = p * (pacc[g][n][r] * dm[n] - delta[g]);
                }
            }
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 dh[QB], dl[QB];
#pragma unroll
            for (int g = 0; g < QB; ++g) {
                const float df[8] = {sacc[g][2 * kk][0], sacc[g][2 * kk][1], sacc[g][2 * kk][2], sacc[g][2 * kk][3],
                                     sacc[g][2 * kk + 1][0], sacc[g][2 * kk + 1][1], sacc[g][2 * kk + 1][2], sacc[g][2 * kk + 1][3]};
                split8(df, dh[g], dl[g]);
            }
#pragma unroll
            for (int m = 0; m < HD / 16; ++m) {
                const bf16x8 kh = tr_frag2<HD, PASSES>(Kimg, 32 * kk + 4 * lq, 32 * kk + 16 + 4 * lq, 16 * m, lr);
                const bf16x8 kl = PASSES == 3 ? tr_frag2<HD, PASSES>(Kimg + I::BYTES, 32 * kk + 4 * lq, 32 * kk + 16 + 4 * lq, 16 * m, lr) : kh;
#pragma unroll
                for (int g = 0; g < QB; ++g) dq[g][m] = mma<PASSES>(dq[g][m], kh, kl, dh[g], dl[g]);
            }
        }
    }
#pragma unroll
    for (int g = 0; g < QB; ++g) {
        if (qrow[g] >= Tq) continue;
        if (a.dq_bf) {
            __bf16 *dp = a.dq_bf + (long)b * a.dqb_bs + (long)qrow[g] * a.dqb_rs + (long)h * HD + 4 * lq;
#pragma unroll
            for (int m = 0; m < HD / 16; ++m)
                *reinterpret_cast<bf16x4 *>(dp + 16 * m) = bf16x4{(__bf16)(dq[g][m][0] * a.scale), (__bf16)(dq[g][m][1] * a.scale),
                                                                 (__bf16)(dq[g][m][2] * a.scale), (__bf16)(dq[g][m][3] * a.scale)};
        } else {
            float *dp = a.dq + (long)b * a.dq_bs + (long)qrow[g] * a.dq_rs + (long)h * HD + 4 * lq;
#pragma unroll
            for (int m = 0; m < HD / 16; ++m)
                *reinterpret_cast<f32x4 *>(dp + 16 * m) = f32x4{dq[g][m][0] * a.scale, dq[g][m][1] * a.scale, dq[g][m][2] * a.scale, dq[g][m][3] * a.scale};
        }
    }
}

// ---- backward, dK/dV sweep: one workgroup = 64 keys, walks the query tiles -----------------------------------
// Here the lane-fixed index is the KEY: S = Q K^T and dP = dO V^T (this wave's 16 keys as the B operand, from registers)
// leave a lane with 16 queries of one key, and P / dS are the B operands of dV^T = dO^T P and dK^T = Q^T dS.
template <int HD, int PASSES, bool DROP>
__global__ __launch_bounds__(256, 2) void attention_bwd_dkv_mx_kernel(AttnBwdArgs a) {
    using I = Img<HD, PASSES>;
    __shared__ __attribute__((aligned(16))) char Qimg[(PASSES == 3 ? 2 : 1) * I::BYTES];      // hi | lo images (hi only in single-pass mode)
    __shared__ __attribute__((aligned(16))) char Oimg[(PASSES == 3 ? 2 : 1) * I::BYTES];      // dO tile
    __shared__ __attribute__((aligned(16))) float lse_s[64], del_s[64];
    // Causal work per key tile falls from n (tile 0 sees every query tile) to 1: a workgroup takes tile x and then tile n-1-x, so every
    // workgroup walks n + 1 query tiles and the grid (ceil(n/2) wide) drains evenly instead of leaving the chip to a few long tails.
    // (dkv_longest_first, the single-pass launches: one key tile per workgroup, grid (heads * N, tiles) with the tile in blockIdx.y -- under
    // the causal mask tile 0 is the longest job and is dispatched first, and the 512 resident workgroups are topped up with ever
    // shorter ones, as in the forward: at B = 8, T = 1024, 12 heads the pairs were 768 equal jobs of 17 tile-steps on 512 slots, 34 on
    // the CUs that drew two against a mean of 25.5)
    const int n_tiles_x = (a.Tk + 63) / 64;
    const int h = a.dkv_longest_first ? (int)blockIdx.x % a.heads : (int)blockIdx.y, b = a.dkv_longest_first ? (int)blockIdx.x / a.heads : (int)blockIdx.z;
    for (int pass = 0; pass < 2; ++pass) {
        const int px = a.dkv_longest_first ? (int)blockIdx.y : (int)blockIdx.x;
        const int kt = (a.causal && !a.dkv_longest_first) ? (pass == 0 ? px : n_tiles_x - 1 - px) : px;
        if (pass == 1 && (!a.causal || a.dkv_longest_first || kt == px)) break;
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
        const int Tq = a.Tq, Tk = a.Tk;
        const float *qb = a.q + (long)b * a.q_bs + (long)h * HD;
        const float *dyb = a.dy + (long)b * a.dy_bs + (long)h * HD;
        const float *kb = a.k + (long)b * a.kv_bs + (long)h * HD;
        const float *vb = a.v + (long)b * a.kv_bs + (long)h * HD;
        const int k0 = kt * 64 + wave * 16;
        const int key = k0 + lr;                                    // this lane's key
        const int klim = a.key_len ? max(0, min(Tk, a.key_len[b])) : Tk;
        const int coff = Tk - Tq;
        bf16x8 kh[I::KSTEPS], kl[I::KSTEPS], vh[I::KSTEPS], vl[I::KSTEPS];      // B[k = dims][col = key lr]
        {
            const int krow = min(key, Tk - 1);
    #pragma unroll
            for (int ks = 0; ks < I::KSTEPS; ++ks) {
                load_split8(kb + (long)krow * a.kv_rs + 32 * ks + 8 * lq, a.scale * LOG2E, kh[ks], kl[ks]);   // scores in log2 units
                load_split8(vb + (long)krow * a.kv_rs + 32 * ks + 8 * lq, 1.0f, vh[ks], vl[ks]);
            }
        }
        f32x4 dk[HD / 16], dv[HD / 16];                            // dK^T, dV^T [dim = 16m + 4lq + r][key lr]
    #pragma unroll
        for (int m = 0; m < HD / 16; ++m) { dk[m] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[m] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        const int n_qtiles = (Tq + 63) / 64;
        const int qt0 = (a.causal && kt * 64 < klim) ? min(n_qtiles, max(0, kt * 64 - coff) / 64) : (kt * 64 < klim ? 0 : n_qtiles);
        const long stat0 = ((long)b * a.heads + h) * Tq;
        f32x4 qreg[I::UNITS], oreg[I::UNITS];
        if (qt0 < n_qtiles) { fetch_tile<HD>(qreg, qb, a.q_rs, qt0 * 64, Tq); fetch_tile<HD>(oreg, dyb, a.dy_rs, qt0 * 64, Tq); }
        for (int qt = qt0; qt < n_qtiles; ++qt) {
            __syncthreads();
            stage_tile<HD, PASSES>(Qimg, qreg);
            stage_tile<HD, PASSES>(Oimg, oreg);
            if (threadIdx.x < 64) {
                const int qrow = min(qt * 64 + (int)threadIdx.x, Tq - 1);
                lse_s[threadIdx.x] = a.lse[stat0 + qrow] * LOG2E;
                del_s[threadIdx.x] = a.delta[stat0 + qrow];
            }
            __syncthreads();
            if (qt + 1 < n_qtiles) { fetch_tile<HD>(qreg, qb, a.q_rs, (qt + 1) * 64, Tq); fetch_tile<HD>(oreg, dyb, a.dy_rs, (qt + 1) * 64, Tq); }
            f32x4 sacc[4], pacc[4];                                // S, dP [query = 64qt + 16n + 4lq + r][key lr]
    #pragma unroll
            for (int n = 0; n < 4; ++n) {
                sacc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
                pacc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    #pragma unroll
                for (int ks = 0; ks < I::KSTEPS; ++ks) {
                    const bf16x8 qfh = row_frag<HD, PASSES>(Qimg, 16 * n + lr, 32 * ks + 8 * lq);
                    const bf16x8 qfl = PASSES == 3 ? row_frag<HD, PASSES>(Qimg + I::BYTES, 16 * n + lr, 32 * ks + 8 * lq) : qfh;
                    const bf16x8 ofh = row_frag<HD, PASSES>(Oimg, 16 * n + lr, 32 * ks + 8 * lq);
                    const bf16x8 ofl = PASSES == 3 ? row_frag<HD, PASSES>(Oimg + I::BYTES, 16 * n + lr, 32 * ks + 8 * lq) : ofh;
                    sacc[n] = mma<PASSES>(sacc[n], qfh, qfl, kh[ks], kl[ks]);
                    pacc[n] = mma<PASSES>(pacc[n], ofh, ofl, vh[ks], vl[ks]);
                }
            }
            // P (dropped) stays in pacc's place, dS in sacc's: both in place
            const bool edge = k0 + 15 >= klim || qt * 64 + 63 >= Tq || (a.causal && k0 + 15 > qt * 64 + coff);   // wave-uniform
    #pragma unroll
            for (int n = 0; n < 4; ++n) {
                const f32x4 l4 = *reinterpret_cast<const f32x4 *>(&lse_s[16 * n + 4 * lq]);
                const f32x4 d4 = *reinterpret_cast<const f32x4 *>(&del_s[16 * n + 4 * lq]);
                if (!edge && !(DROP && a.use_drop)) {       // the tile lies inside the mask, no dropout: no compares
    #pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float p = __builtin_amdgcn_exp2f(sacc[n][r] - l4[r]);
                        sacc[n][r] = p * (pacc[n][r] - d4[r]);
                        pacc[n][r] = p;
                    }
                    continue;
                }
    #pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int qrow = qt * 64 + 16 * n + 4 * lq + r;
                    float p = __builtin_amdgcn_exp2f(sacc[n][r] - l4[r]);
                    if (edge && (key >= klim || qrow >= Tq || (a.causal && key > qrow + coff))) p = 0.f;
                    float dm = 1.0f;
                    if (DROP && a.use_drop)
                        dm = dropout_mult(a.drop, attn_drop_tile_base(b, a.heads, h, Tq, min(qrow, Tq - 1), (Tk + 63) / 64, kt) + 4 * lr + wave);
                    sacc[n][r] = p * (dm * pacc[n][r] - d4[r]);
                    pacc[n][r] = p * dm;
                }
            }
    #pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const float pf[8] = {pacc[2 * kk][0], pacc[2 * kk][1], pacc[2 * kk][2], pacc[2 * kk][3],
                                     pacc[2 * kk + 1][0], pacc[2 * kk + 1][1], pacc[2 * kk + 1][2], pacc[2 * kk + 1][3]};
                const float df[8] = {sacc[2 * kk][0], sacc[2 * kk][1], sacc[2 * kk][2], sacc[2 * kk][3],
                                     sacc[2 * kk + 1][0], sacc[2 * kk + 1][1], sacc[2 * kk + 1][2], sacc[2 * kk + 1][3]};
                bf16x8 ph, pl, dh, dl;
                split8(pf, ph, pl);
                split8(df, dh, dl);
    #pragma unroll
                for (int m = 0; m < HD / 16; ++m) {
                    const bf16x8 oth = tr_frag2<HD, PASSES>(Oimg, 32 * kk + 4 * lq, 32 * kk + 16 + 4 * lq, 16 * m, lr);
                    const bf16x8 otl = PASSES == 3 ? tr_frag2<HD, PASSES>(Oimg + I::BYTES, 32 * kk + 4 * lq, 32 * kk + 16 + 4 * lq, 16 * m, lr) : oth;
                    const bf16x8 qth = tr_frag2<HD, PASSES>(Qimg, 32 * kk + 4 * lq, 32 * kk + 16 + 4 * lq, 16 * m, lr);
                    const bf16x8 qtl = PASSES == 3 ? tr_frag2<HD, PASSES>(Qimg + I::BYTES, 32 * kk + 4 * lq, 32 * kk + 16 + 4 * lq, 16 * m, lr) : qth;
                    dv[m] = mma<PASSES>(dv[m], oth, otl, ph, pl);
                    dk[m] = mma<PASSES>(dk[m], qth, qtl, dh, dl);
                }
            }
        }
        if (key < Tk && a.dk_bf) {
            __bf16 *kp = a.dk_bf + (long)b * a.dqb_bs + (long)key * a.dqb_rs + (long)h * HD + 4 * lq;
            __bf16 *vp = a.dv_bf + (long)b * a.dqb_bs + (long)key * a.dqb_rs + (long)h * HD + 4 * lq;
    #pragma unroll
            for (int m = 0; m < HD / 16; ++m) {
                *reinterpret_cast<bf16x4 *>(kp + 16 * m) =
                    bf16x4{(__bf16)(dk[m][0] * a.scale), (__bf16)(dk[m][1] * a.scale), (__bf16)(dk[m][2] * a.scale), (__bf16)(dk[m][3] * a.scale)};
                *reinterpret_cast<bf16x4 *>(vp + 16 * m) = bf16x4{(__bf16)dv[m][0], (__bf16)dv[m][1], (__bf16)dv[m][2], (__bf16)dv[m][3]};
            }
        } else if (key < Tk) {
            float *kp = a.dk + (long)b * a.dkv_bs + (long)key * a.dkv_rs + (long)h * HD + 4 * lq;
            float *vp = a.dv + (long)b * a.dkv_bs + (long)key * a.dkv_rs + (long)h * HD + 4 * lq;
    #pragma unroll
            for (int m = 0; m < HD / 16; ++m) {
                *reinterpret_cast<f32x4 *>(kp + 16 * m) = f32x4{dk[m][0] * a.scale, dk[m][1] * a.scale, dk[m][2] * a.scale, dk[m][3] * a.scale};
                *reinterpret_cast<f32x4 *>(vp + 16 * m) = dv[m];
            }
        }
    }
}

template <int HD, int PASSES>
int launch_fwd(const AttnArgs &a, int N, hipStream_t st) {
    // 128-query workgroups (two 16-query blocks per wave) for the single-pass kernel once they still fill the chip: measured at
    // B=8, T=1024, 12 heads (causal, tiles paired): 61.5 vs 77.6 us; the three-pass kernel is faster on 64 (89.7 vs 116.2 us: its
    // 214 registers leave two waves per SIMD).  HALO_ATTN_QB=1|2 forces either.
    static int qb2 = -1;
    if (qb2 < 0) { const char *e = getenv("HALO_ATTN_QB"); qb2 = e ? atoi(e) : 0; }
    const long wg128 = (long)((a.Tq + 127) / 128) * a.heads * N;
    const bool two = qb2 == 2 || (qb2 == 0 && PASSES == 1 && a.Tq >= 256 && wg128 >= 512);
    const int n2 = (a.Tq + 127) / 128, n1 = (a.Tq + 63) / 64;
    const dim3 g2(a.heads * N, n2), g1(a.heads * N, n1);
    if (two && a.use_drop) hipLaunchKernelGGL((attention_fwd_mx_kernel<HD, PASSES, 2, true>), g2, dim3(256), 0, st, a);
    else if (two) hipLaunchKernelGGL((attention_fwd_mx_kernel<HD, PASSES, 2, false>), g2, dim3(256), 0, st, a);
    else if (a.use_drop) hipLaunchKernelGGL((attention_fwd_mx_kernel<HD, PASSES, 1, true>), g1, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((attention_fwd_mx_kernel<HD, PASSES, 1, false>), g1, dim3(256), 0, st, a);
    return halo_launch_status();
}

template <int HD, int PASSES>
int launch_bwd(const AttnBwdArgs &a, int N, hipStream_t st) {
    const int nq = (a.Tq + 63) / 64, nk = (a.Tk + 63) / 64;
    const dim3 gq(a.causal ? (nq + 1) / 2 : nq, a.heads, N), gk(a.causal ? (nk + 1) / 2 : nk, a.heads, N);
    // dQ sweep: 128-query workgroups (two query blocks per wave, longest tile first) in the single-pass mode once they fill the chip, as in
    // the forward; HALO_ATTN_BWD_QB=1|2 forces either
    const char *eq = getenv("HALO_ATTN_BWD_QB");               // (read per call: the tests switch it)
    const int qb = eq ? atoi(eq) : 0;
    const long wg128 = (long)((a.Tq + 127) / 128) * a.heads * N;
    const bool two = PASSES == 1 && (qb == 2 || (qb == 0 && a.Tq >= 256 && wg128 >= 512));      // (three passes: 248-256 registers, spills)
    const dim3 gq2(a.heads * N, (a.Tq + 127) / 128);
    // dK/dV sweep: one key tile per workgroup, longest first, where the paired jobs would not fill the resident slots evenly
    const char *el = getenv("HALO_ATTN_DKV_LF");
    const int lf = el ? atoi(el) : 2;
    AttnBwdArgs ak = a;
    ak.dkv_longest_first = (lf == 1 || (lf == 2 && a.causal && (long)nk * a.heads * N >= 1024)) ? 1 : 0;
    const dim3 gkl(a.heads * N, nk);
    const dim3 gkk = ak.dkv_longest_first ? gkl : gk;
    // (the dQ sweep with ONE block per wave, same rule: longest first instead of pairs)
    const bool one_lf = !two && ak.dkv_longest_first;
    const dim3 gq1(a.heads * N, nq);
    if (a.use_drop) {
        if (two) hipLaunchKernelGGL((attention_bwd_dq_mx_qb_kernel<HD, PASSES, true, 2>), gq2, dim3(256), 0, st, a);
        else if (one_lf) hipLaunchKernelGGL((attention_bwd_dq_mx_qb_kernel<HD, PASSES, true, 1>), gq1, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((attention_bwd_dq_mx_kernel<HD, PASSES, true>), gq, dim3(256), 0, st, a);
        hipLaunchKernelGGL((attention_bwd_dkv_mx_kernel<HD, PASSES, true>), gkk, dim3(256), 0, st, ak);
    } else {
        if (two) hipLaunchKernelGGL((attention_bwd_dq_mx_qb_kernel<HD, PASSES, false, 2>), gq2, dim3(256), 0, st, a);
        else if (one_lf) hipLaunchKernelGGL((attention_bwd_dq_mx_qb_kernel<HD, PASSES, false, 1>), gq1, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((attention_bwd_dq_mx_kernel<HD, PASSES, false>), gq, dim3(256), 0, st, a);
        hipLaunchKernelGGL((attention_bwd_dkv_mx_kernel<HD, PASSES, false>), gkk, dim3(256), 0, st, ak);
    }
    return halo_launch_status();
}

}  // namespace

// the fragment loads are 16-byte vector loads: row strides and head offsets must keep them aligned
static bool aligned16(const void *p) { return ((uintptr_t)p % 16) == 0; }

int halo_attention_fwd_mx(const AttnArgs &a, int N, int head_dim, int passes, hipStream_t st) {
    if (!aligned16(a.q) || a.q_rs % 4 || a.q_bs % 4 || a.q_hs % 4) return HALO_ENOTSUP;
    if (!aligned16(a.y) || a.y_rs % 4 || a.y_bs % 4) return HALO_ENOTSUP;          // the output goes out as 16-byte stores
    if (head_dim == 64) return passes == 1 ? launch_fwd<64, 1>(a, N, st) : launch_fwd<64, 3>(a, N, st);
    if (head_dim == 32) return passes == 1 ? launch_fwd<32, 1>(a, N, st) : launch_fwd<32, 3>(a, N, st);
    return HALO_ENOTSUP;
}

int halo_attention_bwd_mx(const AttnBwdArgs &a, int N, int head_dim, int passes, hipStream_t st) {
    // gradients go out as 16-byte stores
    if (a.dq_bf) {
        if (!a.dk_bf || !a.dv_bf || ((uintptr_t)a.dq_bf | (uintptr_t)a.dk_bf | (uintptr_t)a.dv_bf) % 8 || a.dqb_rs % 4 || a.dqb_bs % 4) return HALO_EINVAL;
    } else
    if (!aligned16(a.dq) || !aligned16(a.dk) || !aligned16(a.dv) || a.dq_rs % 4 || a.dq_bs % 4 || a.dkv_rs % 4 || a.dkv_bs % 4) return HALO_ENOTSUP;
    if (head_dim == 64) return passes == 1 ? launch_bwd<64, 1>(a, N, st) : launch_bwd<64, 3>(a, N, st);
    if (head_dim == 32) return passes == 1 ? launch_bwd<32, 1>(a, N, st) : launch_bwd<32, 3>(a, N, st);
    return HALO_ENOTSUP;
}
