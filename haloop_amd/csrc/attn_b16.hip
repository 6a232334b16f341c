// Softmax attention forward with ROW-MAJOR bf16 q / k / v (round 5: what halo_gemm_rows leaves as the c_attn product's result), single-pass
// bf16 arithmetic, head dimension 64, 128 queries per workgroup -- attn_mx.hip's forward formulation (scores transposed so that a lane
// holds 16 keys of ONE query: the running reference / sum are lane scalars, the probabilities are as they sit in registers the B operand
// of O^T = V^T P^T, V^T through the hardware transpose read) with the operand path rebuilt around bf16 inputs:
//   * a K / V tile (64 keys x 64 dims) is 8 KiB of bf16: two 16-byte loads per thread and tile, stored to the LDS image as they are -- no
//     fp32 fetch (half the bytes through L2), no split / convert pass between the load and the store;
//   * the freed registers hold a SECOND tile in flight: tile kt + 2 is requested while tile kt is multiplied (attn_mx.hip requests one
//     tile ahead, and its workgroups -- two per CU -- wait for that request on every tile: the forward spent most of its 41 us at B = 8,
//     T = 1024 waiting, its MFMAs busy for a sixth of it).
// Causal or full, any Tq / Tk (edge tiles clamp their rows), no key lengths, no dropout (the callers keep attn_mx.hip for those).
#include <stdlib.h>
#include "halo_common.h"
#include "halo_internal.h"
#include "attn_args.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
constexpr int HD = 64, ROWB = 160, IMG = 64 * ROWB;          // row stride of the LDS images as attn_mx.hip (conflict-free b128 and tr reads)

struct B16Args {
    const __bf16 *q, *k, *v;
    long q_rs, q_bs, kv_rs, kv_bs;           // row / batch strides in elements; heads packed inside a row (head h at column 64 h)
    float *y; long y_rs, y_bs;               // optional fp32 output
    __bf16 *yb; long yb_rs, yb_bs;           // optional bf16 output
    float *lse;
    int Tq, Tk, heads, causal;
    float scale;
};

__device__ __forceinline__ bf16x8 row_frag(const char *img, int row, int k0) { return *reinterpret_cast<const bf16x8 *>(img + row * ROWB + k0 * 2); }
__device__ __forceinline__ bf16x8 tr_frag2(const char *img, int rowA, int rowB, int col0, int lr) {
    const int off = (lr >> 2) * ROWB + (col0 + 4 * (lr & 3)) * 2;
    const bf16x4 x0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(img + rowA * ROWB + off));
    const bf16x4 x1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(img + rowB * ROWB + off));
    return bf16x8{x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
}

template <int QB>
__global__ __launch_bounds__(256, 2) void attention_fwd_b16_kernel(const B16Args a) {
    constexpr int WQ = 16 * QB, TQ = 64 * QB, KSTEPS = HD / 32;
    __shared__ __attribute__((aligned(16))) char Kimg[IMG];
    __shared__ __attribute__((aligned(16))) char Vimg[IMG];
    // one query tile per workgroup, longest first when causal (as attn_mx.hip): grid (heads * N, tiles)
    const int n_tiles_x = (a.Tq + TQ - 1) / TQ;
    const int rank = blockIdx.y, h = (int)blockIdx.x % a.heads, b = (int)blockIdx.x / a.heads;
    const int qt = a.causal ? n_tiles_x - 1 - rank : rank;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
    const int Tq = a.Tq, Tk = a.Tk;
    const __bf16 *qb = a.q + (long)b * a.q_bs + (long)h * HD;
    const __bf16 *kb = a.k + (long)b * a.kv_bs + (long)h * HD;
    const __bf16 *vb = a.v + (long)b * a.kv_bs + (long)h * HD;
    const int q0 = qt * TQ + wave * WQ;
    const int coff = Tk - Tq;

    int qrow[QB];
    bf16x8 qf[QB][KSTEPS];                                     // B[k = 32 ks + 8 lq + e][col = query lr], pre-scaled (scores in log2 units)
    f32x4 o[QB][HD / 16];
    constexpr float REBASE = 8.0f;
    float mref[QB], lsum[QB];
    f32x4 negm[QB];
    const float qs = a.scale * LOG2E;
#pragma unroll
    for (int g = 0; g < QB; ++g) {
        qrow[g] = q0 + 16 * g + lr;
        const __bf16 *qp = qb + (long)min(qrow[g], Tq - 1) * a.q_rs;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const bf16x8 raw = *reinterpret_cast<const bf16x8 *>(qp + 32 * ks + 8 * lq);
#pragma unroll
            for (int e = 0; e < 8; ++e) qf[g][ks][e] = (__bf16)((float)raw[e] * qs);
        }
#pragma unroll
        for (int m = 0; m < HD / 16; ++m) o[g][m] = f32x4{0.f, 0.f, 0.f, 0.f};
        mref[g] = -INFINITY; lsum[g] = 0.f;
        negm[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    int n_ktiles = (Tk + 63) / 64;
    if (a.causal) n_ktiles = min(n_ktiles, max(0, (min(qt * TQ + TQ - 1, Tq - 1) + coff) / 64 + 1));

    // a tile = 64 rows x 128 B: 512 units of 16 B, two per thread: unit u -> row u / 8, dims 8 (u % 8) ..
    const int urow[2] = {(int)threadIdx.x >> 3, ((int)threadIdx.x + 256) >> 3}, ud = ((int)threadIdx.x & 7) * 8;
    auto fetch = [&](int t, u32x4 (&kr)[2], u32x4 (&vr)[2]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const long off = (long)min(t * 64 + urow[i], Tk - 1) * a.kv_rs + ud;
            kr[i] = *reinterpret_cast<const u32x4 *>(kb + off);
            vr[i] = *reinterpret_cast<const u32x4 *>(vb + off);
        }
    };
    auto stage = [&](const u32x4 (&kr)[2], const u32x4 (&vr)[2]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<u32x4 *>(Kimg + urow[i] * ROWB + ud * 2) = kr[i];
            *reinterpret_cast<u32x4 *>(Vimg + urow[i] * ROWB + ud * 2) = vr[i];
        }
    };
    auto compute = [&](int kt) {
        f32x4 sacc[QB][4];                                     // S^T[key = 64 kt + 16 n + 4 lq + r][query lr] - m_ref[query]
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                const bf16x8 kf = row_frag(Kimg, 16 * n + lr, 32 * ks + 8 * lq);
#pragma unroll
                for (int g = 0; g < QB; ++g) sacc[g][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[g][ks], ks == 0 ? negm[g] : sacc[g][n], 0, 0, 0);
            }
        bf16x8 pf[QB][2];
#pragma unroll
        for (int g = 0; g < QB; ++g) {
            if (kt * 64 + 63 >= Tk || (a.causal && kt * 64 + 63 > q0 + 16 * g + coff)) {       // (wave-uniform) the edge of the keys, the causal diagonal
                const int kmax = a.causal ? min(Tk - 1, qrow[g] + coff) : Tk - 1;
                const int th = kmax - (kt * 64 + 4 * lq);
#pragma unroll
                for (int n = 0; n < 4; ++n)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (16 * n + r > th) sacc[g][n][r] = -INFINITY;
            }
            float mx = fmaxf(fmaxf(sacc[g][0][0], sacc[g][0][1]), fmaxf(sacc[g][0][2], sacc[g][0][3]));
#pragma unroll
            for (int n = 1; n < 4; ++n) mx = fmaxf(mx, fmaxf(fmaxf(sacc[g][n][0], sacc[g][n][1]), fmaxf(sacc[g][n][2], sacc[g][n][3])));
            mx = rows4_max(mx);
            if (__any((mref[g] == -INFINITY && mx > -INFINITY) || mx > REBASE)) {              // move the reference (attn_mx.hip: rarely after the first tile)
                const float shift = mref[g] == -INFINITY ? (mx == -INFINITY ? 0.f : mx) : fmaxf(mx, 0.f);
                const float mnew = mref[g] == -INFINITY ? (mx == -INFINITY ? -INFINITY : mx) : mref[g] + shift;
                const float alpha = __builtin_amdgcn_exp2f(-shift);
                if (mref[g] != -INFINITY) {
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
            lsum[g] += ps;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int e = 0; e < 8; ++e) pf[g][kk][e] = (__bf16)sacc[g][2 * kk + (e >> 2)][e & 3];
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int m = 0; m < HD / 16; ++m) {
                const bf16x8 vf = tr_frag2(Vimg, 32 * kk + 4 * lq, 32 * kk + 16 + 4 * lq, 16 * m, lr);
#pragma unroll
                for (int g = 0; g < QB; ++g) o[g][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[g][kk], o[g][m], 0, 0, 0);
            }
    };

    u32x4 k0[2], v0[2], k1[2], v1[2];                          // two tiles in flight
    if (n_ktiles > 0) fetch(0, k0, v0);
    if (n_ktiles > 1) fetch(1, k1, v1);
    for (int kt = 0; kt < n_ktiles; kt += 2) {
        __syncthreads();
        stage(k0, v0);
        __syncthreads();
        if (kt + 2 < n_ktiles) fetch(kt + 2, k0, v0);
        compute(kt);
        if (kt + 1 < n_ktiles) {
            __syncthreads();
            stage(k1, v1);
            __syncthreads();
            if (kt + 3 < n_ktiles) fetch(kt + 3, k1, v1);
            compute(kt + 1);
        }
    }
#pragma unroll
    for (int g = 0; g < QB; ++g) {
        const float lrow = rows4_sum(lsum[g]);
        if (qrow[g] < Tq) {
            const float inv = 1.0f / lrow;
            if (a.y) {
                float *yp = a.y + (long)b * a.y_bs + (long)qrow[g] * a.y_rs + (long)h * HD + 4 * lq;
#pragma unroll
                for (int m = 0; m < HD / 16; ++m)
                    *reinterpret_cast<f32x4 *>(yp + 16 * m) = f32x4{o[g][m][0] * inv, o[g][m][1] * inv, o[g][m][2] * inv, o[g][m][3] * inv};
            }
            if (a.yb) {
                __bf16 *yb = a.yb + (long)b * a.yb_bs + (long)qrow[g] * a.yb_rs + (long)h * HD + 4 * lq;
#pragma unroll
                for (int m = 0; m < HD / 16; ++m)
                    *reinterpret_cast<bf16x4 *>(yb + 16 * m) =
                        bf16x4{(__bf16)(o[g][m][0] * inv), (__bf16)(o[g][m][1] * inv), (__bf16)(o[g][m][2] * inv), (__bf16)(o[g][m][3] * inv)};
            }
            if (a.lse && lq == 0) a.lse[((long)b * a.heads + h) * Tq + qrow[g]] = mref[g] * LN2 + logf(lrow);
        }
    }
}


// ---- backward ------------------------------------------------------------------------------------------------------------------
// The two sweeps of attn_mx.hip (dQ: a workgroup = 128 queries walking the key tiles; dK / dV: a workgroup = 64 keys walking the query
// tiles; one tile per workgroup, longest first under the causal mask) on bf16 rows: q / k / v from the c_attn product's bf16 result, dy
// from the c_proj input-gradient product's bf16 result, y as the forward's bf16 output (delta = rowsum(dy * y) is formed in the dQ sweep
// from those and handed to the dK / dV sweep through delta_w).  Gradients leave as row-major bf16, as halo_attention_bwd_bf16's.
struct B16BwdArgs {
    const __bf16 *q, *k, *v, *y, *dy;
    long q_rs, q_bs, kv_rs, kv_bs, y_rs, y_bs;     // y and dy share their strides
    const float *lse;
    float *delta;                                  // [N * heads * Tq]: written by the dQ sweep, read by the dK / dV sweep
    __bf16 *dq, *dk, *dv;
    long d_rs, d_bs;
    int Tq, Tk, heads, causal;
    float scale;
};

__global__ __launch_bounds__(256, 2) void attention_bwd_dq_b16_kernel(const B16BwdArgs a) {
    constexpr int QB = 2, WQ = 16 * QB, TQ = 64 * QB, KSTEPS = HD / 32;
    __shared__ __attribute__((aligned(16))) char Kimg[IMG];
    __shared__ __attribute__((aligned(16))) char Vimg[IMG];
    const int n_tiles_x = (a.Tq + TQ - 1) / TQ;
    const int rank = blockIdx.y, h = (int)blockIdx.x % a.heads, b = (int)blockIdx.x / a.heads;
    const int qt = a.causal ? n_tiles_x - 1 - rank : rank;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
    const int Tq = a.Tq, Tk = a.Tk;
    const __bf16 *qb = a.q + (long)b * a.q_bs + (long)h * HD;
    const __bf16 *kb = a.k + (long)b * a.kv_bs + (long)h * HD;
    const __bf16 *vb = a.v + (long)b * a.kv_bs + (long)h * HD;
    const __bf16 *yb = a.y + (long)b * a.y_bs + (long)h * HD;
    const __bf16 *dyb = a.dy + (long)b * a.y_bs + (long)h * HD;
    const int q0 = qt * TQ + wave * WQ;
    const int coff = Tk - Tq;
    int qrow[QB];
    bf16x8 qf[QB][KSTEPS], dof[QB][KSTEPS];
    float delta[QB], lse[QB];
    f32x4 dq[QB][HD / 16];
    const float qs = a.scale * LOG2E;
#pragma unroll
    for (int g = 0; g < QB; ++g) {
        qrow[g] = q0 + 16 * g + lr;
        const int qsafe = min(qrow[g], Tq - 1);
        float part = 0.f;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const bf16x8 raw = *reinterpret_cast<const bf16x8 *>(qb + (long)qsafe * a.q_rs + 32 * ks + 8 * lq);
            dof[g][ks] = *reinterpret_cast<const bf16x8 *>(dyb + (long)qsafe * a.y_rs + 32 * ks + 8 * lq);
            const bf16x8 yv = *reinterpret_cast<const bf16x8 *>(yb + (long)qsafe * a.y_rs + 32 * ks + 8 * lq);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                qf[g][ks][e] = (__bf16)((float)raw[e] * qs);
                part += (float)dof[g][ks][e] * (float)yv[e];
            }
        }
        part += __shfl_xor(part, 16, 64);
        part += __shfl_xor(part, 32, 64);
        delta[g] = part;
        const long stat = ((long)b * a.heads + h) * Tq + qsafe;
        if (lq == 0 && qrow[g] < Tq) a.delta[stat] = part;
        lse[g] = a.lse[stat] * LOG2E;
#pragma unroll
        for (int m = 0; m < HD / 16; ++m) dq[g][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    int n_ktiles = (Tk + 63) / 64;
    if (a.causal) n_ktiles = min(n_ktiles, max(0, (min(qt * TQ + TQ - 1, Tq - 1) + coff) / 64 + 1));
    const int urow[2] = {(int)threadIdx.x >> 3, ((int)threadIdx.x + 256) >> 3}, ud = ((int)threadIdx.x & 7) * 8;
    auto fetch = [&](int t, u32x4 (&kr)[2], u32x4 (&vr)[2]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const long off = (long)min(t * 64 + urow[i], Tk - 1) * a.kv_rs + ud;
            kr[i] = *reinterpret_cast<const u32x4 *>(kb + off);
            vr[i] = *reinterpret_cast<const u32x4 *>(vb + off);
        }
    };
    auto stage = [&](const u32x4 (&kr)[2], const u32x4 (&vr)[2]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<u32x4 *>(Kimg + urow[i] * ROWB + ud * 2) = kr[i];
            *reinterpret_cast<u32x4 *>(Vimg + urow[i] * ROWB + ud * 2) = vr[i];
        }
    };
    auto compute = [&](int kt) {
        f32x4 sacc[QB][4], pacc[QB][4];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
#pragma unroll
            for (int g = 0; g < QB; ++g) { sacc[g][n] = f32x4{0.f, 0.f, 0.f, 0.f}; pacc[g][n] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                const bf16x8 kf = row_frag(Kimg, 16 * n + lr, 32 * ks + 8 * lq);
                const bf16x8 vf = row_frag(Vimg, 16 * n + lr, 32 * ks + 8 * lq);
#pragma unroll
                for (int g = 0; g < QB; ++g) {
                    sacc[g][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[g][ks], sacc[g][n], 0, 0, 0);
                    pacc[g][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, dof[g][ks], pacc[g][n], 0, 0, 0);
                }
            }
        }
        // dS^T = P^T (dP^T - delta), in place in sacc
        const bool edge = kt * 64 + 63 >= Tk || (a.causal && kt * 64 + 63 > q0 + coff);      // wave-uniform
#pragma unroll
        for (int g = 0; g < QB; ++g) {
            const int kmax = a.causal ? min(Tk - 1, qrow[g] + coff) : Tk - 1;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    float p = __builtin_amdgcn_exp2f(sacc[g][n][r] - lse[g]);
                    if (edge && kt * 64 + 16 * n + 4 * lq + r > kmax) p = 0.f;
                    sacc[g][n][r] = p * (pacc[g][n][r] - delta[g]);
                }
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 df[QB];
#pragma unroll
            for (int g = 0; g < QB; ++g)
#pragma unroll
                for (int e = 0; e < 8; ++e) df[g][e] = (__bf16)sacc[g][2 * kk + (e >> 2)][e & 3];
#pragma unroll
            for (int m = 0; m < HD / 16; ++m) {
                const bf16x8 kt_f = tr_frag2(Kimg, 32 * kk + 4 * lq, 32 * kk + 16 + 4 * lq, 16 * m, lr);
#pragma unroll
                for (int g = 0; g < QB; ++g) dq[g][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt_f, df[g], dq[g][m], 0, 0, 0);
            }
        }
    };
    u32x4 k0[2], v0[2], k1[2], v1[2];
    if (n_ktiles > 0) fetch(0, k0, v0);
    if (n_ktiles > 1) fetch(1, k1, v1);
    for (int kt = 0; kt < n_ktiles; kt += 2) {
        __syncthreads();
        stage(k0, v0);
        __syncthreads();
        if (kt + 2 < n_ktiles) fetch(kt + 2, k0, v0);
        compute(kt);
        if (kt + 1 < n_ktiles) {
            __syncthreads();
            stage(k1, v1);
            __syncthreads();
            if (kt + 3 < n_ktiles) fetch(kt + 3, k1, v1);
            compute(kt + 1);
        }
    }
#pragma unroll
    for (int g = 0; g < QB; ++g) {
        if (qrow[g] >= Tq) continue;
        __bf16 *dp = a.dq + (long)b * a.d_bs + (long)qrow[g] * a.d_rs + (long)h * HD + 4 * lq;
#pragma unroll
        for (int m = 0; m < HD / 16; ++m)
            *reinterpret_cast<bf16x4 *>(dp + 16 * m) = bf16x4{(__bf16)(dq[g][m][0] * a.scale), (__bf16)(dq[g][m][1] * a.scale),
                                                             (__bf16)(dq[g][m][2] * a.scale), (__bf16)(dq[g][m][3] * a.scale)};
    }
}

// one workgroup = 64 keys (a wave 16: the lane-fixed index is the KEY), walking the query tiles from the causal diagonal on
__global__ __launch_bounds__(256, 2) void attention_bwd_dkv_b16_kernel(const B16BwdArgs a) {
    constexpr int KSTEPS = HD / 32;
    __shared__ __attribute__((aligned(16))) char Qimg[IMG];
    __shared__ __attribute__((aligned(16))) char Oimg[IMG];
    __shared__ __attribute__((aligned(16))) float lse_s[64], del_s[64];
    const int h = (int)blockIdx.x % a.heads, b = (int)blockIdx.x / a.heads, kt = blockIdx.y;       // (causal: tile 0 is the longest job and the first dispatched)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
    const int Tq = a.Tq, Tk = a.Tk;
    const __bf16 *qb = a.q + (long)b * a.q_bs + (long)h * HD;
    const __bf16 *dyb = a.dy + (long)b * a.y_bs + (long)h * HD;
    const __bf16 *kb = a.k + (long)b * a.kv_bs + (long)h * HD;
    const __bf16 *vb = a.v + (long)b * a.kv_bs + (long)h * HD;
    const int k0 = kt * 64 + wave * 16, key = k0 + lr;
    const int coff = Tk - Tq;
    bf16x8 kf[KSTEPS], vf[KSTEPS];                             // B[k = dims][col = key lr]; K pre-scaled: scores in log2 units
    {
        const int krow = min(key, Tk - 1);
        const float ksc = a.scale * LOG2E;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const bf16x8 raw = *reinterpret_cast<const bf16x8 *>(kb + (long)krow * a.kv_rs + 32 * ks + 8 * lq);
#pragma unroll
            for (int e = 0; e < 8; ++e) kf[ks][e] = (__bf16)((float)raw[e] * ksc);
            vf[ks] = *reinterpret_cast<const bf16x8 *>(vb + (long)krow * a.kv_rs + 32 * ks + 8 * lq);
        }
    }
    f32x4 dk[HD / 16], dv[HD / 16];
#pragma unroll
    for (int m = 0; m < HD / 16; ++m) { dk[m] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[m] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const int n_qtiles = (Tq + 63) / 64;
    const int qt0 = a.causal ? min(n_qtiles, max(0, kt * 64 - coff) / 64) : 0;
    const long stat0 = ((long)b * a.heads + h) * Tq;
    const int urow[2] = {(int)threadIdx.x >> 3, ((int)threadIdx.x + 256) >> 3}, ud = ((int)threadIdx.x & 7) * 8;
    auto fetch = [&](int t, u32x4 (&qr)[2], u32x4 (&orr)[2], float &l, float &d) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const long row = min(t * 64 + urow[i], Tq - 1);
            qr[i] = *reinterpret_cast<const u32x4 *>(qb + row * a.q_rs + ud);
            orr[i] = *reinterpret_cast<const u32x4 *>(dyb + row * a.y_rs + ud);
        }
        const int qrow = min(t * 64 + ((int)threadIdx.x & 63), Tq - 1);       // (threads 0-63 keep theirs)
        l = a.lse[stat0 + qrow] * LOG2E;
        d = a.delta[stat0 + qrow];
    };
    auto stage = [&](const u32x4 (&qr)[2], const u32x4 (&orr)[2], float l, float d) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<u32x4 *>(Qimg + urow[i] * ROWB + ud * 2) = qr[i];
            *reinterpret_cast<u32x4 *>(Oimg + urow[i] * ROWB + ud * 2) = orr[i];
        }
        if (threadIdx.x < 64) { lse_s[threadIdx.x] = l; del_s[threadIdx.x] = d; }
    };
    auto compute = [&](int qt) {
        f32x4 sacc[4], pacc[4];                                // S, dP [query = 64 qt + 16 n + 4 lq + r][key lr]
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            sacc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
            pacc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                const bf16x8 qfr = row_frag(Qimg, 16 * n + lr, 32 * ks + 8 * lq);
                const bf16x8 ofr = row_frag(Oimg, 16 * n + lr, 32 * ks + 8 * lq);
                sacc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qfr, kf[ks], sacc[n], 0, 0, 0);
                pacc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ofr, vf[ks], pacc[n], 0, 0, 0);
            }
        }
        const bool edge = k0 + 15 >= Tk || qt * 64 + 63 >= Tq || (a.causal && k0 + 15 > qt * 64 + coff);   // wave-uniform
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const f32x4 l4 = *reinterpret_cast<const f32x4 *>(&lse_s[16 * n + 4 * lq]);
            const f32x4 d4 = *reinterpret_cast<const f32x4 *>(&del_s[16 * n + 4 * lq]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qrow = qt * 64 + 16 * n + 4 * lq + r;
                float p = __builtin_amdgcn_exp2f(sacc[n][r] - l4[r]);
                if (edge && (key >= Tk || qrow >= Tq || (a.causal && key > qrow + coff))) p = 0.f;
                sacc[n][r] = p * (pacc[n][r] - d4[r]);
                pacc[n][r] = p;
            }
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 pf, df;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                pf[e] = (__bf16)pacc[2 * kk + (e >> 2)][e & 3];
                df[e] = (__bf16)sacc[2 * kk + (e >> 2)][e & 3];
            }
#pragma unroll
            for (int m = 0; m < HD / 16; ++m) {
                const bf16x8 ot = tr_frag2(Oimg, 32 * kk + 4 * lq, 32 * kk + 16 + 4 * lq, 16 * m, lr);
                const bf16x8 qtt = tr_frag2(Qimg, 32 * kk + 4 * lq, 32 * kk + 16 + 4 * lq, 16 * m, lr);
                dv[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ot, pf, dv[m], 0, 0, 0);
                dk[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qtt, df, dk[m], 0, 0, 0);
            }
        }
    };
    u32x4 q0r[2], o0r[2], q1r[2], o1r[2];
    float l0 = 0.f, d0 = 0.f, l1 = 0.f, d1 = 0.f;
    if (qt0 < n_qtiles) fetch(qt0, q0r, o0r, l0, d0);
    if (qt0 + 1 < n_qtiles) fetch(qt0 + 1, q1r, o1r, l1, d1);
    for (int qt = qt0; qt < n_qtiles; qt += 2) {
        __syncthreads();
        stage(q0r, o0r, l0, d0);
        __syncthreads();
        if (qt + 2 < n_qtiles) fetch(qt + 2, q0r, o0r, l0, d0);
        compute(qt);
        if (qt + 1 < n_qtiles) {
            __syncthreads();
            stage(q1r, o1r, l1, d1);
            __syncthreads();
            if (qt + 3 < n_qtiles) fetch(qt + 3, q1r, o1r, l1, d1);
            compute(qt + 1);
        }
    }
    if (key < Tk) {
        __bf16 *kp = a.dk + (long)b * a.d_bs + (long)key * a.d_rs + (long)h * HD + 4 * lq;
        __bf16 *vp = a.dv + (long)b * a.d_bs + (long)key * a.d_rs + (long)h * HD + 4 * lq;
#pragma unroll
        for (int m = 0; m < HD / 16; ++m) {
            *reinterpret_cast<bf16x4 *>(kp + 16 * m) =
                bf16x4{(__bf16)(dk[m][0] * a.scale), (__bf16)(dk[m][1] * a.scale), (__bf16)(dk[m][2] * a.scale), (__bf16)(dk[m][3] * a.scale)};
            *reinterpret_cast<bf16x4 *>(vp + 16 * m) = bf16x4{(__bf16)dv[m][0], (__bf16)dv[m][1], (__bf16)dv[m][2], (__bf16)dv[m][3]};
        }
    }
}

}  // namespace

extern "C" {

int halo_attention_fwd_b16(const void *q, long q_row_stride, long q_batch_stride, const void *k, const void *v, long kv_row_stride,
                           long kv_batch_stride, float *y, long y_row_stride, long y_batch_stride, void *y_bf16, long yb_row_stride,
                           long yb_batch_stride, float *lse, int N, int heads, int head_dim, int Tq, int Tk, int causal, halo_stream_t stream) {
    HALO_CHECK_ARG(q && k && v && (y || y_bf16) && N > 0 && heads > 0 && Tq > 0 && Tk > 0 && N <= 65535 && heads <= 65535);
    if (head_dim != 64 || halo_math_mode() != HALO_MATH_BF16) return HALO_ENOTSUP;
    HALO_CHECK_ARG(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) % 16 == 0 && q_row_stride % 8 == 0 && q_batch_stride % 8 == 0 &&
                   kv_row_stride % 8 == 0 && kv_batch_stride % 8 == 0);
    HALO_CHECK_ARG(!y || ((uintptr_t)y % 16 == 0 && y_row_stride % 4 == 0 && y_batch_stride % 4 == 0));
    HALO_CHECK_ARG(!y_bf16 || ((uintptr_t)y_bf16 % 8 == 0 && yb_row_stride % 4 == 0 && yb_batch_stride % 4 == 0));
    B16Args a;
    a.q = (const __bf16 *)q; a.k = (const __bf16 *)k; a.v = (const __bf16 *)v;
    a.q_rs = q_row_stride; a.q_bs = q_batch_stride; a.kv_rs = kv_row_stride; a.kv_bs = kv_batch_stride;
    a.y = y; a.y_rs = y_row_stride; a.y_bs = y_batch_stride;
    a.yb = (__bf16 *)y_bf16; a.yb_rs = yb_row_stride; a.yb_bs = yb_batch_stride;
    a.lse = lse; a.Tq = Tq; a.Tk = Tk; a.heads = heads; a.causal = causal;
    a.scale = 1.0f / sqrtf((float)head_dim);
    // one block of 16 queries per wave (64 per workgroup: twice the workgroups, 28.5 against 30.1 us on 8 x 1024 x 12 heads); HALO_ATTN_B16_QB=2:
    // two blocks per wave (every K / V fragment feeds two MFMAs)
    const char *qb_env = getenv("HALO_ATTN_B16_QB");
    if (!qb_env || atoi(qb_env) != 2)
        hipLaunchKernelGGL(attention_fwd_b16_kernel<1>, dim3((unsigned)(heads * N), (unsigned)((Tq + 63) / 64)), dim3(256), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(attention_fwd_b16_kernel<2>, dim3((unsigned)(heads * N), (unsigned)((Tq + 127) / 128)), dim3(256), 0, (hipStream_t)stream, a);
    return halo_launch_status();
}

int halo_attention_bwd_b16(const void *q, long q_row_stride, long q_batch_stride, const void *k, const void *v, long kv_row_stride,
                           long kv_batch_stride, const void *y_bf16, const void *dy_bf16, long y_row_stride, long y_batch_stride, const float *lse,
                           float *delta, void *dq_bf16, void *dk_bf16, void *dv_bf16, long d_row_stride, long d_batch_stride, int N, int heads,
                           int head_dim, int Tq, int Tk, int causal, halo_stream_t stream) {
    HALO_CHECK_ARG(q && k && v && y_bf16 && dy_bf16 && lse && delta && dq_bf16 && dk_bf16 && dv_bf16 && N > 0 && heads > 0 && Tq > 0 && Tk > 0 &&
                   N <= 65535 && heads <= 65535);
    if (head_dim != 64 || halo_math_mode() != HALO_MATH_BF16) return HALO_ENOTSUP;
    HALO_CHECK_ARG(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)y_bf16 | (uintptr_t)dy_bf16) % 16 == 0);
    HALO_CHECK_ARG(q_row_stride % 8 == 0 && q_batch_stride % 8 == 0 && kv_row_stride % 8 == 0 && kv_batch_stride % 8 == 0 && y_row_stride % 8 == 0 &&
                   y_batch_stride % 8 == 0);
    HALO_CHECK_ARG(((uintptr_t)dq_bf16 | (uintptr_t)dk_bf16 | (uintptr_t)dv_bf16) % 8 == 0 && d_row_stride % 4 == 0 && d_batch_stride % 4 == 0);
    B16BwdArgs a;
    a.q = (const __bf16 *)q; a.k = (const __bf16 *)k; a.v = (const __bf16 *)v; a.y = (const __bf16 *)y_bf16; a.dy = (const __bf16 *)dy_bf16;
    a.q_rs = q_row_stride; a.q_bs = q_batch_stride; a.kv_rs = kv_row_stride; a.kv_bs = kv_batch_stride; a.y_rs = y_row_stride; a.y_bs = y_batch_stride;
    a.lse = lse; a.delta = delta;
    a.dq = (__bf16 *)dq_bf16; a.dk = (__bf16 *)dk_bf16; a.dv = (__bf16 *)dv_bf16; a.d_rs = d_row_stride; a.d_bs = d_batch_stride;
    a.Tq = Tq; a.Tk = Tk; a.heads = heads; a.causal = causal;
    a.scale = 1.0f / sqrtf((float)head_dim);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(attention_bwd_dq_b16_kernel, dim3((unsigned)(heads * N), (unsigned)((Tq + 127) / 128)), dim3(256), 0, st, a);
    hipLaunchKernelGGL(attention_bwd_dkv_b16_kernel, dim3((unsigned)(heads * N), (unsigned)((Tk + 63) / 64)), dim3(256), 0, st, a);
    return halo_launch_status();
}

}  // extern "C"
