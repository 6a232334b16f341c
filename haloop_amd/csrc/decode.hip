// Fused launches of one batched greedy decode step of the encoder-decoder ASR model (ha/transformer.py:160-195, the
// T == 1 branch of Block.forward :476-494 under kv caches): per decoder layer
//   1. LayerNorm(ln_time) + ONE product for the cross query and the self q | k | v     (dec_linear_kernel, LN variant)
//   2. cross-attention over the cached memory keys AND cache store + rotary + self-attention, one launch (dec_attention_pair_kernel)
//   3. [cross | self] attention outputs x [proj_memory | proj_time] weights, residual add   (dec_linear_kernel, K = 2C, accumulate)
//   4. LayerNorm(ln_chan) + mix_chan[0] + exact GELU                                     (dec_linear_kernel, LN variant, GELU)
//   5. mix_chan[2], residual add                                                        (dec_linear_kernel, K = 4C, accumulate)
// and per token LayerNorm(ln_f) + lm_head (dec_linear_kernel) and log-softmax / argmax / entropy / bookkeeping / next embedding
// (dec_token_kernel): 5 launches per layer and 2 per token instead of 13 and 5.
//
// A decode step has M = N utterances (64) rows: a tiled GEMM has nothing to tile.  dec_linear_kernel gives one workgroup 16 rows x
// (16 * NT) output features and the whole K: activations are read as fp32 rows straight into the A-fragment layout of
// v_mfma_f32_16x16x32_bf16 (lane = (row, 8 consecutive k)), split into bf16 hi + lo in registers, and multiplied against weight
// fragments that were split and laid out once ("decode images": [feature tile][k step][hi | lo][lane][8], 1 KiB per wave load,
// global -> registers, no LDS staging: every weight element is used by exactly one MFMA of the workgroup).  The four waves split K and
// their partial tiles are summed in a fixed order through LDS (bitwise reproducible).  Arithmetic: split-bf16, three MFMAs per
// product, fp32 accumulate (the bf16x3 mode of the library, in every math mode but exact f32, where the caller keeps the old path).
#include <hip/hip_fp16.h>
#include <type_traits>
#include "halo_common.h"
#include "halo_internal.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split8(const float *x, bf16x8 &hi, bf16x8 &lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const __bf16 h = (__bf16)x[e];
        hi[e] = h;
        lo[e] = (__bf16)(x[e] - (float)h);
    }
}

// decode image of W [n_out][K] (row-major, leading dimension ld): block (ks, nt) writes the hi and the lo fragment of feature tile nt,
// k step ks: lane l holds W[nt*16 + l%16][ks*32 + 8*(l/16) .. +8] (the B operand of the 16x16x32 MFMA); zero past the edges
__global__ __launch_bounds__(128) void dec_pack_kernel(const float *__restrict__ W, int n_out, int K, long ld, char *__restrict__ img) {
    const int ks = blockIdx.x, nt = blockIdx.y, KS = gridDim.x;
    const int part = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = nt * 16 + (lane & 15), k0 = ks * 32 + 8 * (lane >> 4);
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = (n < n_out && k0 + j < K) ? W[(long)n * ld + k0 + j] : 0.f;
    bf16x8 hi, lo;
    split8(x, hi, lo);
    *reinterpret_cast<bf16x8 *>(img + (((long)nt * KS + ks) * 2 + part) * 1024 + lane * 16) = part ? lo : hi;
}

struct DecLinearArgs {
    const float *x;        // [rows][K] fp32, leading dimension ldx
    long ldx;
    int rows, K;
    const float *lnw;      // LayerNorm weight [K] (LN variants)
    float eps;
    const char *w;         // decode image of W [n_out][K]
    int n_tiles, n_out;
    float *out;            // [rows][n_out], leading dimension ldo
    long ldo;
    int flags;             // HALO_GEMM_ACCUM: out += ...; HALO_GEMM_GELU_ERF: exact GELU
    // The residual stream as a PAIR (main, side): x = main + side.  LN variants: x2 != NULL is the side (same strides as x), added to the rows
    // before the statistics.  Accumulating products with side_out != NULL run as TWO K-slices (grid.z): slice 0 writes
    // out = (out + side_in) + its half of the product, slice 1 its half alone to side_out -- twice the workgroups, each streaming half of K,
    // every sum in a fixed order; the next launch reads out + side_out.
    const float *x2, *side_in;
    float *side_out;
};

// grid (feature groups of 16*NT, row groups of 16); KSW_LN > 0: F.layer_norm (no bias) of the rows first, K == 128 * KSW_LN
template <int NT, int KSW_LN>
__global__ __launch_bounds__(256) void dec_linear_kernel(const DecLinearArgs p) {
    __shared__ float red[4][NT][64][4];
    __shared__ float stat[4][16];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int row = blockIdx.y * 16 + r;
    const bool rok = row < p.rows;
    const int nt0 = blockIdx.x * NT;
    const int nsl = (KSW_LN == 0 && p.side_out) ? 2 : 1, kslice = nsl == 2 ? (int)blockIdx.z : 0;
    const int KS = p.K / 32, ksw = KS / nsl / 4, ks0 = kslice * (KS / nsl) + wave * ksw;
    const float *xr = p.x + (long)(rok ? row : 0) * p.ldx + g * 8;
    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // a feature tile past the last one (odd tile count, NT = 2) re-reads the last tile; its sums are never stored
    const char *wbase[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) wbase[nt] = p.w + (long)min(nt0 + nt, p.n_tiles - 1) * KS * 2048 + lane * 16;
    auto product = [&](const float *xv, const bf16x8 (&wh)[NT], const bf16x8 (&wl)[NT]) {
        bf16x8 ah, al;
        split8(xv, ah, al);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, wh[nt], acc[nt], 0, 0, 0);
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wl[nt], acc[nt], 0, 0, 0);
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wh[nt], acc[nt], 0, 0, 0);
        }
    };
    if constexpr (KSW_LN > 0) {
        constexpr int NK = KSW_LN;
        float xv[NK][8];
        bf16x8 wh[NK][NT], wl[NK][NT];
#pragma unroll
        for (int i = 0; i < NK; ++i) {
            f32x4 a = *reinterpret_cast<const f32x4 *>(xr + (ks0 + i) * 32), b = *reinterpret_cast<const f32x4 *>(xr + (ks0 + i) * 32 + 4);
            if (p.x2) {                                      // x = main + side (uniform branch)
                const float *x2r = p.x2 + (xr - p.x);
                a += *reinterpret_cast<const f32x4 *>(x2r + (ks0 + i) * 32);
                b += *reinterpret_cast<const f32x4 *>(x2r + (ks0 + i) * 32 + 4);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) { xv[i][j] = rok ? a[j] : 0.f; xv[i][4 + j] = rok ? b[j] : 0.f; }
        }
#pragma unroll
        for (int i = 0; i < NK; ++i)                     // the weight fragments are in flight under the LayerNorm statistics
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                wh[i][nt] = *reinterpret_cast<const bf16x8 *>(wbase[nt] + (long)(ks0 + i) * 2048);
                wl[i][nt] = *reinterpret_cast<const bf16x8 *>(wbase[nt] + (long)(ks0 + i) * 2048 + 1024);
            }
        // biased variance around the mean, eps inside the square root (F.layer_norm); the row is spread over 4 lanes x 4 waves
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NK; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) s += xv[i][j];
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        if (g == 0) stat[wave][r] = s;
        __syncthreads();
        const float mean = ((stat[0][r] + stat[1][r]) + (stat[2][r] + stat[3][r])) / (float)p.K;
        __syncthreads();
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < NK; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = xv[i][j] - mean; v += d * d; }
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (g == 0) stat[wave][r] = v;
        __syncthreads();
        const float rstd = rsqrtf(((stat[0][r] + stat[1][r]) + (stat[2][r] + stat[3][r])) / (float)p.K + p.eps);
#pragma unroll
        for (int i = 0; i < NK; ++i) {
            const float *wn = p.lnw + (ks0 + i) * 32 + g * 8;
            const f32x4 a = *reinterpret_cast<const f32x4 *>(wn), b = *reinterpret_cast<const f32x4 *>(wn + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                xv[i][j] = (xv[i][j] - mean) * rstd * a[j];
                xv[i][4 + j] = (xv[i][4 + j] - mean) * rstd * b[j];
            }
        }
#pragma unroll
        for (int i = 0; i < NK; ++i) product(xv[i], wh[i], wl[i]);
    } else {
        // K % 512 == 0: every wave walks its K quarter U k steps at a time (8 while that divides it: K = 1024 is then one round of
        // loads, K = 2048 two), all loads of a group issued before its first MFMA
        auto group = [&](int i0, auto uc) {
            constexpr int U = decltype(uc)::value;
            float xv[U][8];
            bf16x8 wh[U][NT], wl[U][NT];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int ks = ks0 + i0 + u;
                const f32x4 a = *reinterpret_cast<const f32x4 *>(xr + ks * 32), b = *reinterpret_cast<const f32x4 *>(xr + ks * 32 + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { xv[u][j] = rok ? a[j] : 0.f; xv[u][4 + j] = rok ? b[j] : 0.f; }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    wh[u][nt] = *reinterpret_cast<const bf16x8 *>(wbase[nt] + (long)ks * 2048);
                    wl[u][nt] = *reinterpret_cast<const bf16x8 *>(wbase[nt] + (long)ks * 2048 + 1024);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) product(xv[u], wh[u], wl[u]);
        };
        if (ksw % 8 == 0) for (int i0 = 0; i0 < ksw; i0 += 8) group(i0, std::integral_constant<int, 8>{});
        else for (int i0 = 0; i0 < ksw; i0 += 4) group(i0, std::integral_constant<int, 4>{});
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[wave][nt][lane][e] = acc[nt][e];
    __syncthreads();
    // D layout of the 16x16 MFMA: column = lane % 16 (feature), rows 4*(lane/16) + e
    for (int u = threadIdx.x; u < NT * 64; u += 256) {
        const int nt = u >> 6, l = u & 63;
        if (nt0 + nt >= p.n_tiles) continue;
        const int col = (nt0 + nt) * 16 + (l & 15);
        if (col >= p.n_out) continue;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int orow = blockIdx.y * 16 + 4 * (l >> 4) + e;
            if (orow >= p.rows) continue;
            float v = (red[0][nt][l][e] + red[1][nt][l][e]) + (red[2][nt][l][e] + red[3][nt][l][e]);
            v = gemm_activation(v, p.flags & 8);
            const long oi = (long)orow * p.ldo + col;
            if (kslice == 1) { p.side_out[oi] = v; continue; }         // the second half of K: alone, for the next launch to add
            float *o = p.out + oi;
            if (p.flags & 4) v += p.side_in ? *o + p.side_in[oi] : *o;
            *o = v;
        }
    }
}

// ---- both attentions of one decode step: blockIdx.z = 0 cross-attention over the cached memory keys (key-length mask, no rotary),
//      1 = self-attention: this step's k / v rounded to float16 and stored at cache position n_keys - 1, q rotated by that position,
//      cached keys by theirs (from the fp16 bytes, transformer.py:341-343).  One wave per (utterance, head, kind). ----
struct DecAttnArgs {
    const float *a;          // [N][4C]: cross query | self q | k | v
    long a_rs;
    int C, heads, hd;
    const __half *mem_k, *mem_v;     // [N][heads][S][hd]
    int S;
    const int *mem_len;
    __half *time_k, *time_v;         // [N][heads][Tc][hd]
    int Tc, n_keys;
    const float *cs, *sn;            // rotary tables [>= n_keys][hd/2]
    float scale;
    float *y;                        // [N][2C]: cross output | self output
    long y_rs;
};

// Eight lanes share a key (lane = 8 * (key within the pass) + dim chunk), each holding DPL = HD / 8 dims (whole rotary pairs): a pass of
// the wave scores eight keys (the partial dots meet by three xor shuffles) and, after the softmax, accumulates eight keys' p * v (the
// eight key groups meet by three more) -- one vector load per lane and pass for K and one for V, nothing serial over the keys.
template <int DPL>
__device__ __forceinline__ void load_halfs(const __half *src, float *out) {
    if constexpr (DPL == 2) {
        const __half2 v = *reinterpret_cast<const __half2 *>(src);
        out[0] = __low2float(v); out[1] = __high2float(v);
    } else {
        typedef unsigned uvec __attribute__((ext_vector_type(DPL == 4 ? 2 : 4)));
#pragma unroll
        for (int part = 0; part < (DPL == 16 ? 2 : 1); ++part) {
            const uvec raw = *reinterpret_cast<const uvec *>(src + part * 8);
#pragma unroll
            for (int w = 0; w < (DPL == 4 ? 2 : 4); ++w) {
                const unsigned u = raw[w];
                const __half2 v = *reinterpret_cast<const __half2 *>(&u);
                out[part * 8 + 2 * w] = __low2float(v);
                out[part * 8 + 2 * w + 1] = __high2float(v);
            }
        }
    }
}

template <int HD, int MAXK>
__global__ __launch_bounds__(64) void dec_attention_pair_kernel(const DecAttnArgs p) {
    constexpr int DPL = HD / 8, PPL = DPL / 2, HALF = HD / 2;
    __shared__ float ps[MAXK];
    const int h = blockIdx.x, n = blockIdx.y, self = blockIdx.z, lane = threadIdx.x;
    const int c = lane & 7, jl = lane >> 3;
    const float *qp = p.a + (long)n * p.a_rs + (self ? p.C : 0) + (long)h * HD;
    const int n_keys = self ? p.n_keys : p.S, Tc = self ? p.Tc : p.S;
    const int tq = n_keys - 1;
    const float *cs = self ? p.cs : nullptr, *sn = p.sn;
    float q[DPL];
#pragma unroll
    for (int i = 0; i < DPL; ++i) q[i] = qp[c * DPL + i];
    if (cs) {
#pragma unroll
        for (int i = 0; i < PPL; ++i) {
            const float cc = cs[(long)tq * HALF + c * PPL + i], sv = sn[(long)tq * HALF + c * PPL + i];
            const float r0 = q[2 * i] * cc + (-q[2 * i + 1]) * sv, r1 = q[2 * i + 1] * cc + q[2 * i] * sv;
            q[2 * i] = r0; q[2 * i + 1] = r1;
        }
    }
#pragma unroll
    for (int i = 0; i < DPL; ++i) q[i] *= p.scale;
    const long base = ((long)n * p.heads + h) * Tc * HD;
    const __half *kb = self ? p.time_k + base : p.mem_k + base;
    const __half *vb = self ? p.time_v + base : p.mem_v + base;
    // self: this step's key / value chunk of this lane (fp32 rows, rounded to fp16 as the cache holds them); lanes of key group 0 store it
    float kf[DPL], vf[DPL];
    if (self) {
#pragma unroll
        for (int i = 0; i < DPL; ++i) {
            const __half kh = __float2half(qp[p.C + c * DPL + i]), vh = __float2half(qp[2 * p.C + c * DPL + i]);
            kf[i] = __half2float(kh); vf[i] = __half2float(vh);
            if (jl == 0) {
                p.time_k[base + (long)tq * HD + c * DPL + i] = kh;
                p.time_v[base + (long)tq * HD + c * DPL + i] = vh;
            }
        }
    }
    const int klim = self ? n_keys : max(0, min(n_keys, p.mem_len ? p.mem_len[n] : n_keys));
    for (int j0 = 0; j0 < klim; j0 += 8) {
        const int j = j0 + jl;
        float sc = 0.f;
        if (j < klim) {
            float k[DPL];
            if (self && j == tq) {                                // this step's row: not from the store just issued
#pragma unroll
                for (int i = 0; i < DPL; ++i) k[i] = kf[i];
            } else {
                load_halfs<DPL>(kb + (long)j * HD + c * DPL, k);
            }
            if (cs) {
#pragma unroll
                for (int i = 0; i < PPL; ++i) {
                    const float cc = cs[(long)j * HALF + c * PPL + i], sv = sn[(long)j * HALF + c * PPL + i];
                    const float r0 = k[2 * i] * cc + (-k[2 * i + 1]) * sv, r1 = k[2 * i + 1] * cc + k[2 * i] * sv;
                    k[2 * i] = r0; k[2 * i + 1] = r1;
                }
            }
#pragma unroll
            for (int i = 0; i < DPL; ++i) sc = fmaf(q[i], k[i], sc);
        }
        sc += __shfl_xor(sc, 1, 64);
        sc += __shfl_xor(sc, 2, 64);
        sc += __shfl_xor(sc, 4, 64);
        if (c == 0 && j < klim) ps[j] = sc;
    }
    __syncthreads();
    float mx = -INFINITY;
    for (int j = lane; j < klim; j += 64) mx = fmaxf(mx, ps[j]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < klim; j += 64) {
        const float e = expf(ps[j] - mx);
        ps[j] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    __syncthreads();
    const float inv = 1.0f / sum;
    float acc[DPL];
#pragma unroll
    for (int i = 0; i < DPL; ++i) acc[i] = 0.f;
    for (int j0 = 0; j0 < klim; j0 += 8) {
        const int j = j0 + jl;
        if (j < klim) {
            float v[DPL];
            if (self && j == tq) {
#pragma unroll
                for (int i = 0; i < DPL; ++i) v[i] = vf[i];
            } else {
                load_halfs<DPL>(vb + (long)j * HD + c * DPL, v);
            }
            const float pj = ps[j];
#pragma unroll
            for (int i = 0; i < DPL; ++i) acc[i] = fmaf(pj, v[i], acc[i]);
        }
    }
    float *yp = p.y + (long)n * p.y_rs + (self ? p.C : 0) + (long)h * HD + c * DPL;
#pragma unroll
    for (int i = 0; i < DPL; ++i) {
        float v = acc[i];
        v += __shfl_xor(v, 8, 64);
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (jl == 0) yp[i] = v * inv;
    }
}

// ---- the cross-attention caches of ALL layers from one product (transformer.py:324-334, warmed once per decode): kv rows [N*S] hold,
//      for layer l, its memory keys at columns [l*2C, l*2C + C) and values at [l*2C + C, (l+1)*2C); caches [L][2][N][heads][S][hd] fp16
__global__ __launch_bounds__(256) void dec_memory_caches_kernel(const float *__restrict__ kv, long rs, __half *__restrict__ caches, int N,
                                                                int S, int heads, int hd, int L) {
    const long C = (long)heads * hd, per_layer = (long)N * S * 2 * C;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= per_layer * L) return;
    const int l = (int)(idx / per_layer);
    const long r = idx % per_layer;
    const long row = r / (2 * C);                         // n * S + s
    const int col = (int)(r % (2 * C)), which = col >= C, c = col - which * (int)C;
    const int n = (int)(row / S), sidx = (int)(row % S), h = c / hd, d = c % hd;
    const long plane = (long)N * C * S;
    caches[((long)l * 2 + which) * plane + (((long)n * heads + h) * S + sidx) * hd + d] = __float2half(kv[row * rs + (long)l * 2 * C + col]);
}

// ---- the greedy head of one step (transformer.py:175-192) and the next step's input: per row log_softmax max / argmax (first index
//      on ties) / sum p*logp/log 2, the alive-row bookkeeping (every alive row receives the entropy sum over ALL alive rows, sic),
//      then y[n] = wte[tokens[n, t + 1]].  Workgroup b owns rows [b*TOK_ROWS, +TOK_ROWS): their bookkeeping and embedding; the entropy
//      total needs every row, so each workgroup computes all N rows' statistics itself (N * V floats, L2 hits) rather than wait for
//      a neighbour.  alive is double-buffered ([2][N]; step t reads plane t & 1 and writes the other) so that no workgroup reads a flag
//      another one has already updated.  N <= 1024. ----
constexpr int TOK_ROWS = 4;
struct DecTokenArgs {
    const float *logits;
    long ld;
    int V, N;
    int64_t *tokens;
    long tok_ld;
    int t, plen, etx;
    uint8_t *alive;      // [2][N]
    int *out_len;
    float *log_probs, *sum_ent;
    const float *wte;
    int C, vocab;
    float *y;            // [N][C] (NULL: no embedding, the last step)
};

__global__ __launch_bounds__(256) void dec_token_kernel(const DecTokenArgs p) {
    __shared__ float s_val[1024], s_ne[1024];
    __shared__ int s_idx[1024];
    __shared__ float red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint8_t *alive_in = p.alive + (long)(p.t & 1) * p.N;
    uint8_t *alive_out = p.alive + (long)((p.t + 1) & 1) * p.N;
    if (p.V <= 1024) {
        // a small vocabulary: FOUR threads per row (columns q, q + 4, ...), 64 rows per pass of the workgroup; a wave per row leaves
        // most lanes idle and pays a 64-lane shuffle tree per row (23 us for 64 rows of 32)
        const int q = threadIdx.x & 3;
        for (int n0r = 0; n0r < p.N; n0r += 64) {
            const int n = n0r + (threadIdx.x >> 2);
            const float *row = p.logits + (long)min(n, p.N - 1) * p.ld;
            float m = -INFINITY;
            int am = 0x7fffffff;
            for (int c = q; c < p.V; c += 4) {
                const float v = row[c];
                if (v > m) { m = v; am = c; }
            }
#pragma unroll
            for (int o = 1; o <= 2; o <<= 1) {
                const float om = __shfl_xor(m, o, 64);
                const int oa = __shfl_xor(am, o, 64);
                if (om > m || (om == m && oa < am)) { m = om; am = oa; }
            }
            float s = 0.f;
            for (int c = q; c < p.V; c += 4) s += expf(row[c] - m);
            s += __shfl_xor(s, 1, 64);
            s += __shfl_xor(s, 2, 64);
            const float lse = m + logf(s);
            float e = 0.f;
            for (int c = q; c < p.V; c += 4) {
                const float lp = row[c] - lse;
                e += expf(lp) * lp / 0.6931471805599453f;
            }
            e += __shfl_xor(e, 1, 64);
            e += __shfl_xor(e, 2, 64);
            if (q == 0 && n < p.N) { s_val[n] = m - lse; s_idx[n] = am; s_ne[n] = e; }
        }
    } else
    for (int n = wave; n < p.N; n += 4) {
        const float *row = p.logits + (long)n * p.ld;
        float m = -INFINITY;
        int am = 0x7fffffff;
        for (int c = lane; c < p.V; c += 64) {
            const float v = row[c];
            if (v > m || (v == m && c < am)) { m = v; am = c; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float om = __shfl_xor(m, o, 64);
            const int oa = __shfl_xor(am, o, 64);
            if (om > m || (om == m && oa < am)) { m = om; am = oa; }
        }
        float s = 0.f;
        for (int c = lane; c < p.V; c += 64) s += expf(row[c] - m);
        const float lse = m + logf(wave_sum(s));
        float e = 0.f;
        for (int c = lane; c < p.V; c += 64) {
            const float lp = row[c] - lse;
            e += expf(lp) * lp / 0.6931471805599453f;
        }
        e = wave_sum(e);
        if (lane == 0) { s_val[n] = m - lse; s_idx[n] = am; s_ne[n] = e; }
    }
    __syncthreads();
    float e = 0.f;
    for (int n = threadIdx.x; n < p.N; n += 256)
        if (alive_in[n]) e += s_ne[n];
    e = wave_sum(e);
    if (lane == 0) red[wave] = e;
    __syncthreads();
    const float total = (red[0] + red[1]) + (red[2] + red[3]);
    const int n0 = blockIdx.x * TOK_ROWS;
    if (threadIdx.x < TOK_ROWS && n0 + threadIdx.x < p.N) {
        const int n = n0 + threadIdx.x;
        uint8_t al = alive_in[n];
        if (al) {
            p.sum_ent[n] += total;
            p.out_len[n] += 1;
            p.log_probs[n] += s_val[n];
            int64_t tok = s_idx[n];
            if (p.t < p.plen) tok = p.tokens[(long)n * p.tok_ld + p.t + 1];
            p.tokens[(long)n * p.tok_ld + p.t + 1] = tok;
            if (tok == p.etx) al = 0;
            s_idx[n] = (int)tok;
        } else {
            s_idx[n] = (int)p.tokens[(long)n * p.tok_ld + p.t + 1];
        }
        alive_out[n] = al;
    }
    if (!p.y) return;
    __syncthreads();
    const int C4 = p.C / 4;
    for (int u = threadIdx.x; u < TOK_ROWS * C4; u += 256) {
        const int n = n0 + u / C4, c = u % C4;
        if (n >= p.N) break;
        long id = s_idx[n];
        id = id < 0 ? 0 : (id >= p.vocab ? p.vocab - 1 : id);
        reinterpret_cast<f32x4 *>(p.y + (long)n * p.C)[c] = reinterpret_cast<const f32x4 *>(p.wte + id * p.C)[c];
    }
}

}  // namespace

extern "C" {

size_t halo_decode_image_bytes(int n_out, int k) {
    if (n_out <= 0 || k <= 0) return 0;
    return (size_t)((n_out + 15) / 16) * ((k + 31) / 32) * 2048;
}

int halo_decode_image(const float *weight, int n_out, int k, long ld, void *image, halo_stream_t stream) {
    HALO_CHECK_ARG(weight && image && n_out > 0 && k > 0 && ld >= k && (uintptr_t)image % 16 == 0);
    hipLaunchKernelGGL(dec_pack_kernel, dim3((k + 31) / 32, (n_out + 15) / 16), dim3(128), 0, (hipStream_t)stream, weight, n_out, k, ld,
                       (char *)image);
    return halo_launch_status();
}

int halo_decode_linear_supported(int k, int layernorm) {
    if (layernorm) return k == 512 || k == 768 || k == 1024;
    return k > 0 && k % 512 == 0;
}

int halo_decode_linear(const float *x, long ldx, int rows, int k, const float *ln_weight, float eps, const void *w_image, int n_out,
                       float *out, long ldo, int flags, halo_stream_t stream) {
    return halo_decode_linear_pair(x, nullptr, ldx, rows, k, ln_weight, eps, w_image, n_out, out, nullptr, nullptr, ldo, flags, stream);
}

int halo_decode_linear_pair(const float *x, const float *x_side, long ldx, int rows, int k, const float *ln_weight, float eps, const void *w_image,
                            int n_out, float *out, const float *side_in, float *side_out, long ldo, int flags, halo_stream_t stream) {
    HALO_CHECK_ARG(x && w_image && out && rows > 0 && n_out > 0 && ldx >= k && ldo >= n_out);
    // the side of the input rows: LayerNorm variants only; the K-sliced accumulate: no LayerNorm, ACCUM alone, whole k-steps per wave and slice
    HALO_CHECK_ARG(!x_side || ln_weight);
    HALO_CHECK_ARG(!side_out || (!ln_weight && flags == HALO_GEMM_ACCUM && k % 256 == 0 && side_out != out && side_out != side_in));
    HALO_CHECK_ARG(!side_in || side_out);
    HALO_CHECK_ARG(((uintptr_t)x_side) % 16 == 0);
    HALO_CHECK_ARG(halo_decode_linear_supported(k, ln_weight != nullptr));
    HALO_CHECK_ARG((flags & ~(HALO_GEMM_ACCUM | HALO_GEMM_GELU_ERF)) == 0);
    HALO_CHECK_ARG(((uintptr_t)x | (uintptr_t)w_image | (uintptr_t)ln_weight) % 16 == 0 && ldx % 4 == 0);
    DecLinearArgs p;
    p.x = x; p.ldx = ldx; p.rows = rows; p.K = k; p.lnw = ln_weight; p.eps = eps; p.w = (const char *)w_image;
    p.n_tiles = (n_out + 15) / 16; p.n_out = n_out; p.out = out; p.ldo = ldo; p.flags = flags;
    p.x2 = x_side; p.side_in = side_in; p.side_out = side_out;
    const int row_groups = (rows + 15) / 16, slices = side_out ? 2 : 1;
    // two feature tiles per workgroup while that still gives the chip a workgroup per CU, else one
    const bool two = (long)((p.n_tiles + 1) / 2) * row_groups * slices >= 256;
    const dim3 grid((unsigned)(two ? (p.n_tiles + 1) / 2 : p.n_tiles), (unsigned)row_groups, (unsigned)slices);
    hipStream_t st = (hipStream_t)stream;
#define HALO_DEC_LAUNCH(LN)                                                                     \
    do {                                                                                        \
        if (two) hipLaunchKernelGGL((dec_linear_kernel<2, LN>), grid, dim3(256), 0, st, p);     \
        else hipLaunchKernelGGL((dec_linear_kernel<1, LN>), grid, dim3(256), 0, st, p);         \
    } while (0)
    if (!ln_weight) HALO_DEC_LAUNCH(0);
    else if (k == 512) HALO_DEC_LAUNCH(4);
    else if (k == 768) HALO_DEC_LAUNCH(6);
    else HALO_DEC_LAUNCH(8);
#undef HALO_DEC_LAUNCH
    return halo_launch_status();
}

int halo_decode_attention_pair(const float *a, long a_row_stride, int N, int heads, int head_dim, const void *mem_k, const void *mem_v,
                               int S, const int *memory_lengths, void *time_k, void *time_v, int cache_len, int n_keys,
                               const float *cos_table, const float *sin_table, float *y, long y_row_stride, halo_stream_t stream) {
    HALO_CHECK_ARG(a && mem_k && mem_v && time_k && time_v && y && N > 0 && heads > 0 && S > 0);
    HALO_CHECK_ARG(head_dim == 16 || head_dim == 32 || head_dim == 64 || head_dim == 128);
    HALO_CHECK_ARG(n_keys >= 1 && n_keys <= cache_len && (!cos_table) == (!sin_table));
    const int C = heads * head_dim, maxk = n_keys > S ? n_keys : S;
    HALO_CHECK_ARG(a_row_stride >= 4L * C && y_row_stride >= 2L * C && maxk <= 8192);
    DecAttnArgs p;
    p.a = a; p.a_rs = a_row_stride; p.C = C; p.heads = heads; p.hd = head_dim;
    p.mem_k = (const __half *)mem_k; p.mem_v = (const __half *)mem_v; p.S = S; p.mem_len = memory_lengths;
    p.time_k = (__half *)time_k; p.time_v = (__half *)time_v; p.Tc = cache_len; p.n_keys = n_keys;
    p.cs = cos_table; p.sn = sin_table; p.scale = 1.0f / sqrtf((float)head_dim); p.y = y; p.y_rs = y_row_stride;
    const dim3 grid((unsigned)heads, (unsigned)N, 2);
    hipStream_t st = (hipStream_t)stream;
#define HALO_DEC_ATTN(HD)                                                                                   \
    do {                                                                                                    \
        if (maxk <= 1024) hipLaunchKernelGGL((dec_attention_pair_kernel<HD, 1024>), grid, dim3(64), 0, st, p); \
        else hipLaunchKernelGGL((dec_attention_pair_kernel<HD, 8192>), grid, dim3(64), 0, st, p);           \
    } while (0)
    if (head_dim == 64) HALO_DEC_ATTN(64);
    else if (head_dim == 128) HALO_DEC_ATTN(128);
    else if (head_dim == 32) HALO_DEC_ATTN(32);
    else HALO_DEC_ATTN(16);
#undef HALO_DEC_ATTN
    return halo_launch_status();
}

int halo_decode_memory_caches(const float *kv, long row_stride, int layers, void *caches, int N, int S, int heads, int head_dim,
                              halo_stream_t stream) {
    HALO_CHECK_ARG(kv && caches && layers > 0 && N > 0 && S > 0 && heads > 0 && head_dim > 0);
    HALO_CHECK_ARG(row_stride >= 2L * layers * heads * head_dim);
    const long n = (long)layers * N * S * 2 * heads * head_dim;
    hipLaunchKernelGGL(dec_memory_caches_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, kv, row_stride,
                       (__half *)caches, N, S, heads, head_dim, layers);
    return halo_launch_status();
}

int halo_decode_token(const float *logits, long ld, int N, int V, int64_t *tokens, long tokens_ld, int t, int plen, int etx,
                      uint8_t *alive, int *output_lengths, float *log_probs, float *sum_entropies, const float *wte, int vocab, int C,
                      float *y_next, halo_stream_t stream) {
    HALO_CHECK_ARG(logits && tokens && alive && output_lengths && log_probs && sum_entropies && N > 0 && N <= 1024 && V > 0 && ld >= V);
    HALO_CHECK_ARG(t >= 0 && tokens_ld > t + 1);
    HALO_CHECK_ARG(!y_next || (wte && vocab > 0 && C > 0 && C % 4 == 0 && ((uintptr_t)wte | (uintptr_t)y_next) % 16 == 0));
    DecTokenArgs p;
    p.logits = logits; p.ld = ld; p.V = V; p.N = N; p.tokens = tokens; p.tok_ld = tokens_ld; p.t = t; p.plen = plen; p.etx = etx;
    p.alive = alive; p.out_len = output_lengths; p.log_probs = log_probs; p.sum_ent = sum_entropies;
    p.wte = wte; p.C = C; p.vocab = vocab; p.y = y_next;
    hipLaunchKernelGGL(dec_token_kernel, dim3((unsigned)((N + TOK_ROWS - 1) / TOK_ROWS)), dim3(256), 0, (hipStream_t)stream, p);
    return halo_launch_status();
}

}  // extern "C"
