// Attention kernels for gfx950 shared by the GPT path (ha/attention.py) and the encoder-decoder ASR path
// (ha/transformer.py): general softmax attention forward (causal / key-length masked / plain, optional
// log-sum-exp and entropy outputs), interleaved rotary embedding, fp16 KV caches and the one-token
// decode attention over them, and the greedy head of Decoder.decode.
#include <hip/hip_fp16.h>
#include "halo_common.h"
#include "halo_internal.h"
#include "attn_args.h"

namespace {


// ---- softmax attention forward ---------------------------------------------------------------------
// One workgroup = 64 query rows of one (batch, head); 4 waves x 16 rows.  Keys/values stream through
// LDS in tiles of 64; S = Q K^T and O += P V run on v_mfma_f32_16x16x4_f32 (exact f32), softmax is the
// online (running max / running sum) form.  LDS images are chosen for conflict-free MFMA operand reads:
//   Ks [key][dim]  stride HD+2  (B operand of Q K^T: lane (key = l&15, dim-group = l>>4); 2*key + group is
//                                distinct over a 32-lane read group, and staging writes are 8-byte aligned)
//   Vs [key][dim]  stride HD+16 (B operand of P V  : lane (dim = l&15, key-group = l>>4))
//   Ps [row][key]  stride 66, per wave (P re-laid from the MFMA D layout to the A layout)
// Visibility: key j is seen by query i iff j < min(Tk, key_len[n]) and (not causal or j <= i + Tk - Tq).
// ENT: a second sweep over the keys accumulates the reference's attention-entropy monitor
// -sum_j att_j * log(att_j + 1e-8) per query row (ha/transformer.py:426).
template <int HD, bool ENT>
__global__ __launch_bounds__(256) void attention_fwd_kernel(AttnArgs a) {
    constexpr int KS_STRIDE = HD + 2, VS_STRIDE = HD + 16, PS_STRIDE = 66;
    __shared__ __attribute__((aligned(16))) float Ks[64 * KS_STRIDE];
    __shared__ __attribute__((aligned(16))) float Vs[64 * VS_STRIDE];
    __shared__ float Ps[4][16 * PS_STRIDE];
    // causal: query tile n-1-x (long) and then tile x (short): n + 1 key tiles per workgroup whichever x
    const int n_tiles_x = (a.Tq + 63) / 64, h = blockIdx.y, b = blockIdx.z;
    for (int pass = 0; pass < 2; ++pass) {
        const int qt = a.causal ? (pass == 0 ? n_tiles_x - 1 - (int)blockIdx.x : (int)blockIdx.x) : (int)blockIdx.x;
        if (pass == 1 && (!a.causal || qt == n_tiles_x - 1 - (int)blockIdx.x)) break;
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        const int lr = lane & 15, lq = lane >> 4;
        const int Tq = a.Tq, Tk = a.Tk;
        const float *qb = a.q + (long)b * a.q_bs + (long)h * a.q_hs;
        const float *kb = a.k + (long)b * a.kv_bs + (long)h * a.kv_hs;
        const float *vb = a.v + (long)b * a.kv_bs + (long)h * a.kv_hs;
        const int q0 = qt * 64 + wave * 16;                   // this wave's first query row
        const int klim = a.key_len ? max(0, min(Tk, a.key_len[b])) : Tk;
        const int coff = Tk - Tq;                             // causal: key <= row + coff

        // Q fragments, pre-scaled: A[row = lr][k = 4s + lq]
        float qa[HD / 4];
        {
            const int qrow = min(q0 + lr, Tq - 1);
            const float *qp = qb + (long)qrow * a.q_rs;
    #pragma unroll
            for (int s = 0; s < HD / 4; ++s) qa[s] = qp[4 * s + lq] * a.scale;
        }
        f32x4 o[HD / 16];
    #pragma unroll
        for (int m = 0; m < HD / 16; ++m) o[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        float mrow[4], lrow[4], erow[4];
    #pragma unroll
        for (int r = 0; r < 4; ++r) { mrow[r] = -INFINITY; lrow[r] = 0.f; erow[r] = 0.f; }

        int n_ktiles = (klim + 63) / 64;
        if (a.causal) n_ktiles = min(n_ktiles, max(0, (min(qt * 64 + 63, Tq - 1) + coff) / 64 + 1));
        // K/V tiles are fetched one tile ahead into registers (issue early, write to LDS late), so the global
        // latency of tile kt+1 hides under the MFMAs of tile kt
        constexpr int UNITS = 64 * (HD / 4) / 256;            // float4 units of K (and of V) per thread and tile
        f32x4 kreg[UNITS], vreg[UNITS];
        auto fetch = [&](int kt, bool with_v) {
    #pragma unroll
            for (int i = 0; i < UNITS; ++i) {
                const int u = threadIdx.x + 256 * i;
                const int key = u / (HD / 4), d4 = (u % (HD / 4)) * 4;
                const int krow = min(kt * 64 + key, Tk - 1);
                kreg[i] = *reinterpret_cast<const f32x4 *>(kb + (long)krow * a.kv_rs + d4);
                if (with_v) vreg[i] = *reinterpret_cast<const f32x4 *>(vb + (long)krow * a.kv_rs + d4);
            }
        };
        constexpr int N_PASS = ENT ? 2 : 1;
    #pragma unroll
        for (int pass = 0; pass < N_PASS; ++pass) {
            if (n_ktiles > 0) fetch(0, pass == 0);
            for (int kt = 0; kt < n_ktiles; ++kt) {
                __syncthreads();                              // previous tile fully consumed
    #pragma unroll
                for (int i = 0; i < UNITS; ++i) {
                    const int u = threadIdx.x + 256 * i;
                    const int key = u / (HD / 4), d4 = (u % (HD / 4)) * 4;
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    *reinterpret_cast<f32x2 *>(&Ks[key * KS_STRIDE + d4]) = f32x2{kreg[i][0], kreg[i][1]};
                    *reinterpret_cast<f32x2 *>(&Ks[key * KS_STRIDE + d4 + 2]) = f32x2{kreg[i][2], kreg[i][3]};
                    if (pass == 0) *reinterpret_cast<f32x4 *>(&Vs[key * VS_STRIDE + d4]) = vreg[i];
                }
                __syncthreads();
                if (kt + 1 < n_ktiles) fetch(kt + 1, pass == 0);
                // S = Q K^T : 4 key sub-tiles of 16
                f32x4 sacc[4];
    #pragma unroll
                for (int n = 0; n < 4; ++n) {
                    sacc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    #pragma unroll
                    for (int s = 0; s < HD / 4; ++s)
                        sacc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[s], Ks[(16 * n + lr) * KS_STRIDE + 4 * s + lq], sacc[n], 0, 0, 0);
                }
                if (pass == 1) {
                    // entropy sweep: att = exp(s - m) / l with the final m, l; element (row = 4*lq + r, key = kt*64 + 16n + lr)
    #pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int qrow = q0 + 4 * lq + r;
                        const float inv = 1.0f / lrow[r];
    #pragma unroll
                        for (int n = 0; n < 4; ++n) {
                            const int key = kt * 64 + 16 * n + lr;
                            const bool hidden = key >= klim || (a.causal && key > qrow + coff);
                            const float att = hidden ? 0.f : expf(sacc[n][r] - mrow[r]) * inv;
                            erow[r] -= att * logf(att + 1e-8f);
                        }
                    }
                    continue;
                }
                // mask + online softmax; element (row = 4*lq + r, key = kt*64 + 16n + lr)
                float alpha[4];
    #pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int qrow = q0 + 4 * lq + r;
                    float mx = -INFINITY;
    #pragma unroll
                    for (int n = 0; n < 4; ++n) {
                        const int key = kt * 64 + 16 * n + lr;
                        if (key >= klim || (a.causal && key > qrow + coff)) sacc[n][r] = -INFINITY;
                        mx = fmaxf(mx, sacc[n][r]);
                    }
                    mx = row16_max(mx);
                    const float mnew = fmaxf(mrow[r], mx);
                    const float msafe = mnew == -INFINITY ? 0.f : mnew;
                    alpha[r] = __expf(mrow[r] - msafe);       // exp(-inf) = 0 on the first tile (hardware exp2: ~1e-7 rel.)
                    float ps = 0.f;
    #pragma unroll
                    for (int n = 0; n < 4; ++n) {
                        const float pv = __expf(sacc[n][r] - msafe);
                        sacc[n][r] = pv;
                        ps += pv;
                    }
                    ps = row16_sum(ps);
                    lrow[r] = lrow[r] * alpha[r] + ps;
                    mrow[r] = mnew;
                    if (a.use_drop) {                          // the normaliser keeps the undropped sum (dropout acts on softmax's output)
                        const f32x4 dm = dropout_mult4(a.drop, attn_drop_tile_base(b, a.heads, h, Tq, min(qrow, Tq - 1), (Tk + 63) / 64, kt) + 4 * lr);
    #pragma unroll
                        for (int n = 0; n < 4; ++n) sacc[n][r] *= dm[n];
                    }
                }
                // P from the D layout to the A layout through this wave's LDS patch
                float *pw = Ps[wave];
    #pragma unroll
                for (int n = 0; n < 4; ++n)
    #pragma unroll
                    for (int r = 0; r < 4; ++r) pw[(4 * lq + r) * PS_STRIDE + 16 * n + lr] = sacc[n][r];
    #pragma unroll
                for (int m = 0; m < HD / 16; ++m)
    #pragma unroll
                    for (int r = 0; r < 4; ++r) o[m][r] *= alpha[r];
                __builtin_amdgcn_wave_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                // O += P V : 16 key steps of 4
    #pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const float pa = pw[lr * PS_STRIDE + 4 * s + lq];
    #pragma unroll
                    for (int m = 0; m < HD / 16; ++m)
                        o[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa, Vs[(4 * s + lq) * VS_STRIDE + 16 * m + lr], o[m], 0, 0, 0);
                }
            }
        }
        // y[b, row, h*HD + dim] = O / l
    #pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qrow = q0 + 4 * lq + r;
            if (qrow >= Tq) continue;
            const float inv = 1.0f / lrow[r];
            float *yp = a.y + (long)b * a.y_bs + (long)qrow * a.y_rs + (long)h * HD;
    #pragma unroll
            for (int m = 0; m < HD / 16; ++m) yp[16 * m + lr] = o[m][r] * inv;
            const long stat = ((long)b * a.heads + h) * Tq + qrow;
            if (a.lse && lr == 0) a.lse[stat] = mrow[r] + logf(lrow[r]);
            if (ENT) {
                float e = erow[r];
                e = row16_sum(e);
                if (lr == 0) a.ent[stat] = e;
            }
        }
    }
}

// ---- interleaved rotary embedding (ha/transformer.py:16-31) -----------------------------------------
// tables [T, hd/2]: angle(t, i) = t * base^(-2i/hd) evaluated as the reference does (fp32 pow, fp32 product)
__global__ __launch_bounds__(256) void rope_table_kernel(float *__restrict__ cs, float *__restrict__ sn, int T, int half, float base) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= T * half) return;
    const int t = idx / half, i = idx % half;
    const float ex = -2.0f * (float)i / (float)(2 * half);
    const float theta = (float)pow((double)base, (double)ex);
    const float ang = theta * (float)t;
    cs[idx] = (float)cos((double)ang);
    sn[idx] = (float)sin((double)ang);
}

// x[row, h*hd + 2i], x[.., 2i+1] rotated by the angle of position t0 + row % T, in place; sign = -1 undoes it
__global__ __launch_bounds__(256) void rope_apply_kernel(float *__restrict__ x, long row_stride, int n_rows, int T, int heads,
                                                         int half, int t0, const float *__restrict__ cs,
                                                         const float *__restrict__ sn, float sign) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long per_row = (long)heads * half;
    if (idx >= n_rows * per_row) return;
    const long row = idx / per_row;
    const int rem = (int)(idx % per_row), hh = rem / half, i = rem % half;
    const int t = t0 + (int)(row % T);
    const float c = cs[(long)t * half + i], s = sign * sn[(long)t * half + i];
    float *p = x + row * row_stride + (long)hh * 2 * half + 2 * i;
    const float x0 = p[0], x1 = p[1];
    p[0] = x0 * c + (-x1) * s;
    p[1] = x1 * c + x0 * s;
}

// ---- fp16 KV cache (ha/transformer.py:150-153, 315-339) -----------------------------------------------
// cache[n, h, t0 + s, :] = half(src[n*S + s, h*hd : (h+1)*hd]) for K and V at once; src rows hold k at col 0, v at col v_off
__device__ __forceinline__ void cache_put(__half *p, float v) { *p = __float2half(v); }
__device__ __forceinline__ void cache_put(float *p, float v) { *p = v; }

template <typename CT>
__global__ __launch_bounds__(256) void kv_cache_store_kernel(const float *__restrict__ src, long src_rs, long v_off,
                                                             CT *__restrict__ ck, CT *__restrict__ cv, int N, int S,
                                                             int heads, int hd, int Tc, int t0) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long C = (long)heads * hd;
    if (idx >= (long)N * S * C) return;
    const int c = (int)(idx % C);
    const long row = idx / C;
    const int s = (int)(row % S), n = (int)(row / S), h = c / hd, d = c % hd;
    const float *p = src + row * src_rs + c;
    const long o = (((long)n * heads + h) * Tc + (t0 + s)) * hd + d;
    cache_put(ck + o, p[0]);
    cache_put(cv + o, p[v_off]);
}

// One wave per (n, head): one query token against n_keys cached keys (fp16), optional rotary on the cached keys
// (positions 0..n_keys-1, rotated from the fp16 bytes exactly like transformer.py:341-343), optional key-length mask.
// knew / vnew (optional, rows like q): this step's key / value.  They are rounded to fp16, stored at cache position n_keys - 1
// (kv_cache[layer, :, alive, :, t0:t0+1, :] = k, v) and used from LDS, so the store, the query rotation (rope_q: by position
// n_keys - 1) and the attention of one decode step are ONE launch instead of three.
template <int MAXK>
__global__ __launch_bounds__(64) void attention_decode_kernel(const float *__restrict__ q, long q_rs, __half *__restrict__ ck,
                                                              __half *__restrict__ cv, float *__restrict__ y, long y_rs,
                                                              int heads, int hd, int Tc, int n_keys, const int *__restrict__ key_len,
                                                              const float *__restrict__ cs, const float *__restrict__ sn, float scale,
                                                              const float *__restrict__ knew, const float *__restrict__ vnew, int rope_q) {
    __shared__ float qs[128], kn[128], vn[128];
    __shared__ float ps[MAXK];
    const int h = blockIdx.x, n = blockIdx.y, lane = threadIdx.x;
    const int half = hd / 2;
    const float *qp = q + (long)n * q_rs + (long)h * hd;
    const int tq = n_keys - 1;                                   // the query's own position
    for (int i = lane; i < half; i += 64) {
        float q0 = qp[2 * i], q1 = qp[2 * i + 1];
        if (rope_q && cs) {
            const float c = cs[(long)tq * half + i], sv = sn[(long)tq * half + i];
            const float r0 = q0 * c + (-q1) * sv, r1 = q1 * c + q0 * sv;
            q0 = r0; q1 = r1;
        }
        qs[2 * i] = q0 * scale;
        qs[2 * i + 1] = q1 * scale;
    }
    __half *kb = ck + ((long)n * heads + h) * Tc * hd;
    __half *vb = cv + ((long)n * heads + h) * Tc * hd;
    if (knew) {
        const float *kp = knew + (long)n * q_rs + (long)h * hd, *vp = vnew + (long)n * q_rs + (long)h * hd;
        for (int d = lane; d < hd; d += 64) {
            const __half kh = __float2half(kp[d]), vh = __float2half(vp[d]);
            kb[(long)tq * hd + d] = kh;
            vb[(long)tq * hd + d] = vh;
            kn[d] = __half2float(kh);
            vn[d] = __half2float(vh);
        }
    }
    __syncthreads();
    const int klim = key_len ? max(0, min(n_keys, key_len[n])) : n_keys;
    float mx = -INFINITY;
    for (int j = lane; j < klim; j += 64) {
        const __half2 *kr = reinterpret_cast<const __half2 *>(kb + (long)j * hd);
        const bool fresh = knew && j == tq;                      // this step's row: from LDS, not from the store just issued
        float s = 0.f;
        for (int i = 0; i < half; ++i) {
            const float2 kk = fresh ? float2{kn[2 * i], kn[2 * i + 1]} : __half22float2(kr[i]);
            float k0 = kk.x, k1 = kk.y;
            if (cs) {
                const float c = cs[(long)j * half + i], sv = sn[(long)j * half + i];
                const float r0 = k0 * c + (-k1) * sv, r1 = k1 * c + k0 * sv;
                k0 = r0; k1 = r1;
            }
            s = fmaf(qs[2 * i], k0, s);
            s = fmaf(qs[2 * i + 1], k1, s);
        }
        ps[j] = s;
        mx = fmaxf(mx, s);
    }
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
    for (int d = lane; d < hd; d += 64) {
        float acc = 0.f;
        const int jcache = (knew && tq < klim) ? tq : klim;      // cached rows; the fresh row (always the last one) comes from LDS
        for (int j = 0; j < jcache; ++j) acc = fmaf(ps[j], __half2float(vb[(long)j * hd + d]), acc);
        if (knew && tq < klim) acc = fmaf(ps[tq], vn[d], acc);
        y[(long)n * y_rs + (long)h * hd + d] = acc * inv;
    }
}

// ---- greedy head of Decoder.decode (ha/transformer.py:175-192) -----------------------------------------
// per row: log_softmax, its max / argmax (first index on ties), and sum_v p*logp/log(2) (the negative entropy in bits)
__global__ __launch_bounds__(256) void logprob_max_kernel(const float *__restrict__ logits, long ld, int V, float *__restrict__ val,
                                                          int64_t *__restrict__ idx, float *__restrict__ negent) {
    __shared__ float redf[4];
    __shared__ int redi[4];
    const int n = blockIdx.x;
    const float *row = logits + (long)n * ld;
    float m = -INFINITY;
    int am = 0x7fffffff;
    for (int c = threadIdx.x; c < V; c += 256) {
        const float v = row[c];
        if (v > m || (v == m && c < am)) { m = v; am = c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(m, o, 64);
        const int oa = __shfl_xor(am, o, 64);
        if (om > m || (om == m && oa < am)) { m = om; am = oa; }
    }
    if ((threadIdx.x & 63) == 0) { redf[threadIdx.x >> 6] = m; redi[threadIdx.x >> 6] = am; }
    __syncthreads();
    m = redf[0]; am = redi[0];
#pragma unroll
    for (int w = 1; w < 4; ++w)
        if (redf[w] > m || (redf[w] == m && redi[w] < am)) { m = redf[w]; am = redi[w]; }
    __syncthreads();
    float s = 0.f;
    for (int c = threadIdx.x; c < V; c += 256) s += expf(row[c] - m);
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) redf[threadIdx.x >> 6] = s;
    __syncthreads();
    const float lse = m + logf((redf[0] + redf[1]) + (redf[2] + redf[3]));
    __syncthreads();
    float e = 0.f;
    if (negent) {
        for (int c = threadIdx.x; c < V; c += 256) {
            const float lp = row[c] - lse;
            e += expf(lp) * lp / 0.6931471805599453f;
        }
        e = wave_sum(e);
        if ((threadIdx.x & 63) == 0) redf[threadIdx.x >> 6] = e;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        val[n] = m - lse;
        idx[n] = am;
        if (negent) negent[n] = (redf[0] + redf[1]) + (redf[2] + redf[3]);
    }
}

// one workgroup: the bookkeeping of one greedy step over all N rows (alive rows only change)
__global__ __launch_bounds__(256) void greedy_update_kernel(const float *__restrict__ val, const int64_t *__restrict__ idx,
                                                            const float *__restrict__ negent, int64_t *__restrict__ tokens, long tok_ld,
                                                            int t, int plen, int etx, uint8_t *__restrict__ alive,
                                                            int *__restrict__ out_len, float *__restrict__ log_probs,
                                                            float *__restrict__ sum_ent, int N) {
    __shared__ float red[4];
    float e = 0.f;
    for (int n = threadIdx.x; n < N; n += 256)
        if (alive[n]) e += negent[n];
    e = wave_sum(e);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = e;
    __syncthreads();
    const float total = (red[0] + red[1]) + (red[2] + red[3]);     // sic: every alive row receives the sum over ALL alive rows
    for (int n = threadIdx.x; n < N; n += 256) {
        if (!alive[n]) continue;
        sum_ent[n] += total;
        out_len[n] += 1;
        log_probs[n] += val[n];
        int64_t tok = idx[n];
        if (t < plen) tok = tokens[(long)n * tok_ld + t + 1];
        tokens[(long)n * tok_ld + t + 1] = tok;
        if (tok == etx) alive[n] = 0;
    }
}

// ---- softmax attention backward --------------------------------------------------------------------------
// Two sweeps, no atomics (bitwise reproducible): attention_bwd_dq_kernel owns 64 query rows and walks the key
// tiles (dQ), attention_bwd_dkv_kernel owns 64 keys and walks the query tiles (dK, dV).  Both recompute
// P = exp(scale * q.k - lse) from the forward's log-sum-exp; delta = rowsum(dO * O) comes from attention_delta_kernel.
//   dV = P^T dO,  dP = dO V^T,  dS = P * (dP - delta),  dQ = scale * dS K,  dK = scale * dS^T Q
// All four/five products per tile run on v_mfma_f32_16x16x4_f32.  LDS tiles are [row][dim] with stride HD+2; an
// operand is read either as B[k = dim][col = row] (row = l&15) or as B[k = row][col = dim] (dim = l&15) -- the second
// pattern is 2-way bank-conflicted, which the 32-cycle f32 MFMA hides.

// delta[n, h, t] = sum_d dy[n, t, h*HD + d] * y[n, t, h*HD + d]
template <int HD>
__global__ __launch_bounds__(256) void attention_delta_kernel(const float *__restrict__ dy, long dy_rs, const float *__restrict__ y,
                                                              long y_rs, float *__restrict__ delta, int Tq, int heads) {
    const long row = blockIdx.x;                              // n * Tq + t
    const int n = (int)(row / Tq), t = (int)(row % Tq);
    const int C = heads * HD;
    for (int c = threadIdx.x; c < ((C + 255) / 256) * 256; c += 256) {
        float p = c < C ? dy[row * dy_rs + c] * y[row * y_rs + c] : 0.f;
#pragma unroll
        for (int o = HD / 2; o > 0; o >>= 1) p += __shfl_xor(p, o, 64);
        if (c < C && (c % HD) == 0) delta[((long)n * heads + c / HD) * Tq + t] = p;
    }
}

template <int HD>
__device__ __forceinline__ void stage_tile(float *lds, const f32x4 *reg) {
    constexpr int STRIDE = HD + 2, UNITS = 64 * (HD / 4) / 256;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int i = 0; i < UNITS; ++i) {
        const int u = threadIdx.x + 256 * i;
        const int row = u / (HD / 4), d4 = (u % (HD / 4)) * 4;
        *reinterpret_cast<f32x2 *>(&lds[row * STRIDE + d4]) = f32x2{reg[i][0], reg[i][1]};
        *reinterpret_cast<f32x2 *>(&lds[row * STRIDE + d4 + 2]) = f32x2{reg[i][2], reg[i][3]};
    }
}

template <int HD>
__device__ __forceinline__ void fetch_tile(f32x4 *reg, const float *base, long rs, int row0, int n_rows) {
    constexpr int UNITS = 64 * (HD / 4) / 256;
#pragma unroll
    for (int i = 0; i < UNITS; ++i) {
        const int u = threadIdx.x + 256 * i;
        const int row = min(row0 + u / (HD / 4), n_rows - 1), d4 = (u % (HD / 4)) * 4;
        reg[i] = *reinterpret_cast<const f32x4 *>(base + (long)row * rs + d4);
    }
}

template <int HD>
__global__ __launch_bounds__(256) void attention_bwd_dq_kernel(AttnBwdArgs a) {
    constexpr int ST = HD + 2, PS = 66, UNITS = 64 * (HD / 4) / 256;
    __shared__ __attribute__((aligned(16))) float Ks[64 * ST];
    __shared__ __attribute__((aligned(16))) float Vs[64 * ST];
    __shared__ float Ps[4][16 * PS];
    // causal: query tile n-1-x (long) and then tile x (short): n + 1 key tiles per workgroup whichever x
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
        const int klim = a.key_len ? max(0, min(Tk, a.key_len[b])) : Tk;
        const int coff = Tk - Tq;
        float qa[HD / 4], doa[HD / 4];
        {
            const int qrow = min(q0 + lr, Tq - 1);
    #pragma unroll
            for (int s = 0; s < HD / 4; ++s) {
                qa[s] = qb[(long)qrow * a.q_rs + 4 * s + lq] * a.scale;
                doa[s] = dyb[(long)qrow * a.dy_rs + 4 * s + lq];
            }
        }
        float lse_r[4], del_r[4];
    #pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long stat = ((long)b * a.heads + h) * Tq + min(q0 + 4 * lq + r, Tq - 1);
            lse_r[r] = a.lse[stat];
            del_r[r] = a.delta[stat];
        }
        f32x4 dq[HD / 16];
    #pragma unroll
        for (int m = 0; m < HD / 16; ++m) dq[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        int n_ktiles = (klim + 63) / 64;
        if (a.causal) n_ktiles = min(n_ktiles, max(0, (min(qt * 64 + 63, Tq - 1) + coff) / 64 + 1));
        f32x4 kreg[UNITS], vreg[UNITS];
        if (n_ktiles > 0) { fetch_tile<HD>(kreg, kb, a.kv_rs, 0, Tk); fetch_tile<HD>(vreg, vb, a.kv_rs, 0, Tk); }
        for (int kt = 0; kt < n_ktiles; ++kt) {
            __syncthreads();
            stage_tile<HD>(Ks, kreg);
            stage_tile<HD>(Vs, vreg);
            __syncthreads();
            if (kt + 1 < n_ktiles) { fetch_tile<HD>(kreg, kb, a.kv_rs, (kt + 1) * 64, Tk); fetch_tile<HD>(vreg, vb, a.kv_rs, (kt + 1) * 64, Tk); }
            f32x4 sacc[4], pacc[4];
    #pragma unroll
            for (int n = 0; n < 4; ++n) {
                sacc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
                pacc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    #pragma unroll
                for (int s = 0; s < HD / 4; ++s) {
                    sacc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[s], Ks[(16 * n + lr) * ST + 4 * s + lq], sacc[n], 0, 0, 0);
                    pacc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(doa[s], Vs[(16 * n + lr) * ST + 4 * s + lq], pacc[n], 0, 0, 0);
                }
            }
            float *pw = Ps[wave];
    #pragma unroll
            for (int n = 0; n < 4; ++n)
    #pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int qrow = q0 + 4 * lq + r, key = kt * 64 + 16 * n + lr;
                    const bool hidden = key >= klim || (a.causal && key > qrow + coff);
                    const float p = hidden ? 0.f : __expf(sacc[n][r] - lse_r[r]);
                    float dp = pacc[n][r];
                    if (a.use_drop)
                        dp *= dropout_mult(a.drop, attn_drop_tile_base(b, a.heads, h, Tq, min(qrow, Tq - 1), (Tk + 63) / 64, kt) + 4 * lr + n);
                    pw[(4 * lq + r) * PS + 16 * n + lr] = p * (dp - del_r[r]);
                }
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    #pragma unroll
            for (int s = 0; s < 16; ++s) {
                const float da = pw[lr * PS + 4 * s + lq];
    #pragma unroll
                for (int m = 0; m < HD / 16; ++m)
                    dq[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(da, Ks[(4 * s + lq) * ST + 16 * m + lr], dq[m], 0, 0, 0);
            }
        }
    #pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qrow = q0 + 4 * lq + r;
            if (qrow >= Tq) continue;
            float *dp = a.dq + (long)b * a.dq_bs + (long)qrow * a.dq_rs + (long)h * HD;
    #pragma unroll
            for (int m = 0; m < HD / 16; ++m) dp[16 * m + lr] = dq[m][r] * a.scale;
        }
    }
}

template <int HD>
__global__ __launch_bounds__(256) void attention_bwd_dkv_kernel(AttnBwdArgs a) {
    constexpr int ST = HD + 2, PS = 66, UNITS = 64 * (HD / 4) / 256;
    __shared__ __attribute__((aligned(16))) float Qs[64 * ST];
    __shared__ __attribute__((aligned(16))) float Os[64 * ST];      // dO tile
    __shared__ float Ps[4][2][16 * PS];
    __shared__ float lse_s[64], del_s[64];
    // causal: key tile x (long: it sees every later query tile) and then tile n-1-x (short), so every workgroup walks n + 1 query
    // tiles and the grid (ceil(n/2) wide) drains evenly (same pairing as csrc/attn_mx.hip)
    const int n_tiles_x = (a.Tk + 63) / 64, h = blockIdx.y, b = blockIdx.z;
    for (int pass = 0; pass < 2; ++pass) {
        const int kt = a.causal ? (pass == 0 ? (int)blockIdx.x : n_tiles_x - 1 - (int)blockIdx.x) : (int)blockIdx.x;
        if (pass == 1 && (!a.causal || kt == (int)blockIdx.x)) break;
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
        const int Tq = a.Tq, Tk = a.Tk;
        const float *qb = a.q + (long)b * a.q_bs + (long)h * HD;
        const float *dyb = a.dy + (long)b * a.dy_bs + (long)h * HD;
        const float *kb = a.k + (long)b * a.kv_bs + (long)h * HD;
        const float *vb = a.v + (long)b * a.kv_bs + (long)h * HD;
        const int k0 = kt * 64 + wave * 16;
        const int klim = a.key_len ? max(0, min(Tk, a.key_len[b])) : Tk;
        const int coff = Tk - Tq;
        float ka[HD / 4], va[HD / 4];
        {
            const int krow = min(k0 + lr, Tk - 1);
    #pragma unroll
            for (int s = 0; s < HD / 4; ++s) {
                ka[s] = kb[(long)krow * a.kv_rs + 4 * s + lq] * a.scale;
                va[s] = vb[(long)krow * a.kv_rs + 4 * s + lq];
            }
        }
        f32x4 dk[HD / 16], dv[HD / 16];
    #pragma unroll
        for (int m = 0; m < HD / 16; ++m) { dk[m] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[m] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        const int n_qtiles = (Tq + 63) / 64;
        const int qt0 = (a.causal && kt * 64 < klim) ? min(n_qtiles, max(0, kt * 64 - coff) / 64) : (kt * 64 < klim ? 0 : n_qtiles);
        const long stat0 = ((long)b * a.heads + h) * Tq;
        f32x4 qreg[UNITS], oreg[UNITS];
        if (qt0 < n_qtiles) { fetch_tile<HD>(qreg, qb, a.q_rs, qt0 * 64, Tq); fetch_tile<HD>(oreg, dyb, a.dy_rs, qt0 * 64, Tq); }
        for (int qt = qt0; qt < n_qtiles; ++qt) {
            __syncthreads();
            stage_tile<HD>(Qs, qreg);
            stage_tile<HD>(Os, oreg);
            if (threadIdx.x < 64) {
                const int qrow = min(qt * 64 + (int)threadIdx.x, Tq - 1);
                lse_s[threadIdx.x] = a.lse[stat0 + qrow];
                del_s[threadIdx.x] = a.delta[stat0 + qrow];
            }
            __syncthreads();
            if (qt + 1 < n_qtiles) { fetch_tile<HD>(qreg, qb, a.q_rs, (qt + 1) * 64, Tq); fetch_tile<HD>(oreg, dyb, a.dy_rs, (qt + 1) * 64, Tq); }
            f32x4 sacc[4], pacc[4];
    #pragma unroll
            for (int n = 0; n < 4; ++n) {
                sacc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
                pacc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    #pragma unroll
                for (int s = 0; s < HD / 4; ++s) {
                    sacc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[s], Qs[(16 * n + lr) * ST + 4 * s + lq], sacc[n], 0, 0, 0);
                    pacc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(va[s], Os[(16 * n + lr) * ST + 4 * s + lq], pacc[n], 0, 0, 0);
                }
            }
            float *pw = Ps[wave][0], *dw = Ps[wave][1];
    #pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int qi = 16 * n + lr, qrow = qt * 64 + qi;
                const float l = lse_s[qi], dl = del_s[qi];
    #pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = k0 + 4 * lq + r;
                    const bool hidden = key >= klim || qrow >= Tq || (a.causal && key > qrow + coff);
                    const float p = hidden ? 0.f : __expf(sacc[n][r] - l);
                    float dm = 1.0f;
                    if (a.use_drop)
                        dm = dropout_mult(a.drop, attn_drop_tile_base(b, a.heads, h, Tq, min(qrow, Tq - 1), (Tk + 63) / 64, kt) +
                                                      4 * (4 * lq + r) + wave);
                    pw[(4 * lq + r) * PS + qi] = p * dm;
                    dw[(4 * lq + r) * PS + qi] = p * (dm * pacc[n][r] - dl);
                }
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    #pragma unroll
            for (int s = 0; s < 16; ++s) {
                const float pa = pw[lr * PS + 4 * s + lq], da = dw[lr * PS + 4 * s + lq];
    #pragma unroll
                for (int m = 0; m < HD / 16; ++m) {
                    dv[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa, Os[(4 * s + lq) * ST + 16 * m + lr], dv[m], 0, 0, 0);
                    dk[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(da, Qs[(4 * s + lq) * ST + 16 * m + lr], dk[m], 0, 0, 0);
                }
            }
        }
    #pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = k0 + 4 * lq + r;
            if (key >= Tk) continue;
            float *kp = a.dk + (long)b * a.dkv_bs + (long)key * a.dkv_rs + (long)h * HD;
            float *vp = a.dv + (long)b * a.dkv_bs + (long)key * a.dkv_rs + (long)h * HD;
    #pragma unroll
            for (int m = 0; m < HD / 16; ++m) {
                kp[16 * m + lr] = dk[m][r] * a.scale;
                vp[16 * m + lr] = dv[m][r];
            }
        }
    }
}

template <int HD>
int launch_attention_bwd(const AttnBwdArgs &a, const float *y, long y_rs, float *delta, int N, hipStream_t st) {
    hipLaunchKernelGGL(attention_delta_kernel<HD>, dim3(N * a.Tq), dim3(256), 0, st, a.dy, a.dy_rs, y, y_rs, delta, a.Tq, a.heads);
    const int nq = (a.Tq + 63) / 64, nk = (a.Tk + 63) / 64;
    hipLaunchKernelGGL(attention_bwd_dq_kernel<HD>, dim3(a.causal ? (nq + 1) / 2 : nq, a.heads, N), dim3(256), 0, st, a);
    hipLaunchKernelGGL(attention_bwd_dkv_kernel<HD>, dim3(a.causal ? (nk + 1) / 2 : nk, a.heads, N), dim3(256), 0, st, a);
    return halo_launch_status();
}

template <int HD>
int launch_attention(const AttnArgs &a, int N, bool ent, hipStream_t st) {
    const int nq = (a.Tq + 63) / 64;
    dim3 grid(a.causal ? (nq + 1) / 2 : nq, a.heads, N);
    if (ent) hipLaunchKernelGGL((attention_fwd_kernel<HD, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((attention_fwd_kernel<HD, false>), grid, dim3(256), 0, st, a);
    return halo_launch_status();
}

}  // namespace

extern "C" {

int halo_attention_fwd_strided(const float *q, long q_row_stride, long q_batch_stride, long q_head_stride, const float *k,
                               const float *v, long kv_row_stride, long kv_batch_stride, long kv_head_stride, float *y,
                               long y_row_stride, long y_batch_stride, float *lse, float *entropy, int N, int heads, int head_dim,
                               int Tq, int Tk, int causal, const int *key_lengths, float p_drop, uint64_t seed, uint32_t stream_id,
                               uint32_t offset, const uint32_t *offset_dev, halo_stream_t stream) {
    HALO_CHECK_ARG(q && k && v && y && N > 0 && heads > 0 && Tq > 0 && Tk > 0);
    HALO_CHECK_ARG(((uintptr_t)k | (uintptr_t)v) % 16 == 0 && kv_row_stride % 4 == 0 && kv_batch_stride % 4 == 0 &&
                   kv_head_stride % 4 == 0);
    HALO_CHECK_ARG(N <= 65535 && heads <= 65535);
    AttnArgs a;
    a.q = q; a.k = k; a.v = v; a.y = y; a.lse = lse; a.ent = entropy;
    a.q_hs = q_head_stride; a.kv_hs = kv_head_stride;
    a.q_rs = q_row_stride; a.q_bs = q_batch_stride; a.kv_rs = kv_row_stride; a.kv_bs = kv_batch_stride;
    a.y_rs = y_row_stride; a.y_bs = y_batch_stride; a.key_len = key_lengths;
    a.Tq = Tq; a.Tk = Tk; a.heads = heads; a.causal = causal;
    a.scale = 1.0f / sqrtf((float)head_dim);
    a.drop = make_dropout(p_drop, seed, stream_id, offset, offset_dev);
    a.use_drop = p_drop > 0.f;
    HALO_CHECK_ARG(!(a.use_drop && entropy));       // the entropy monitor is an eval-time measurement
    hipStream_t st = (hipStream_t)stream;
    if (halo_math_mode() != HALO_MATH_F32 && !entropy && (head_dim == 64 || head_dim == 32)) {
        const int rc = halo_attention_fwd_mx(a, N, head_dim, halo_math_mode() == HALO_MATH_BF16 ? 1 : 3, st);
        if (rc != HALO_ENOTSUP) return rc;           // unaligned operands: the exact-f32 kernel takes them
    }
    switch (head_dim) {
        case 64: return launch_attention<64>(a, N, entropy != nullptr, st);
        case 32: return launch_attention<32>(a, N, entropy != nullptr, st);
        case 16: return launch_attention<16>(a, N, entropy != nullptr, st);
        default: return HALO_ENOTSUP;
    }
}

// Any boolean mask, any head dimension: ha/transformer.py:413-430 `attend(q, k, v, mask)` as the reference states it -- a mask of shape
// (N, ..., T, S) broadcast over what it lacks, True = the key is hidden from the query -- for call sites outside the models' shapes (the
// reference's own tests/test_attention.py calls it with head_dim 7 and a (T, S) triangle).  One wave per query row, exact fp32, no tiling:
// a slow path, kept off every model's forward (those run the tiled kernels through key lengths / the causal flag).
// q, k, v: [N, H, T|S, hd] contiguous; mask: bytes with element strides (0 = broadcast); y like q; ent [N, H, T] (may be NULL).
namespace {
__global__ __launch_bounds__(256) void attention_generic_kernel(const float *__restrict__ q, const float *__restrict__ k,
                                                                const float *__restrict__ v, const unsigned char *__restrict__ mask,
                                                                long m_sn, long m_sh, long m_st, float *__restrict__ y,
                                                                float *__restrict__ ent, int rows, int H, int T, int S, int hd, float scale) {
    extern __shared__ float dyn[];                       // per wave: [q: hd][p: S]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + wave;               // (n * H + h) * T + t
    if (row >= rows) return;
    float *qs = dyn + (long)wave * (hd + S), *ps = qs + hd;
    const int t = row % T, nh = row / T, h = nh % H, n = nh / H;
    const float *qr = q + (long)row * hd, *kb = k + (long)nh * S * hd, *vb = v + (long)nh * S * hd;
    for (int d = lane; d < hd; d += 64) qs[d] = qr[d] * scale;
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned char *mr = mask ? mask + n * m_sn + h * m_sh + t * m_st : nullptr;
    float m = -INFINITY;
    for (int j = lane; j < S; j += 64) {
        float s = 0.f;
        for (int d = 0; d < hd; ++d) s += qs[d] * kb[(long)j * hd + d];
        if (mr && mr[j]) s = -INFINITY;
        ps[j] = s;
        m = fmaxf(m, s);
    }
    m = wave_max(m);
    float l = 0.f;
    for (int j = lane; j < S; j += 64) {
        const float p = expf(ps[j] - m);                 // a row with every key hidden: exp(nan) -> nan, as softmax gives
        ps[j] = p;
        l += p;
    }
    l = wave_sum(l);
    float e = 0.f;
    for (int j = lane; j < S; j += 64) {
        const float att = ps[j] / l;
        ps[j] = att;
        e -= att * logf(att + 1e-8f);
    }
    e = wave_sum(e);
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int d = lane; d < hd; d += 64) {
        float acc = 0.f;
        for (int j = 0; j < S; ++j) acc += ps[j] * vb[(long)j * hd + d];
        y[(long)row * hd + d] = acc;
    }
    if (ent && lane == 0) ent[row] = e;
}
}  // namespace

int halo_attention_masked(const float *q, const float *k, const float *v, const unsigned char *mask, long mask_stride_n, long mask_stride_h,
                          long mask_stride_t, float *y, float *entropy, int N, int heads, int T, int S, int head_dim, halo_stream_t stream) {
    HALO_CHECK_ARG(q && k && v && y && N > 0 && heads > 0 && T > 0 && S > 0 && head_dim > 0);
    const size_t lds = (size_t)4 * (head_dim + S) * sizeof(float);
    if (lds > 160 * 1024 - 256 || (long)N * heads * T >= (1L << 31)) return HALO_ENOTSUP;
    static size_t opted = 0;
    if (lds > opted) {
        if (hipFuncSetAttribute((const void *)attention_generic_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return HALO_ELAUNCH;
        opted = lds;
    }
    const int rows = N * heads * T;
    hipLaunchKernelGGL(attention_generic_kernel, dim3((rows + 3) / 4), dim3(256), lds, (hipStream_t)stream, q, k, v, mask, mask_stride_n,
                       mask_stride_h, mask_stride_t, y, entropy, rows, heads, T, S, head_dim, 1.0f / sqrtf((float)head_dim));
    return halo_launch_status();
}

// the training forward on the matrix-core kernel with the output ALSO as row-major bf16 (the c_proj operand); packed rows (head stride = head_dim)
int halo_attention_fwd_bf16(const float *q, long q_row_stride, long q_batch_stride, const float *k, const float *v, long kv_row_stride,
                            long kv_batch_stride, float *y, long y_row_stride, long y_batch_stride, void *y_bf16, long ybf_row_stride,
                            long ybf_batch_stride, float *lse, int N, int heads, int head_dim, int Tq, int Tk, int causal,
                            const int *key_lengths, float p_drop, uint64_t seed, uint32_t stream_id, uint32_t offset,
                            const uint32_t *offset_dev, halo_stream_t stream) {
    HALO_CHECK_ARG(q && k && v && y && y_bf16 && N > 0 && heads > 0 && Tq > 0 && Tk > 0 && N <= 65535 && heads <= 65535);
    HALO_CHECK_ARG(((uintptr_t)k | (uintptr_t)v) % 16 == 0 && kv_row_stride % 4 == 0 && kv_batch_stride % 4 == 0);
    HALO_CHECK_ARG((uintptr_t)y_bf16 % 8 == 0 && ybf_row_stride % 4 == 0 && ybf_batch_stride % 4 == 0);
    if (halo_math_mode() == HALO_MATH_F32 || !(head_dim == 64 || head_dim == 32)) return HALO_ENOTSUP;
    AttnArgs a;
    a.q = q; a.k = k; a.v = v; a.y = y; a.lse = lse; a.ent = nullptr;
    a.q_hs = head_dim; a.kv_hs = head_dim;
    a.q_rs = q_row_stride; a.q_bs = q_batch_stride; a.kv_rs = kv_row_stride; a.kv_bs = kv_batch_stride;
    a.y_rs = y_row_stride; a.y_bs = y_batch_stride; a.key_len = key_lengths;
    a.y_bf = (__bf16 *)y_bf16; a.ybf_rs = ybf_row_stride; a.ybf_bs = ybf_batch_stride;
    a.Tq = Tq; a.Tk = Tk; a.heads = heads; a.causal = causal;
    a.scale = 1.0f / sqrtf((float)head_dim);
    a.drop = make_dropout(p_drop, seed, stream_id, offset, offset_dev);
    a.use_drop = p_drop > 0.f;
    return halo_attention_fwd_mx(a, N, head_dim, halo_math_mode() == HALO_MATH_BF16 ? 1 : 3, (hipStream_t)stream);
}

// the matrix-core backward with dq, dk, dv as row-major bf16 (one row / batch stride) instead of fp32
int halo_attention_bwd_bf16(const float *q, long q_row_stride, long q_batch_stride, const float *k, const float *v, long kv_row_stride,
                            long kv_batch_stride, const float *y, const float *dy, long y_row_stride, long y_batch_stride, const float *lse,
                            float *delta, void *dq_bf16, void *dk_bf16, void *dv_bf16, long d_row_stride, long d_batch_stride, int N,
                            int heads, int head_dim, int Tq, int Tk, int causal, const int *key_lengths, float p_drop, uint64_t seed,
                            uint32_t stream_id, uint32_t offset, const uint32_t *offset_dev, halo_stream_t stream) {
    HALO_CHECK_ARG(q && k && v && y && dy && lse && delta && dq_bf16 && dk_bf16 && dv_bf16 && N > 0 && heads > 0 && Tq > 0 && Tk > 0);
    HALO_CHECK_ARG(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)dy | (uintptr_t)y) % 16 == 0);
    HALO_CHECK_ARG(q_row_stride % 4 == 0 && q_batch_stride % 4 == 0 && kv_row_stride % 4 == 0 && kv_batch_stride % 4 == 0 &&
                   y_row_stride % 4 == 0 && y_batch_stride % 4 == 0);
    HALO_CHECK_ARG(N <= 65535 && heads <= 65535 && y_batch_stride == y_row_stride * Tq);
    if (halo_math_mode() == HALO_MATH_F32 || !(head_dim == 64 || head_dim == 32)) return HALO_ENOTSUP;
    AttnBwdArgs a;
    a.q = q; a.k = k; a.v = v; a.dy = dy; a.lse = lse; a.delta = delta; a.dq = nullptr; a.dk = nullptr; a.dv = nullptr;
    a.y = y; a.delta_w = delta;
    a.q_rs = q_row_stride; a.q_bs = q_batch_stride; a.kv_rs = kv_row_stride; a.kv_bs = kv_batch_stride;
    a.dy_rs = y_row_stride; a.dy_bs = y_batch_stride; a.dq_rs = 0; a.dq_bs = 0; a.dkv_rs = 0; a.dkv_bs = 0; a.key_len = key_lengths;
    a.dq_bf = (__bf16 *)dq_bf16; a.dk_bf = (__bf16 *)dk_bf16; a.dv_bf = (__bf16 *)dv_bf16; a.dqb_rs = d_row_stride; a.dqb_bs = d_batch_stride;
    a.Tq = Tq; a.Tk = Tk; a.heads = heads; a.causal = causal;
    a.scale = 1.0f / sqrtf((float)head_dim);
    a.drop = make_dropout(p_drop, seed, stream_id, offset, offset_dev);
    a.use_drop = p_drop > 0.f;
    return halo_attention_bwd_mx(a, N, head_dim, halo_math_mode() == HALO_MATH_BF16 ? 1 : 3, (hipStream_t)stream);
}

int halo_attention_fwd(const float *q, long q_row_stride, long q_batch_stride, const float *k, const float *v,
                       long kv_row_stride, long kv_batch_stride, float *y, long y_row_stride, long y_batch_stride, float *lse,
                       float *entropy, int N, int heads, int head_dim, int Tq, int Tk, int causal, const int *key_lengths,
                       halo_stream_t stream) {
    return halo_attention_fwd_strided(q, q_row_stride, q_batch_stride, head_dim, k, v, kv_row_stride, kv_batch_stride, head_dim, y,
                                      y_row_stride, y_batch_stride, lse, entropy, N, heads, head_dim, Tq, Tk, causal, key_lengths,
                                      0.f, 0, 0, 0, nullptr, stream);
}

int halo_attention_bwd(const float *q, long q_row_stride, long q_batch_stride, const float *k, const float *v, long kv_row_stride,
                       long kv_batch_stride, const float *y, const float *dy, long y_row_stride, long y_batch_stride, const float *lse,
                       float *delta, float *dq, long dq_row_stride, long dq_batch_stride, float *dk, float *dv, long dkv_row_stride,
                       long dkv_batch_stride, int N, int heads, int head_dim, int Tq, int Tk, int causal, const int *key_lengths,
                       float p_drop, uint64_t seed, uint32_t stream_id, uint32_t offset, const uint32_t *offset_dev,
                       halo_stream_t stream) {
    HALO_CHECK_ARG(q && k && v && y && dy && lse && delta && dq && dk && dv && N > 0 && heads > 0 && Tq > 0 && Tk > 0);
    HALO_CHECK_ARG(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)dy) % 16 == 0);
    HALO_CHECK_ARG(q_row_stride % 4 == 0 && q_batch_stride % 4 == 0 && kv_row_stride % 4 == 0 && kv_batch_stride % 4 == 0 &&
                   y_row_stride % 4 == 0 && y_batch_stride % 4 == 0);
    HALO_CHECK_ARG(N <= 65535 && heads <= 65535 && y_batch_stride == y_row_stride * Tq);
    AttnBwdArgs a;
    a.q = q; a.k = k; a.v = v; a.dy = dy; a.lse = lse; a.delta = delta; a.dq = dq; a.dk = dk; a.dv = dv; a.y = nullptr; a.delta_w = nullptr;
    a.q_rs = q_row_stride; a.q_bs = q_batch_stride; a.kv_rs = kv_row_stride; a.kv_bs = kv_batch_stride;
    a.dy_rs = y_row_stride; a.dy_bs = y_batch_stride; a.dq_rs = dq_row_stride; a.dq_bs = dq_batch_stride;
    a.dkv_rs = dkv_row_stride; a.dkv_bs = dkv_batch_stride; a.key_len = key_lengths;
    a.Tq = Tq; a.Tk = Tk; a.heads = heads; a.causal = causal;
    a.scale = 1.0f / sqrtf((float)head_dim);
    a.drop = make_dropout(p_drop, seed, stream_id, offset, offset_dev);
    a.use_drop = p_drop > 0.f;
    hipStream_t st = (hipStream_t)stream;
    if (halo_math_mode() != HALO_MATH_F32 && (head_dim == 64 || head_dim == 32)) {
        // the matrix-core sweeps: the dQ sweep computes delta = rowsum(dy * y) itself and leaves it in `delta` for the dK/dV sweep
        AttnBwdArgs m = a;
        if ((uintptr_t)y % 16 == 0) { m.y = y; m.delta_w = delta; }
        else if (head_dim == 64) hipLaunchKernelGGL(attention_delta_kernel<64>, dim3(N * Tq), dim3(256), 0, st, dy, y_row_stride, y, y_row_stride, delta, Tq, heads);
        else hipLaunchKernelGGL(attention_delta_kernel<32>, dim3(N * Tq), dim3(256), 0, st, dy, y_row_stride, y, y_row_stride, delta, Tq, heads);
        const int rc = halo_attention_bwd_mx(m, N, head_dim, halo_math_mode() == HALO_MATH_BF16 ? 1 : 3, st);
        if (rc != HALO_ENOTSUP) return rc;           // unaligned gradient views: the exact-f32 kernels take them
    }
    switch (head_dim) {
        case 64: return launch_attention_bwd<64>(a, y, y_row_stride, delta, N, st);
        case 32: return launch_attention_bwd<32>(a, y, y_row_stride, delta, N, st);
        case 16: return launch_attention_bwd<16>(a, y, y_row_stride, delta, N, st);
        default: return HALO_ENOTSUP;
    }
}

int halo_attention_causal_fwd(const float *qkv, float *y, int B, int T, int n_head, int C, halo_stream_t stream) {
    HALO_CHECK_ARG(qkv && y && B > 0 && T > 0 && n_head > 0 && C > 0 && C % n_head == 0 && C % 4 == 0);
    return halo_attention_fwd(qkv, 3L * C, 3L * C * T, qkv + C, qkv + 2 * C, 3L * C, 3L * C * T, y, C, (long)C * T, nullptr, nullptr, B,
                              n_head, C / n_head, T, T, 1, nullptr, stream);
}

int halo_rope_table(float *cos_table, float *sin_table, int T, int head_dim, float base, halo_stream_t stream) {
    HALO_CHECK_ARG(cos_table && sin_table && T > 0 && head_dim > 0 && head_dim % 2 == 0 && base > 0.f);
    const int n = T * (head_dim / 2);
    hipLaunchKernelGGL(rope_table_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, cos_table, sin_table, T,
                       head_dim / 2, base);
    return halo_launch_status();
}

int halo_rope_interleaved(float *x, long row_stride, int n_rows, int T, int heads, int head_dim, int t0, const float *cos_table,
                          const float *sin_table, int table_rows, int inverse, halo_stream_t stream) {
    HALO_CHECK_ARG(x && cos_table && sin_table && n_rows > 0 && T > 0 && heads > 0 && head_dim > 0 && head_dim % 2 == 0 && t0 >= 0);
    HALO_CHECK_ARG(t0 + T <= table_rows);
    const long n = (long)n_rows * heads * (head_dim / 2);
    hipLaunchKernelGGL(rope_apply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, row_stride, n_rows,
                       T, heads, head_dim / 2, t0, cos_table, sin_table, inverse ? -1.0f : 1.0f);
    return halo_launch_status();
}

int halo_kv_cache_store(const float *src, long src_row_stride, long v_offset, void *cache_k, void *cache_v, int N, int S, int heads,
                        int head_dim, int cache_len, int t0, halo_stream_t stream) {
    HALO_CHECK_ARG(src && cache_k && cache_v && N > 0 && S > 0 && heads > 0 && head_dim > 0 && t0 >= 0 && t0 + S <= cache_len);
    const long n = (long)N * S * heads * head_dim;
    hipLaunchKernelGGL(kv_cache_store_kernel<__half>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src,
                       src_row_stride, v_offset, (__half *)cache_k, (__half *)cache_v, N, S, heads, head_dim, cache_len, t0);
    return halo_launch_status();
}

int halo_kv_cache_store_f32(const float *src, long src_row_stride, long v_offset, float *cache_k, float *cache_v, int N, int S,
                            int heads, int head_dim, int cache_len, int t0, halo_stream_t stream) {
    HALO_CHECK_ARG(src && cache_k && cache_v && N > 0 && S > 0 && heads > 0 && head_dim > 0 && t0 >= 0 && t0 + S <= cache_len);
    const long n = (long)N * S * heads * head_dim;
    hipLaunchKernelGGL(kv_cache_store_kernel<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src,
                       src_row_stride, v_offset, cache_k, cache_v, N, S, heads, head_dim, cache_len, t0);
    return halo_launch_status();
}

int halo_attention_decode_step(const float *q, const float *k_new, const float *v_new, long row_stride, void *cache_k, void *cache_v,
                               float *y, long y_row_stride, int N, int heads, int head_dim, int cache_len, int n_keys,
                               const float *cos_table, const float *sin_table, halo_stream_t stream) {
    HALO_CHECK_ARG(q && k_new && v_new && cache_k && cache_v && y && N > 0 && heads > 0 && head_dim > 0 && head_dim % 2 == 0 &&
                   head_dim <= 128);
    HALO_CHECK_ARG(n_keys > 0 && n_keys <= cache_len && n_keys <= 8192 && N <= 65535 && (cos_table == nullptr) == (sin_table == nullptr));
    const float scale = 1.0f / sqrtf((float)head_dim);
    dim3 grid(heads, N);
    hipStream_t st = (hipStream_t)stream;
    if (n_keys <= 1024)
        hipLaunchKernelGGL(attention_decode_kernel<1024>, grid, dim3(64), 0, st, q, row_stride, (__half *)cache_k, (__half *)cache_v, y,
                           y_row_stride, heads, head_dim, cache_len, n_keys, (const int *)nullptr, cos_table, sin_table, scale, k_new,
                           v_new, 1);
    else
        hipLaunchKernelGGL(attention_decode_kernel<8192>, grid, dim3(64), 0, st, q, row_stride, (__half *)cache_k, (__half *)cache_v, y,
                           y_row_stride, heads, head_dim, cache_len, n_keys, (const int *)nullptr, cos_table, sin_table, scale, k_new,
                           v_new, 1);
    return halo_launch_status();
}

int halo_attention_decode(const float *q, long q_row_stride, const void *cache_k, const void *cache_v, float *y, long y_row_stride,
                          int N, int heads, int head_dim, int cache_len, int n_keys, const int *key_lengths, const float *cos_table,
                          const float *sin_table, halo_stream_t stream) {
    HALO_CHECK_ARG(q && cache_k && cache_v && y && N > 0 && heads > 0 && head_dim > 0 && head_dim % 2 == 0 && head_dim <= 128);
    HALO_CHECK_ARG(n_keys > 0 && n_keys <= cache_len && N <= 65535 && (cos_table == nullptr) == (sin_table == nullptr));
    const float scale = 1.0f / sqrtf((float)head_dim);
    dim3 grid(heads, N);
    hipStream_t st = (hipStream_t)stream;
    if (n_keys <= 1024)
        hipLaunchKernelGGL(attention_decode_kernel<1024>, grid, dim3(64), 0, st, q, q_row_stride, (__half *)cache_k,
                           (__half *)cache_v, y, y_row_stride, heads, head_dim, cache_len, n_keys, key_lengths, cos_table, sin_table,
                           scale, (const float *)nullptr, (const float *)nullptr, 0);
    else if (n_keys <= 8192)
        hipLaunchKernelGGL(attention_decode_kernel<8192>, grid, dim3(64), 0, st, q, q_row_stride, (__half *)cache_k,
                           (__half *)cache_v, y, y_row_stride, heads, head_dim, cache_len, n_keys, key_lengths, cos_table, sin_table,
                           scale, (const float *)nullptr, (const float *)nullptr, 0);
    else return HALO_ENOTSUP;
    return halo_launch_status();
}

int halo_logprob_max(const float *logits, long ld, int rows, int V, float *values, int64_t *indices, float *neg_entropy_bits,
                     halo_stream_t stream) {
    HALO_CHECK_ARG(logits && values && indices && rows > 0 && V > 0 && ld >= V);
    hipLaunchKernelGGL(logprob_max_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, logits, ld, V, values, indices,
                       neg_entropy_bits);
    return halo_launch_status();
}

int halo_greedy_update(const float *values, const int64_t *indices, const float *neg_entropy_bits, int64_t *tokens, long tokens_ld,
                       int t, int plen, int etx, uint8_t *alive, int *output_lengths, float *log_probs, float *sum_entropies, int N,
                       halo_stream_t stream) {
    HALO_CHECK_ARG(values && indices && neg_entropy_bits && tokens && alive && output_lengths && log_probs && sum_entropies);
    HALO_CHECK_ARG(N > 0 && t >= 0 && t + 1 < tokens_ld);
    hipLaunchKernelGGL(greedy_update_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, values, indices, neg_entropy_bits, tokens,
                       tokens_ld, t, plen, etx, alive, output_lengths, log_probs, sum_entropies, N);
    return halo_launch_status();
}

}  // extern "C"
