// The CTC head of the training step in three launches instead of thirteen (ha/recognizer.py:43-46,61-73):
//
//   halo_ctc_head_fwd   one workgroup per utterance: dropout(features) -> Linear(H -> V) -> log_softmax -> feature lengths
//                       (ha/rnn.py:13-18) -> CTC alpha recursion -> nll; the workgroup that finishes last adds up the mean loss
//   halo_ctc_head_bwd   one workgroup per utterance: CTC beta recursion -> gradient at the log-probs (ATen's convention) ->
//                       log_softmax backward -> d features (with the classifier dropout mask) and this utterance's partial
//                       d W, d bias
//   (reduce)            sums the per-utterance partials in a fixed order
//
// The products are tiny (T' x V x H per utterance: 21 x 32 x 1024) and run on the exact-f32 MFMA (v_mfma_f32_32x32x2_f32), like the
// halo_gemm_f32 launches they replace; the frames of one utterance are one 32-row tile, the classes one 32-column tile, so the fast
// path needs T' <= 32, V <= 32, 2S+1 <= 64 and H % 64 == 0 (the caller falls back to the separate operators otherwise).
#include <float.h>
#include "halo_common.h"
#include "halo_internal.h"
#include "lstm_persist_dev.h"
#include <type_traits>

namespace {

// in-kernel stamps of a measurement build (make EXTRA=-DHALO_HEAD_STAMPS; tools/head_stamps.py): [kernel][workgroup][point], 100 MHz clock
#ifdef HALO_HEAD_STAMPS
__device__ unsigned long long g_head_stamps[2][1024][16];
#define HST(k, i) do { if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 1024) g_head_stamps[k][blockIdx.x][i] = wall_clock64(); } while (0)
#else
#define HST(k, i) do { } while (0)
#endif

constexpr int NW = 16;                   // waves per workgroup: K-slices of the H-deep contraction / column tiles of the H-wide outputs
constexpr int LDT = 33;                  // padded leading dimension of the [32][32] tiles in LDS

__device__ __forceinline__ float log_add_exp_fast2(float a, float b) {       // as ctc.hip's wave kernels
    if (isinf(a) && a == b) return a;
    const float m = fmaxf(a, b);
    return m + __logf(1.0f + __expf(-fabsf(a - b)));
}
__device__ __forceinline__ int ext_label2(const int64_t *tg, int s) { return (s & 1) ? (int)tg[s >> 1] : 0; }
// log(e^a + e^b + e^c) the way ATen's CTC loss writes it (max, three exponentials, one log: LossCTC.cpp): the exponentials are
// independent, so a lattice step pays one exp and one log latency instead of two of each.  All -inf -> -inf.
__device__ __forceinline__ float log_add_exp_fast3(float a, float b, float c) {
    const float m = fmaxf(a, fmaxf(b, c));
    if (m == -INFINITY) return -INFINITY;
    return m + __logf(__expf(a - m) + __expf(b - m) + __expf(c - m));
}
// the whole wave shifted by one lane on the VALU (DPP wave_shr:1 / wave_shl:1) instead of through the LDS crossbar of a
// ds_bpermute-based __shfl_up / __shfl_down; lane 0 (63) keeps `x` itself, like the shuffles
__device__ __forceinline__ float wave_up1(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, x), __builtin_bit_cast(int, x), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_down1(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, x), __builtin_bit_cast(int, x), 0x130, 0xf, 0xf, false));
}

struct HeadFwdArgs {
    const float *feats, *w, *bias;
    DropoutCfg drop;
    const int64_t *il, *targets, *tl;
    long tg_stride;
    float *lp, *alpha, *nll, *grad_out, *loss;
    int64_t *flen;
    unsigned *ticket;
    // GREEDY instantiation (TemporalClassifier.decode, ha/recognizer.py:48-59): instead of the lattice, per frame the best class and its
    // log-prob, the collapsed hypothesis (repeats merged, blanks dropped, zero padded) and its length
    int64_t *ali, *hyp, *hyp_len;
    float *scores;
    int B, T, H, V, S, ks, stride, pad;
};

__device__ __forceinline__ int feature_length(long il, int ks, int stride, int pad, int T) {
    const float o = (float)(il + 2 * pad - ks);
    const int f = (int)floorf(o / (float)stride + 1.0f);          // float arithmetic like ha/rnn.py:13-18
    return max(0, min(f, T));
}

constexpr int KC = 256;                  // K-chunk staged through LDS per pass
constexpr int LDK = KC + 1;              // row stride in floats: lanes (row r, k) hit bank (r + k) % 32 -> conflict-free fragment reads

template <bool GREEDY>
__global__ __launch_bounds__(1024) void ctc_head_fwd_kernel(const HeadFwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) float dyn[];   // [As: 32 x LDK][Ws: 32 x LDK][red: NW x 1024]
    float *As = dyn, *Ws = dyn + 32 * LDK, *red = dyn + 64 * LDK;
    __shared__ float tile[32][LDT];                               // logits, then log-probs
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = p.T, H = p.H, V = p.V;
    const int r = lane & 31, kh = lane >> 5;
    // ---- logits[t][v] = sum_k drop(f[n,t,k]) W[v,k]: K in chunks of KC; rows are read from global memory as whole float4 runs
    //      (coalesced), the MFMA fragments (lane = row) come from LDS; wave w contracts k in [w * KC/NW, (w + 1) * KC/NW) of a chunk ----
    HST(0, 0);
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    // a chunk = 32 x KC/4 float4 units of the features and as many of W: two of each per thread.  The next chunk's loads are issued
    // before this chunk is written to LDS and multiplied, so their latency (which a chunk-by-chunk loop paid four times at H = 1024,
    // ~1.5 us each) runs under it
    constexpr int UPT = 32 * (KC / 4) / 1024;
    f32x4 ca[UPT], cb[UPT], na[UPT], nb[UPT];
    auto load_chunk = [&](int c0, f32x4 (&a)[UPT], f32x4 (&b)[UPT]) {
        const int kc = min(KC, H - c0);
#pragma unroll
        for (int i = 0; i < UPT; ++i) {
            const int u = tid + 1024 * i, row = u / (KC / 4), k4 = (u % (KC / 4)) * 4;
            a[i] = f32x4{0.f, 0.f, 0.f, 0.f}; b[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (k4 < kc) {
                if (row < T) a[i] = *reinterpret_cast<const f32x4 *>(p.feats + ((long)n * T + row) * H + c0 + k4);
                if (row < V) b[i] = *reinterpret_cast<const f32x4 *>(p.w + (long)row * H + c0 + k4);
            }
        }
    };
    load_chunk(0, ca, cb);
    for (int c0 = 0; c0 < H; c0 += KC) {
        const int kc = min(KC, H - c0);                          // H % 64 == 0: a multiple of 64
        if (c0 + KC < H) load_chunk(c0 + KC, na, nb);
#pragma unroll
        for (int i = 0; i < UPT; ++i) {
            const int u = tid + 1024 * i, row = u / (KC / 4), k4 = (u % (KC / 4)) * 4;
            f32x4 a = ca[i];
            if (p.drop.threshold && k4 < kc && row < T) a = a * dropout_mult4(p.drop, (uint64_t)(((long)n * T + row) * H + c0 + k4));
#pragma unroll
            for (int e = 0; e < 4; ++e) { As[row * LDK + k4 + e] = a[e]; Ws[row * LDK + k4 + e] = cb[i][e]; }
        }
        __syncthreads();
        const int kw = wave * (KC / NW);
#pragma unroll
        for (int s2 = 0; s2 < KC / NW / 2; ++s2) {
            const int k = kw + 2 * s2 + kh;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[r * LDK + k], Ws[r * LDK + k], acc, 0, 0, 0);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < UPT; ++i) { ca[i] = na[i]; cb[i] = nb[i]; }
    }
    HST(0, 1);
#pragma unroll
    for (int e = 0; e < 16; ++e) red[wave * 1024 + ((e & 3) + 8 * (e >> 2) + 4 * kh) * 32 + r] = acc[e];
    __syncthreads();
    {
        const int i = tid >> 5, j = tid & 31;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) s += red[w * 1024 + tid];
        tile[i][j] = (j < V) ? s + p.bias[j] : -INFINITY;
    }
    __syncthreads();
    // ---- log-softmax over the classes: 32 lanes per frame ----
    {
        const int i = tid >> 5, j = tid & 31;
        const float x = tile[i][j];
        float m = x;
#pragma unroll
        for (int d = 16; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 32));
        float s = j < V ? expf(x - m) : 0.f;
#pragma unroll
        for (int d = 16; d >= 1; d >>= 1) s += __shfl_xor(s, d, 32);
        const float lpv = x - m - logf(s);
        __syncthreads();
        tile[i][j] = lpv;
        if (p.lp && i < T && j < V) p.lp[((long)n * T + i) * V + j] = lpv;
    }
    __syncthreads();
    HST(0, 2);
    if (GREEDY) {
        // one lane per frame (T <= 32): arg max over the classes (first maximum, like torch.max), then unique_consecutive + drop blanks
        if (wave == 0) {
            const int t = lane;
            int best = -1;
            float bv = -INFINITY;
            if (t < T) {
                best = 0; bv = tile[t][0];
                for (int c = 1; c < V; ++c) {
                    const float v = tile[t][c];
                    if (v > bv) { bv = v; best = c; }
                }
                p.ali[(long)n * T + t] = best;
                p.scores[(long)n * T + t] = bv;
            }
            int left = __shfl_up(best, 1, 64);
            if (lane == 0) left = -1;
            const bool keep = (t < T) && best != left && best != 0;
            const unsigned long long mask = __ballot(keep);
            const int pos = __popcll(mask & ((1ull << lane) - 1ull)), count = __popcll(mask);
            if (keep) p.hyp[(long)n * T + pos] = best;
            if (t < T && t >= count) p.hyp[(long)n * T + t] = 0;          // the padding the separate fill launch wrote
            if (lane == 0) p.hyp_len[n] = count;
        }
        return;
    }
    // ---- CTC alpha (F.ctc_loss semantics, as ctc_alpha_wave_kernel with flags 0), wave 0: one lattice state per lane ----
    if (wave == 0) {
        const int S_ = 2 * p.S + 1;
        const int64_t *tg = p.targets + (long)n * p.tg_stride;
        const int il = feature_length(p.il[n], p.ks, p.stride, p.pad, T);
        const int tl = max(0, min((int)p.tl[n], p.S));
        const int states = 2 * tl + 1;
        float *alpha = p.alpha + (long)n * T * S_;
        const int lab = lane < S_ ? ext_label2(tg, lane) : 0;
        const int lab2 = __shfl_up(lab, 2, 64);
        const bool can_skip = lane >= 2 && lab != 0 && lab != lab2;
        const int tlast = il - 1, slast = 2 * tl, sprev = tl > 0 ? 2 * tl - 1 : -1;
        float prev = (il > 0 && lane < states && lane < 2) ? tile[0][lab] : -INFINITY;
        if (lane < S_) alpha[lane] = prev;
        float ra = 0.f, rb = 0.f;
        if (tlast == 0) { ra = __shfl(prev, slast, 64); rb = sprev >= 0 ? __shfl(prev, sprev, 64) : -INFINITY; }
        float em = tile[min(1, T - 1)][lab];                 // the emission of the next frame: fetched ahead of the dependent chain
        for (int t = 1; t < T; ++t) {
            const float p1 = wave_up1(prev), p2 = wave_up1(p1);
            const float e = em;
            em = tile[min(t + 1, T - 1)][lab];
            float v = -INFINITY;
            if (t < il && lane < states) {
                if (lane == 0) v = prev + e;
                else v = log_add_exp_fast3(prev, p1, (lane >= 2 && can_skip) ? p2 : -INFINITY) + e;
            }
            prev = v;
            if (lane < S_) alpha[(long)t * S_ + lane] = v;
            if (t == tlast) { ra = __shfl(prev, slast, 64); rb = sprev >= 0 ? __shfl(prev, sprev, 64) : -INFINITY; }
        }
        HST(0, 3);
        unsigned old = 0;
        if (lane == 0) {
            const float out = il == 0 ? (tl == 0 ? 0.f : INFINITY) : -log_add_exp_fast2(ra, rb);
            // nll travels to the workgroup that finishes last: write-through store, drained, then the ticket (Guideline 16 R1)
            __hip_atomic_store(p.nll + n, out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            p.flen[n] = il;
            p.grad_out[n] = 1.0f / (fmaxf((float)p.tl[n], 1.0f) * (float)p.B);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            old = __hip_atomic_fetch_add(p.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        old = __builtin_amdgcn_readfirstlane(old);
        HST(0, 4);
        if (old == (unsigned)p.B - 1u) {
            // reduction='mean' (ha/recognizer.py:71): mean_n(nll[n] / max(tl[n], 1)); the whole wave loads (one utterance per lane and
            // pass), partial sums combine in a fixed order
            float s = 0.f;
            for (int i0 = 0; i0 < p.B; i0 += 64) {
                const int i = i0 + lane;
                float v = 0.f;
                if (i < p.B) v = __hip_atomic_load(p.nll + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) / fmaxf((float)p.tl[i], 1.0f);
                s += wave_sum(v);
            }
            if (lane == 0) {
                *p.loss = s / (float)p.B;
                __hip_atomic_store(p.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next call
            }
        }
    }
}

constexpr size_t HEAD_FWD_LDS = (size_t)(64 * LDK + NW * 1024) * sizeof(float);

struct HeadBwdArgs {
    const float *feats, *w;
    DropoutCfg drop;
    const int64_t *flen, *targets, *tl;
    long tg_stride;
    const float *lp, *alpha, *nll, *grad_out;
    float *dfeats;          // [B,T,H]
    float *dw_part;         // [B,V,H]
    float *db_part;         // [B,V]
    int B, T, H, V, S;
};

constexpr int NWB = 8;                   // waves of the backward workgroup (512 threads: 256 VGPRs per lane, no spills)
__global__ __launch_bounds__(512) void ctc_head_bwd_kernel(const HeadBwdArgs p) {
    __shared__ float lps[32][LDT];        // log-probs
    __shared__ float dl[32][LDT];         // gradient at the logits (0 outside [T) x [V))
    __shared__ float ab[32][65];          // alpha, then alpha + beta
    extern __shared__ __attribute__((aligned(16))) float mask_s[];     // [T][H] dropout multipliers of this utterance's features
    // blockIdx.y: which slice of the H-wide outputs (d features, dW) this workgroup produces.  The lattice part (beta, gradient at the
    // logits: ~10 us of latency-bound work) is repeated by every slice of an utterance; the two products and the dropout masks -- the
    // bulk -- are divided, and B utterances fill B * slices CUs instead of B
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = p.T, H = p.H, V = p.V, S_ = 2 * p.S + 1;
    const int Hs = H / (int)gridDim.y, h0 = (int)blockIdx.y * Hs;       // this slice: columns [h0, h0 + Hs), Hs % 32 == 0
    HST(1, 0);
    if (p.drop.threshold) {               // one Philox call covers four consecutive elements
        for (int u = tid; u < T * Hs / 4; u += 512) {
            const int row = (4 * u) / Hs, col = (4 * u) % Hs;
            *reinterpret_cast<f32x4 *>(mask_s + 4 * u) = dropout_mult4(p.drop, (uint64_t)(((long)n * T + row) * H + h0 + col));
        }
    }
    // the operands of this wave's first column tile of the two products (W's and the features' 32 columns) are requested now: they
    // do not depend on the lattice work below, which hides their latency
    const int r = lane & 31, kh = lane >> 5;
    float wv0[16], fv0[16];
    {
        const int c0 = h0 + wave * 32;
        const bool mine = wave < Hs / 32;
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) {
            const int kk = 2 * s2 + kh;
            wv0[s2] = (mine && kk < V) ? p.w[(long)kk * H + c0 + r] : 0.f;
            fv0[s2] = (mine && kk < T) ? p.feats[((long)n * T + kk) * H + c0 + r] : 0.f;
        }
    }
    const int64_t *tg = p.targets + (long)n * p.tg_stride;
    const int il = max(0, min((int)p.flen[n], T));
    const int tl = max(0, min((int)p.tl[n], p.S));
    const int states = 2 * tl + 1;
    for (int u = tid; u < 1024; u += 512) {
        const int i = u >> 5, j = u & 31;
        lps[i][j] = (i < T && j < V) ? p.lp[((long)n * T + i) * V + j] : -INFINITY;
        dl[i][j] = 0.f;
    }
    for (int idx = tid; idx < T * S_; idx += 512) ab[idx / S_][idx % S_] = p.alpha[(long)n * T * S_ + idx];
    __syncthreads();
    HST(1, 1);
    const float nll = p.nll[n], go = p.grad_out[n];
    if (wave == 0 && il > 0) {            // beta recursion, one state per lane (as ctc_beta_grad_wave_kernel)
        const int lab = lane < S_ ? ext_label2(tg, lane) : 0;
        const int labn2 = __shfl_down(lab, 2, 64);
        const bool can_skip = lane + 2 < states && labn2 != 0 && labn2 != lab;
        float nxt = -INFINITY;
        float em = lps[il - 1][lab];
        for (int t = il - 1; t >= 0; --t) {
            const float n1 = wave_down1(nxt), n2 = wave_down1(n1);
            const float e = em;
            em = lps[max(t - 1, 0)][lab];
            float v = -INFINITY;
            if (lane < states) {
                if (t == il - 1) {
                    if (lane == states - 1 || lane == states - 2) v = e;
                } else {
                    v = log_add_exp_fast3(nxt, lane + 1 < states ? n1 : -INFINITY, can_skip ? n2 : -INFINITY) + e;
                }
                ab[t][lane] += v;
            }
            nxt = v;
        }
    }
    __syncthreads();
    HST(1, 2);
    for (int u = tid; u < 1024; u += 512) {   // gradient at the log-probs in ATen's convention, then log_softmax backward: dlogit = g - exp(lp) * sum_c g
        const int t = u >> 5, c = u & 31;
        float g = 0.f;
        if (t < il && c < V) {
            const float *row = ab[t];
            float m = -INFINITY;
            if (c == 0) { for (int s = 0; s < states; s += 2) m = fmaxf(m, row[s]); }
            else        { for (int s = 1; s < states; s += 2) if ((int)tg[s >> 1] == c) m = fmaxf(m, row[s]); }
            float lcab = -INFINITY;
            if (m > -INFINITY) {
                float sum = 0.f;
                if (c == 0) { for (int s = 0; s < states; s += 2) sum += __expf(row[s] - m); }
                else        { for (int s = 1; s < states; s += 2) if ((int)tg[s >> 1] == c) sum += __expf(row[s] - m); }
                lcab = m + __logf(sum);
            }
            const float l = lps[t][c];
            g = (__expf(l) - __expf(lcab + nll - l)) * go;
        }
        float gs = g;
#pragma unroll
        for (int d = 16; d >= 1; d >>= 1) gs += __shfl_xor(gs, d, 32);
        if (t < T && c < V) dl[t][c] = g - expf(lps[t][c]) * gs;
    }
    __syncthreads();
    HST(1, 3);
    if (tid < V && blockIdx.y == 0) {
        float s = 0.f;
        for (int t = 0; t < T; ++t) s += dl[t][tid];
        p.db_part[(long)n * V + tid] = s;
    }
    // ---- the two products, column tiles of 32 over H: wave w takes tiles w, w + NW, ... ----
    for (int nt = wave; nt < Hs / 32; nt += NWB) {
        const int c0 = h0 + nt * 32, m0 = nt * 32;                      // column of the tile in the row / in the slice's mask
        {   // d features[t][c0 + j] = sum_v dl[t][v] * W[v][c0 + j], times the dropout mask of the element
            float wv[16];
#pragma unroll
            for (int s = 0; s < 16; ++s) wv[s] = nt == wave ? wv0[s] : ((2 * s + kh) < V ? p.w[(long)(2 * s + kh) * H + c0 + r] : 0.f);
            f32x16 dx;
#pragma unroll
            for (int e = 0; e < 16; ++e) dx[e] = 0.f;
#pragma unroll
            for (int s = 0; s < 16; ++s) dx = __builtin_amdgcn_mfma_f32_32x32x2f32(dl[r][2 * s + kh], wv[s], dx, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * kh;
                if (row < T) {
                    float v = dx[e];
                    if (p.drop.threshold) v *= mask_s[row * Hs + m0 + r];
                    p.dfeats[((long)n * T + row) * H + c0 + r] = v;
                }
            }
        }
        {   // d W[v][c0 + j] (this utterance's share) = sum_t dl[t][v] * dropped features[t][c0 + j]
            float fv[16];
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int kk = 2 * s + kh;
                float f = nt == wave ? fv0[s] : (kk < T ? p.feats[((long)n * T + kk) * H + c0 + r] : 0.f);
                if (p.drop.threshold && kk < T) f *= mask_s[kk * Hs + m0 + r];
                fv[s] = f;
            }
            f32x16 dwp;
#pragma unroll
            for (int e = 0; e < 16; ++e) dwp[e] = 0.f;
#pragma unroll
            for (int s = 0; s < 16; ++s) dwp = __builtin_amdgcn_mfma_f32_32x32x2f32(dl[2 * s + kh][r], fv[s], dwp, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * kh;
                if (row < V) p.dw_part[((long)n * V + row) * H + c0 + r] = dwp[e];
            }
        }
    }
    HST(1, 4);
}

// ------------------------------------------------------------------------------------------------------------------------------
// The head of a TRAINING step in ONE launch (halo_ctc_head_train; the two launches above stay for the exact-f32 mode and for callers
// that want log-probs and alpha).  What the stamps of those two launches showed (tools/head_stamps.py, B = 64, H = 1024: 21 + 21 us):
// the classifier product 12.5 us (4 x 880-cycle Philox draws per wave in front of 512 exact-f32 MFMAs, on 64 of the 256 CUs), the two
// lattice recursions 4.9 + 4.2 us one after the other, 5.8 us for the gradient at the logits (target labels fetched from global memory
// inside its loops), 4.4 us restaging log-probs / alpha / masks in the second launch.  Here:
//   * grid (B, SL): the SL workgroups of an utterance each take H/SL feature columns -- their share of the dropout mask (kept in LDS as
//     bytes), of the logits' K range, and later those columns of d features / d W.  The partial logits [32 x 32] meet through global
//     memory as 8-byte (value, launch count) pairs, ONE write-through store each (MI355X_MICROARCH.md's data-tagged granule: a reader
//     that sees this launch's count has the value; no drain, no counter, nothing to reset); every workgroup of the utterance then sums
//     the SL partials in slice order and runs the (single-wave) lattice work itself.  B * SL <= the device's CU count, one workgroup
//     per CU: all resident; every wait bounded (0.2 s), and a wait given up raises the caller's status word and still takes its loss
//     ticket, so the launch count advances and the next launch starts clean.
//   * products on split-bf16 MFMA (v_mfma_f32_32x32x16_bf16; hi*hi + hi*lo + lo*hi: fp32-grade like every `bf16x3` product) with the
//     fragments loaded straight from global memory in MFMA layout -- no LDS staging, no barrier until the K-slices are added up.
//   * alpha in wave 0 and beta in wave 1 AT THE SAME TIME; labels, log-probs, alpha + beta stay in LDS.
struct HeadTrainArgs {
    const float *feats, *w, *bias;
    DropoutCfg drop;
    const int64_t *il, *targets, *tl;
    long tg_stride;
    float *lp, *nll, *loss;
    int64_t *flen;
    unsigned *ticket;        // [0] the loss ticket, [1] launches so far, [2 ..) the slices' (partial logit, epoch) pairs; zero before the first launch
    unsigned *status;        // optional caller-owned sticky word (halo_set_status_word): set to 1 when the slice exchange gives up its bounded wait
    int mute;                // test hook (halo_debug_mute_workgroup): the workgroup with this linear index (n * slices + y) never publishes
    float *dfeats, *dw_part, *db_part;
    int B, T, H, V, S, ks, stride, pad;
};

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
constexpr int TNW = 8, TNT = 64 * TNW;    // 512 threads: 256 VGPRs per lane hold two passes of fragments and two prefetched product tiles

__global__ __launch_bounds__(TNT) void ctc_head_train_kernel(const HeadTrainArgs p) {
    extern __shared__ __attribute__((aligned(16))) float dynt[];      // [red: TNW x 1024 floats][mask: 32 x Hs bytes]
    float *red = dynt;
    unsigned char *mk = reinterpret_cast<unsigned char *>(dynt + TNW * 1024);
    __shared__ float tile[32][LDT];       // logits, then log-probs
    __shared__ float dl[32][LDT];         // gradient at the logits (0 outside [T) x [V))
    __shared__ float al[32][65];          // alpha, then alpha + beta
    __shared__ float be[32][65];
    __shared__ int lab_s[64];
    __shared__ unsigned cmask[32];        // per class: the target positions that carry it
    __shared__ float s_nll;
    __shared__ int s_fail;
    const int n = blockIdx.x, y = blockIdx.y, SL = gridDim.y, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = p.T, H = p.H, V = p.V, S_ = 2 * p.S + 1;
    const int Hs = H / SL, h0 = y * Hs;
    const int r = lane & 31, kh = lane >> 5;
    DropoutCfg drop = p.drop;             // the step counter of a graph replay: read once, not once per draw
    if (drop.offset_dev) { drop.offset += *drop.offset_dev; drop.offset_dev = nullptr; }
    const bool masked = drop.threshold != 0;
    HST(0, 0);
    if (tid == 0) s_fail = 0;
    const unsigned epoch = p.ticket[1] + 1u;      // launches so far + 1: what this launch's partial sums are tagged with
    // ---- requested first, used last: the global-memory operands of this wave's first two product tiles (tiles 0 .. ntl-1: d features,
    //      their k index runs over W's rows; ntl .. 2 ntl-1: d W, k runs over the features' rows) ----
    const int ntl = Hs / 32, njobs = 2 * ntl;
    float pre0[16], pre1[16];
    // (buffer accesses through descriptors that end with the operand's last row: rows beyond it -- frames >= T, classes >= V -- lie out of
    //  range, their loads return 0 and their stores are dropped without a branch, a compare or a select)
    const int lane_k = (8 * kh * H + h0 + r) * 4;                // row 8 kh, column h0 + r
    auto load_job = [&](int jb, float (&o)[16]) {
        const bool isw = jb < ntl;
        const int col = (isw ? jb : jb - ntl) * 128, lim = isw ? V : T;
        const __amdgpu_buffer_rsrc_t rs =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(isw ? p.w : p.feats + (long)n * T * H), 0, lim * H * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int krow = 16 * (i >> 3) + (i & 7);                // + 8 kh: this lane's k index; a row >= lim lies beyond the descriptor: 0
            o[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, lane_k + krow * H * 4 + col, 0, 0));
        }
    };
    load_job(min(wave, njobs - 1), pre0);
    load_job(min(wave + TNW, njobs - 1), pre1);
    // ---- partial logits over this slice's K range: wave w takes k-steps (16 deep) w, w + 8, ..., two per pass; lane (r, kh) loads
    //      elements 8 kh .. 8 kh + 7 of row r of both operands (the MFMA's own layout) and draws the row's mask under the loads ----
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const int nks = Hs / 16;
    const long frow = ((long)n * T + min(r, T - 1)) * H + h0, wrow = (long)min(r, V - 1) * H + h0;
    for (int i0 = wave; i0 < nks; i0 += 2 * TNW) {
        f32x4 fa[2][2], wb[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int kb = 16 * min(i0 + TNW * u, nks - 1) + 8 * kh;
            fa[u][0] = *reinterpret_cast<const f32x4 *>(p.feats + frow + kb);
            fa[u][1] = *reinterpret_cast<const f32x4 *>(p.feats + frow + kb + 4);
            wb[u][0] = *reinterpret_cast<const f32x4 *>(p.w + wrow + kb);
            wb[u][1] = *reinterpret_cast<const f32x4 *>(p.w + wrow + kb + 4);
        }
        f32x4 mm[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            mm[u][0] = f32x4{1.f, 1.f, 1.f, 1.f}; mm[u][1] = mm[u][0];
            const int kb = 16 * min(i0 + TNW * u, nks - 1) + 8 * kh;
            if (masked && i0 + TNW * u < nks && r < T) {
                const uint64_t idx = (uint64_t)(((long)n * T + r) * H + h0 + kb);
                mm[u][0] = dropout_mult4(drop, idx);
                mm[u][1] = dropout_mult4(drop, idx + 4);
                unsigned lo4 = 0, hi4 = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) { lo4 |= (mm[u][0][e] != 0.f ? 1u : 0u) << (8 * e); hi4 |= (mm[u][1][e] != 0.f ? 1u : 0u) << (8 * e); }
                *reinterpret_cast<uint2 *>(mk + r * Hs + kb) = make_uint2(lo4, hi4);
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const bool on = i0 + TNW * u < nks;
            float a8[8], b8[8];
            const bool arow = on && r < T, brow = on && r < V;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a8[e] = arow ? fa[u][0][e] * mm[u][0][e] : 0.f; a8[4 + e] = arow ? fa[u][1][e] * mm[u][1][e] : 0.f;
                b8[e] = brow ? wb[u][0][e] : 0.f;               b8[4 + e] = brow ? wb[u][1][e] : 0.f;
            }
            bf16x8 ahi, alo, bhi, blo;
            split8(a8, ahi, alo);
            split8(b8, bhi, blo);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, blo, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo, bhi, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, bhi, acc, 0, 0, 0);
        }
    }
    HST(0, 1);
    const int64_t *tg = p.targets + (long)n * p.tg_stride;
    const int il = feature_length(p.il[n], p.ks, p.stride, p.pad, T);
    const int tl = max(0, min((int)p.tl[n], p.S));
    const int states = 2 * tl + 1;
    if (tid < 64) lab_s[tid] = tid < S_ ? ext_label2(tg, tid) : 0;
#pragma unroll
    for (int e = 0; e < 16; ++e) red[wave * 1024 + ((e & 3) + 8 * (e >> 2) + 4 * kh) * 32 + r] = acc[e];
    __syncthreads();
    float sum[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        sum[q] = 0.f;
#pragma unroll
        for (int w = 0; w < TNW; ++w) sum[q] += red[w * 1024 + tid + TNT * q];
    }
    HST(0, 2);
    if (SL > 1) {
        // the slices' partial sums meet in global memory.  Every value travels as the 8-byte pair (value, epoch of this launch): one
        // write-through store per pair, and a reader that finds the epoch has the value -- no drain, no counter, no second round trip
        // (cdna_hip_programming.md Guideline 16: data and flag in one naturally aligned store).  Everyone adds the SL values in slice order.
        // Only the frames below T travel; a thread whose element lies in a padding row polls row 0's element of its column instead (the
        // same few lines as wave 0's lanes) and stores nothing.
        const __amdgpu_buffer_rsrc_t prs = make_rsrc(p.ticket + 2);
        int slot[2];
        bool need[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            need[q] = (tid >> 5) + 16 * q < T;
            slot[q] = (n * SL * 1024 + (need[q] ? tid + TNT * q : (tid & 31))) * 8;
            u32x2 pr = {__builtin_bit_cast(unsigned, sum[q]), epoch};
            __builtin_amdgcn_raw_buffer_store_b64(pr, prs, (need[q] && n * SL + y != p.mute) ? slot[q] + y * 8192 : -16, 0, 16);      // (-16: out of range, dropped)
        }
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        auto gather = [&](auto ns_c) {
            constexpr int NS = decltype(ns_c)::value;
            u32x2 got[NS][2];
            bool ok = false;
            while (!ok) {
#pragma unroll
                for (int yy = 0; yy < NS; ++yy)
#pragma unroll
                    for (int q = 0; q < 2; ++q) got[yy][q] = __builtin_amdgcn_raw_buffer_load_b64(prs, slot[q] + min(yy, SL - 1) * 8192, 0, 16);
                ok = true;
#pragma unroll
                for (int yy = 0; yy < NS; ++yy) ok = ok && got[yy][0][1] == epoch && got[yy][1][1] == epoch;
                if (!ok && __builtin_amdgcn_s_memrealtime() - t0 > SPIN_TIMEOUT_TICKS) { s_fail = 1; break; }
            }
            sum[0] = sum[1] = 0.f;
#pragma unroll
            for (int yy = 0; yy < NS; ++yy)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const unsigned bits = got[yy][q][0];
                    sum[q] += yy < SL ? __builtin_bit_cast(float, bits) : 0.f;
                }
        };
        if (SL <= 2) gather(std::integral_constant<int, 2>{});
        else if (SL <= 4) gather(std::integral_constant<int, 4>{});
        else gather(std::integral_constant<int, 8>{});
    }
    HST(0, 3);
    const int tj = tid & 31;
#pragma unroll
    for (int q = 0; q < 2; ++q) tile[(tid >> 5) + 16 * q][tj] = (tj < V) ? sum[q] + p.bias[tj] : -INFINITY;
    __syncthreads();
    // A slice that never came (0.2 s: the utterance's workgroups were not all resident -- a CU-masked or partitioned device, a foreign kernel
    // holding CUs): loud, and the launch still ENDS in a clean state.  The workgroup raises the caller's sticky status word (the clip
    // launch then applies no update and LstmCtcTrainer.check_status() raises), poisons the utterance's loss term and skips the lattice
    // and the products (its d features / d W slots keep stale values nobody applies), but it still takes its loss ticket below: the
    // ticket reaches B, the last finisher resets it and advances the launch count, and the next launch's epoch matches none of the
    // pairs this one left behind.
    const bool failed = s_fail != 0;
    if (failed && tid == 0) {
        if (p.status) __hip_atomic_store(p.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_nll = NAN;
    }
    if (!failed) {
    // ---- log-softmax over the classes: 32 lanes per frame, two frames per thread ----
    {
        float lpv[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float x = tile[(tid >> 5) + 16 * q][tj];
            float m = x;
#pragma unroll
            for (int d = 16; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 32));
            float s = tj < V ? expf(x - m) : 0.f;
#pragma unroll
            for (int d = 16; d >= 1; d >>= 1) s += __shfl_xor(s, d, 32);
            lpv[q] = x - m - logf(s);
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int ti = (tid >> 5) + 16 * q;
            tile[ti][tj] = lpv[q];
            dl[ti][tj] = 0.f;
            if (p.lp && y == 0 && ti < T && tj < V) p.lp[((long)n * T + ti) * V + tj] = lpv[q];
        }
    }
    __syncthreads();
    HST(0, 4);
    // ---- the two lattice recursions side by side: alpha in wave 0, beta in wave 1, one lattice state per lane (the recursions of
    //      ctc_head_fwd_kernel / ctc_head_bwd_kernel).  A step is a dependent chain, so it is kept short: values in log2 units (v_exp_f32 /
    //      v_log_f32 are base-2: no scaling inside the chain), the lane's masks folded into -inf operands instead of branches (the clamped
    //      maximum keeps -inf - -inf out: every term then underflows to 0 and log2(0) = -inf), only the frames below the feature length ----
    constexpr float L2E = 1.4426950408889634f, LN2 = 0.6931471805599453f, NEG = -3.0e38f;
    auto lse3 = [](float x0, float x1, float x2) {
        const float m = fmaxf(fmaxf(fmaxf(x0, x1), x2), NEG);
        return m + __builtin_amdgcn_logf((__builtin_amdgcn_exp2f(x0 - m) + __builtin_amdgcn_exp2f(x1 - m)) + __builtin_amdgcn_exp2f(x2 - m));
    };
    if (wave == 0) {
        const int lab = lab_s[lane];
        const int lab2 = __shfl_up(lab, 2, 64);
        const bool skip = lane >= 2 && lab != 0 && lab != lab2, live = lane < states;
        const int slast = 2 * tl, sprev = tl > 0 ? 2 * tl - 1 : slast;
        float prev = (il > 0 && live && lane < 2) ? tile[0][lab] * L2E : -INFINITY;
        al[0][lane] = prev;
        float em = tile[min(1, T - 1)][lab] * L2E;
        for (int t = 1; t < il; ++t) {
            float p1 = wave_up1(prev), p2 = wave_up1(p1);
            p1 = lane >= 1 ? p1 : -INFINITY;
            p2 = skip ? p2 : -INFINITY;
            const float e = em;
            em = tile[min(t + 1, T - 1)][lab] * L2E;
            const float v = live ? lse3(prev, p1, p2) + e : -INFINITY;
            prev = v;
            al[t][lane] = v;
        }
        if (lane == 0) {
            float out = tl == 0 ? 0.f : INFINITY;                 // no frames: only the empty target is reachable
            if (il > 0) {
                const float ra = al[il - 1][slast], rb = tl > 0 ? al[il - 1][sprev] : -INFINITY;
                out = -lse3(ra, rb, -INFINITY) * LN2;
            }
            s_nll = out;
        }
    } else if (wave == 1 && il > 0) {
        const int lab = lab_s[lane];
        const int labn2 = __shfl_down(lab, 2, 64);
        const bool skip = lane + 2 < states && labn2 != 0 && labn2 != lab, live = lane < states, has1 = lane + 1 < states;
        float nxt = (lane == states - 1 || lane == states - 2) ? tile[il - 1][lab] * L2E : -INFINITY;
        be[il - 1][lane] = nxt;
        float em = tile[max(il - 2, 0)][lab] * L2E;
        for (int t = il - 2; t >= 0; --t) {
            float n1 = wave_down1(nxt), n2 = wave_down1(n1);
            n1 = has1 ? n1 : -INFINITY;
            n2 = skip ? n2 : -INFINITY;
            const float e = em;
            em = tile[max(t - 1, 0)][lab] * L2E;
            const float v = live ? lse3(nxt, n1, n2) + e : -INFINITY;
            be[t][lane] = v;
            nxt = v;
        }
    } else if (wave == 2 && lane < 32) {
        unsigned bits = 0;
        for (int j = 0; j < tl; ++j) bits |= (lab_s[2 * j + 1] == lane ? 1u : 0u) << j;
        cmask[lane] = bits;
    }
    __syncthreads();
    HST(0, 5);
    // ---- gradient at the log-probs (ATen's convention), then log_softmax backward: dlogit = g - exp(lp) * sum_c g.  Per frame t:
    //      e[s] = exp(alpha + beta - their maximum over the states) with the lane = the state (wave w takes frames w, w + 8, ...), the blank's
    //      share summed in the wave; then thread (t, c) adds the e[s] of the states that carry class c, found through a bit mask over the
    //      target positions -- no loop over all states, no label fetched twice ----
    for (int t = wave; t < il; t += TNW) {                       // (alpha, beta: log2 units)
        const float x = lane < states ? al[t][lane] + be[t][lane] : -INFINITY;
        const float m = wave_max(x);
        const float e = x > -INFINITY ? __builtin_amdgcn_exp2f(x - m) : 0.f;
        al[t][lane] = e;
        const float blank = wave_sum((lane & 1) ? 0.f : e);
        if (lane == 0) { be[t][64] = m; al[t][64] = blank; }
    }
    __syncthreads();
    {
        const float nll = s_nll, go = 1.0f / (fmaxf((float)p.tl[n], 1.0f) * (float)p.B);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int t = (tid >> 5) + 16 * q, c = tj;
            float g = 0.f;
            if (t < il && c < V) {
                float sm = al[t][64];
                if (c != 0) {
                    sm = 0.f;
                    for (unsigned bits = cmask[c]; bits; bits &= bits - 1u) sm += al[t][2 * (__builtin_ctz(bits)) + 1];
                }
                const float lcab = sm > 0.f ? (be[t][64] + __builtin_amdgcn_logf(sm)) * LN2 : -INFINITY;
                const float l = tile[t][c];
                g = (__expf(l) - __expf(lcab + nll - l)) * go;
            }
            float gs = g;
#pragma unroll
            for (int d = 16; d >= 1; d >>= 1) gs += __shfl_xor(gs, d, 32);
            if (t < T && c < V) dl[t][c] = g - expf(tile[t][c]) * gs;
        }
    }
    __syncthreads();
    HST(0, 6);
    // ---- the two products of this slice's columns, one 32-column tile per wave and pass (split-bf16, K = 32 classes / 32 frames) ----
    const __amdgpu_buffer_rsrc_t df_rs = __builtin_amdgcn_make_buffer_rsrc(p.dfeats + (long)n * T * H, 0, T * H * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t dw_rs = __builtin_amdgcn_make_buffer_rsrc(p.dw_part + (long)n * V * H, 0, V * H * 4, 0x00020000);
    const int lane_o = (4 * kh * H + h0 + r) * 4;                // accumulator row 4 kh, column h0 + r
    auto do_job = [&](auto isw_c, int ti, const float (&pre)[16]) {
        constexpr bool isw = decltype(isw_c)::value;
        const int m0 = ti * 32;
        // the mask bytes this tile needs, all requested at once (rows clamped: the unused ones are never looked at): d W multiplies the
        // DROPPED features (rows = its k index), d features is masked on the way out (rows = the accumulator's rows)
        unsigned mb[16];
        if (masked) {
            const unsigned char *mcol = mk + m0 + r;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = isw ? (i & 3) + 8 * (i >> 2) + 4 * kh : 16 * (i >> 3) + 8 * kh + (i & 7);
                mb[i] = mcol[min(row, T - 1) * Hs];
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) mb[i] = 1u;
        }
        f32x16 o;
#pragma unroll
        for (int e = 0; e < 16; ++e) o[e] = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            float a8[8], b8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int kk = 16 * s2 + 8 * kh + e;
                const float bv = pre[8 * s2 + e];
                b8[e] = isw ? bv : (mb[8 * s2 + e] ? bv * drop.scale : 0.f);
                a8[e] = isw ? dl[r][kk] : dl[kk][r];
            }
            bf16x8 ahi, alo, bhi, blo;
            split8(a8, ahi, alo);
            split8(b8, bhi, blo);
            o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, blo, o, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo, bhi, o, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, bhi, o, 0, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int rowc = (e & 3) + 8 * (e >> 2);                 // + 4 kh: the accumulator element's row; beyond T / V: beyond the descriptor
            const int off = lane_o + rowc * H * 4 + ti * 128;
            const float ov = isw ? (mb[e] ? o[e] * drop.scale : 0.f) : o[e];      // (a scalar first: bit_cast of a vector ELEMENT reads element 0)
            if (isw) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, ov), df_rs, off, 0, 0);
            else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, ov), dw_rs, off, 0, 0);
        }
    };
    auto run_job = [&](int jb, const float (&pre)[16]) {
        if (jb < ntl) do_job(std::true_type{}, jb, pre);
        else do_job(std::false_type{}, jb - ntl, pre);
    };
    if (wave < njobs) run_job(wave, pre0);
    if (wave + TNW < njobs) run_job(wave + TNW, pre1);
    for (int jb = wave + 2 * TNW; jb < njobs; jb += TNW) {       // (more tiles than two per wave: batches above 64 rows; a rotation that
        load_job(jb, pre0);                                      //  requests two tiles ahead measured slower at every batch size)
        run_job(jb, pre0);
    }
    if (wave == TNW - 1 && lane < V && y == 0) {
        float s = 0.f;
        for (int t = 0; t < T; ++t) s += dl[t][lane];
        p.db_part[(long)n * V + lane] = s;
    }
    HST(0, 7);
    }   // (!failed)
    // ---- the utterance's loss terms and, in the workgroup that finishes last, the mean (as ctc_head_fwd_kernel) ----
    if (y == 0 && wave == 0) {
        unsigned old = 0;
        if (lane == 0) {
            __hip_atomic_store(p.nll + n, s_nll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            p.flen[n] = il;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            old = __hip_atomic_fetch_add(p.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        old = __builtin_amdgcn_readfirstlane(old);
        if (old == (unsigned)p.B - 1u) {
            float s = 0.f;
            for (int i0 = 0; i0 < p.B; i0 += 64) {
                const int i = i0 + lane;
                float v = 0.f;
                if (i < p.B) v = __hip_atomic_load(p.nll + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) / fmaxf((float)p.tl[i], 1.0f);
                s += wave_sum(v);
            }
            if (lane == 0) {
                *p.loss = s / (float)p.B;
                __hip_atomic_store(p.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(p.ticket + 1, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// TemporalClassifier.decode's launch (halo_ctc_head_greedy) with the classifier product on split-bf16 MFMA, fragments straight from global
// memory as in ctc_head_train_kernel (every mode but exact f32, which keeps ctc_head_fwd_kernel<true>): 512 exact-f32 MFMAs behind four
// staged chunks were 12 of that launch's 17 us.  One workgroup per utterance, eight waves over the 16-deep k-steps, two per pass.
__global__ __launch_bounds__(TNT) void ctc_head_greedy2_kernel(const HeadFwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) float redg[];      // [TNW][1024]
    __shared__ float tile[32][LDT];
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = p.T, H = p.H, V = p.V;
    const int r = lane & 31, kh = lane >> 5;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const int nks = H / 16;
    const long frow = ((long)n * T + min(r, T - 1)) * H, wrow = (long)min(r, V - 1) * H;
    for (int i0 = wave; i0 < nks; i0 += 2 * TNW) {
        f32x4 fa[2][2], wb[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int kb = 16 * min(i0 + TNW * u, nks - 1) + 8 * kh;
            fa[u][0] = *reinterpret_cast<const f32x4 *>(p.feats + frow + kb);
            fa[u][1] = *reinterpret_cast<const f32x4 *>(p.feats + frow + kb + 4);
            wb[u][0] = *reinterpret_cast<const f32x4 *>(p.w + wrow + kb);
            wb[u][1] = *reinterpret_cast<const f32x4 *>(p.w + wrow + kb + 4);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const bool on = i0 + TNW * u < nks;
            const bool arow = on && r < T, brow = on && r < V;
            float a8[8], b8[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a8[e] = arow ? fa[u][0][e] : 0.f; a8[4 + e] = arow ? fa[u][1][e] : 0.f;
                b8[e] = brow ? wb[u][0][e] : 0.f; b8[4 + e] = brow ? wb[u][1][e] : 0.f;
            }
            bf16x8 ahi, alo, bhi, blo;
            split8(a8, ahi, alo);
            split8(b8, bhi, blo);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, blo, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo, bhi, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, bhi, acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) redg[wave * 1024 + ((e & 3) + 8 * (e >> 2) + 4 * kh) * 32 + r] = acc[e];
    __syncthreads();
    const int tj = tid & 31;
    float lpv[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < TNW; ++w) sum += redg[w * 1024 + tid + TNT * q];
        const float x = (tj < V) ? sum + p.bias[tj] : -INFINITY;
        float m = x;
#pragma unroll
        for (int d = 16; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 32));
        float sm = tj < V ? expf(x - m) : 0.f;
#pragma unroll
        for (int d = 16; d >= 1; d >>= 1) sm += __shfl_xor(sm, d, 32);
        lpv[q] = x - m - logf(sm);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int ti = (tid >> 5) + 16 * q;
        tile[ti][tj] = lpv[q];
        if (p.lp && ti < T && tj < V) p.lp[((long)n * T + ti) * V + tj] = lpv[q];
    }
    __syncthreads();
    // one lane per frame (T <= 32): arg max over the classes (first maximum, like torch.max), then unique_consecutive + drop blanks
    if (wave == 0) {
        const int t = lane;
        int best = -1;
        float bv = -INFINITY;
        if (t < T) {
            best = 0; bv = tile[t][0];
            for (int c = 1; c < V; ++c) {
                const float v = tile[t][c];
                if (v > bv) { bv = v; best = c; }
            }
            p.ali[(long)n * T + t] = best;
            p.scores[(long)n * T + t] = bv;
        }
        int left = __shfl_up(best, 1, 64);
        if (lane == 0) left = -1;
        const bool keep = (t < T) && best != left && best != 0;
        const unsigned long long mask = __ballot(keep);
        const int pos = __popcll(mask & ((1ull << lane) - 1ull)), count = __popcll(mask);
        if (keep) p.hyp[(long)n * T + pos] = best;
        if (t < T && t >= count) p.hyp[(long)n * T + t] = 0;
        if (lane == 0) p.hyp_len[n] = count;
    }
}

// dW[v][k] = sum_n dw_part[n][v][k], db[v] = sum_n db_part[n][v]: small_jobs.h kind 1 (a block takes 64 consecutive elements, its four
// waves a quarter of the utterances each); this launch when the caller does not defer small reductions
__global__ __launch_bounds__(256) void ctc_head_reduce_kernel(const HaloSmallJobs q) {
    __shared__ float part[4][64];
    halo_small_jobs_block(q, blockIdx.x, part);
}

}  // namespace

extern "C" {

int halo_ctc_head_supported(int T, int H, int V, int S) { return T > 0 && T <= 32 && V > 0 && V <= 32 && S >= 0 && 2 * S + 1 <= 64 && H > 0 && H % 64 == 0; }

size_t halo_ctc_head_workspace_bytes(int B, int H, int V) {
    if (B <= 0 || H <= 0 || V <= 0) return 0;
    return ((size_t)B * V * H + (size_t)B * V) * sizeof(float);
}

int halo_ctc_head_fwd(const float *features, const float *weight, const float *bias, float p_drop, uint64_t seed, uint32_t stream_id,
                      uint32_t offset, const uint32_t *offset_dev, const int64_t *input_lengths, int ks, int stride, int pad,
                      const int64_t *targets, long tg_stride, int S, const int64_t *target_lengths, float *lp, float *alpha, float *nll,
                      int64_t *feature_lengths, float *grad_out, float *loss, uint32_t *ticket, int B, int T, int H, int V,
                      halo_stream_t stream) {
    HALO_CHECK_ARG(features && weight && bias && input_lengths && targets && target_lengths && lp && alpha && nll && feature_lengths &&
                   grad_out && loss && ticket && B > 0);
    if (!halo_ctc_head_supported(T, H, V, S)) return HALO_ENOTSUP;
    HALO_CHECK_ARG(((uintptr_t)features % 16 == 0) && ((uintptr_t)weight % 16 == 0));
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void *)ctc_head_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)HEAD_FWD_LDS) != hipSuccess)
            return HALO_ELAUNCH;
        attr = true;
    }
    HeadFwdArgs a;
    a.feats = features; a.w = weight; a.bias = bias;
    a.drop = make_dropout(p_drop, seed, stream_id, offset, offset_dev);
    a.il = input_lengths; a.targets = targets; a.tl = target_lengths; a.tg_stride = tg_stride;
    a.lp = lp; a.alpha = alpha; a.nll = nll; a.grad_out = grad_out; a.loss = loss; a.flen = feature_lengths; a.ticket = ticket;
    a.B = B; a.T = T; a.H = H; a.V = V; a.S = S; a.ks = ks; a.stride = stride; a.pad = pad;
    a.ali = a.hyp = a.hyp_len = nullptr; a.scores = nullptr;
    hipLaunchKernelGGL(ctc_head_fwd_kernel<false>, dim3(B), dim3(1024), HEAD_FWD_LDS, (hipStream_t)stream, a);
    return halo_launch_status();
}

int halo_ctc_head_greedy(const float *features, const float *weight, const float *bias, float *lp, int64_t *alignments, float *scores,
                         int64_t *hyp, int64_t *hyp_len, int B, int T, int H, int V, halo_stream_t stream) {
    HALO_CHECK_ARG(features && weight && bias && alignments && scores && hyp && hyp_len && B > 0);
    if (!halo_ctc_head_supported(T, H, V, 0)) return HALO_ENOTSUP;
    HALO_CHECK_ARG(((uintptr_t)features % 16 == 0) && ((uintptr_t)weight % 16 == 0));
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void *)ctc_head_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)HEAD_FWD_LDS) != hipSuccess)
            return HALO_ELAUNCH;
        attr = true;
    }
    HeadFwdArgs a = {};
    a.feats = features; a.w = weight; a.bias = bias;
    a.drop = make_dropout(0.f, 0, 0, 0, nullptr);
    a.lp = lp; a.ali = alignments; a.scores = scores; a.hyp = hyp; a.hyp_len = hyp_len;
    a.B = B; a.T = T; a.H = H; a.V = V;
    if (halo_math_mode() != HALO_MATH_F32) {
        hipLaunchKernelGGL(ctc_head_greedy2_kernel, dim3(B), dim3(TNT), (size_t)TNW * 1024 * sizeof(float), (hipStream_t)stream, a);
        return halo_launch_status();
    }
    hipLaunchKernelGGL(ctc_head_fwd_kernel<true>, dim3(B), dim3(1024), HEAD_FWD_LDS, (hipStream_t)stream, a);
    return halo_launch_status();
}

int halo_ctc_head_bwd(const float *features, const float *weight, float p_drop, uint64_t seed, uint32_t stream_id, uint32_t offset,
                      const uint32_t *offset_dev, const int64_t *feature_lengths, const int64_t *targets, long tg_stride, int S,
                      const int64_t *target_lengths, const float *lp, const float *alpha, const float *nll, const float *grad_out,
                      float *dfeatures, float *dweight, float *dbias, void *workspace, int B, int T, int H, int V,
                      halo_stream_t stream) {
    HALO_CHECK_ARG(features && weight && feature_lengths && targets && target_lengths && lp && alpha && nll && grad_out && dfeatures &&
                   dweight && dbias && workspace && B > 0);
    if (!halo_ctc_head_supported(T, H, V, S)) return HALO_ENOTSUP;
    HeadBwdArgs a;
    a.feats = features; a.w = weight;
    a.drop = make_dropout(p_drop, seed, stream_id, offset, offset_dev);
    a.flen = feature_lengths; a.targets = targets; a.tl = target_lengths; a.tg_stride = tg_stride;
    a.lp = lp; a.alpha = alpha; a.nll = nll; a.grad_out = grad_out;
    a.dfeats = dfeatures; a.dw_part = (float *)workspace; a.db_part = a.dw_part + (size_t)B * V * H;
    a.B = B; a.T = T; a.H = H; a.V = V; a.S = S;
    // slices of H per utterance: as many as keep the grid within the chip's CUs (4 at B = 64), whole 32-column tiles per slice
    int slices = 1;
    while (slices < 8 && (H / 32) % (2 * slices) == 0 && (long)B * 2 * slices <= 256) slices *= 2;
    const size_t mask_bytes = p_drop > 0.f ? (size_t)T * (H / slices) * sizeof(float) : 0;       // <= 128 KiB at T = 32, H = 1024
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void *)ctc_head_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 32 * 1024 * 4) != hipSuccess)
            return HALO_ELAUNCH;
        attr = true;
    }
    if (mask_bytes > 32 * 1024 * 4) return HALO_ENOTSUP;
    hipLaunchKernelGGL(ctc_head_bwd_kernel, dim3(B, slices), dim3(512), mask_bytes, (hipStream_t)stream, a);
    int rc = halo_launch_status();
    if (rc != HALO_OK) return rc;
    const long VH = (long)V * H;
    // the fixed-order sum of the partials: queued for a later launch's tail blocks when the caller defers small reductions (small_jobs.h)
    HaloSmallJob j = {};
    j.kind = 1; j.n = B; j.m = V; j.len = VH; j.a = a.dw_part; j.b = a.db_part; j.o1 = dweight; j.o2 = dbias;
    if (halo_defer_small_job(j)) return HALO_OK;
    HaloSmallJobs q = {};
    q.n = 1; q.job[0] = j; q.job[0].blocks = q.blocks = halo_small_job_blocks(j);
    hipLaunchKernelGGL(ctc_head_reduce_kernel, dim3((unsigned)q.blocks), dim3(256), 0, (hipStream_t)stream, q);
    return halo_launch_status();
}

static int head_train_slices(int B, int H) {
    // slices of H per utterance: as many as keep the grid within the chip's CUs (4 at B = 64), whole 32-column tiles per slice.  All
    // B * slices workgroups must be resident at once (the slices of an utterance wait for each other): one per CU of THIS device (a
    // partitioned or CU-masked gfx950 reports fewer than 256); a batch that does not fit runs one slice per utterance, which waits for nobody
    const int cus = halo_cu_count() > 0 ? halo_cu_count() : 1;
    int slices = 1;
    while (slices < 8 && (H / 32) % (2 * slices) == 0 && (long)B * 2 * slices <= cus) slices *= 2;
    return slices;
}

size_t halo_ctc_head_train_workspace_bytes(int B, int H, int V) { return halo_ctc_head_workspace_bytes(B, H, V); }

size_t halo_ctc_head_train_ticket_words(int B, int H) {
    if (B <= 0 || H <= 0) return 0;
    const int slices = head_train_slices(B, H);
    return 2 + (slices > 1 ? (size_t)B * slices * 1024 * 2 : 0);       // loss ticket, launch count, the slices' (partial logit, epoch) pairs
}

int halo_ctc_head_train(const float *features, const float *weight, const float *bias, float p_drop, uint64_t seed, uint32_t stream_id,
                        uint32_t offset, const uint32_t *offset_dev, const int64_t *input_lengths, int ks, int stride, int pad,
                        const int64_t *targets, long tg_stride, int S, const int64_t *target_lengths, float *lp, float *nll,
                        int64_t *feature_lengths, float *loss, uint32_t *ticket, float *dfeatures, float *dweight, float *dbias,
                        void *workspace, int B, int T, int H, int V, halo_stream_t stream) {
    HALO_CHECK_ARG(features && weight && bias && input_lengths && targets && target_lengths && nll && feature_lengths && loss && ticket &&
                   dfeatures && dweight && dbias && workspace && B > 0);
    if (!halo_ctc_head_supported(T, H, V, S) || halo_math_mode() == HALO_MATH_F32) return HALO_ENOTSUP;
    if ((long)32 * H * 4 >= (1l << 31)) return HALO_ENOTSUP;
    HALO_CHECK_ARG(((uintptr_t)features % 16 == 0) && ((uintptr_t)weight % 16 == 0));
    const int slices = head_train_slices(B, H);
    const size_t lds = (size_t)TNW * 1024 * sizeof(float) + (p_drop > 0.f ? (size_t)32 * (H / slices) : 0);
    if (!halo_func_attr_done(0)) {         // (the attribute is per device: one flag per device and kernel, halo_internal.h)
        if (hipFuncSetAttribute((const void *)ctc_head_train_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(TNW * 1024 * sizeof(float) + 32 * 2048)) != hipSuccess)
            return HALO_ELAUNCH;
        halo_func_attr_set(0);
    }
    if (H / slices > 2048) return HALO_ENOTSUP;
    HeadTrainArgs a;
    a.feats = features; a.w = weight; a.bias = bias;
    a.drop = make_dropout(p_drop, seed, stream_id, offset, offset_dev);
    a.il = input_lengths; a.targets = targets; a.tl = target_lengths; a.tg_stride = tg_stride;
    a.lp = lp; a.nll = nll; a.loss = loss; a.flen = feature_lengths; a.ticket = ticket;
    a.status = halo_ctx_cur().status;
    a.mute = halo_ctx_cur().mute_block;
    a.dfeats = dfeatures; a.dw_part = (float *)workspace; a.db_part = a.dw_part + (size_t)B * V * H;
    a.B = B; a.T = T; a.H = H; a.V = V; a.S = S; a.ks = ks; a.stride = stride; a.pad = pad;
    hipLaunchKernelGGL(ctc_head_train_kernel, dim3(B, slices), dim3(TNT), lds, (hipStream_t)stream, a);
    int rc = halo_launch_status();
    if (rc != HALO_OK) return rc;
    HaloSmallJob j = {};
    j.kind = 1; j.n = B; j.m = V; j.len = (long)V * H; j.a = a.dw_part; j.b = a.db_part; j.o1 = dweight; j.o2 = dbias;
    if (halo_defer_small_job(j)) return HALO_OK;
    HaloSmallJobs q = {};
    q.n = 1; q.job[0] = j; q.job[0].blocks = q.blocks = halo_small_job_blocks(j);
    hipLaunchKernelGGL(ctc_head_reduce_kernel, dim3((unsigned)q.blocks), dim3(256), 0, (hipStream_t)stream, q);
    return halo_launch_status();
}

#ifdef HALO_HEAD_STAMPS
int halo_debug_head_stamps(void *dst) { return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_head_stamps), sizeof(g_head_stamps)) == hipSuccess ? HALO_OK : HALO_ELAUNCH; }
#endif

}  // extern "C"
