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

namespace {

constexpr int NW = 16;                   // waves per workgroup: K-slices of the H-deep contraction / column tiles of the H-wide outputs
constexpr int LDT = 33;                  // padded leading dimension of the [32][32] tiles in LDS

__device__ __forceinline__ float log_add_exp_fast2(float a, float b) {       // as ctc.hip's wave kernels
    if (isinf(a) && a == b) return a;
    const float m = fmaxf(a, b);
    return m + __logf(1.0f + __expf(-fabsf(a - b)));
}
__device__ __forceinline__ int ext_label2(const int64_t *tg, int s) { return (s & 1) ? (int)tg[s >> 1] : 0; }

struct HeadFwdArgs {
    const float *feats, *w, *bias;
    DropoutCfg drop;
    const int64_t *il, *targets, *tl;
    long tg_stride;
    float *lp, *alpha, *nll, *grad_out, *loss;
    int64_t *flen;
    unsigned *ticket;
    int B, T, H, V, S, ks, stride, pad;
};

__device__ __forceinline__ int feature_length(long il, int ks, int stride, int pad, int T) {
    const float o = (float)(il + 2 * pad - ks);
    const int f = (int)floorf(o / (float)stride + 1.0f);          // float arithmetic like ha/rnn.py:13-18
    return max(0, min(f, T));
}

__global__ __launch_bounds__(1024) void ctc_head_fwd_kernel(const HeadFwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) float red[];   // [NW][32*32] partial tiles
    __shared__ float tile[32][LDT];                               // logits, then log-probs
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = p.T, H = p.H, V = p.V;
    const int r = lane & 31, kh = lane >> 5;
    // ---- logits[t][v] = sum_k drop(f[n,t,k]) W[v,k]: this wave's K-slice ----
    const int ksl = H / NW, k0 = wave * ksl;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const float *frow = p.feats + ((long)n * T + r) * H + k0;
    const float *wrow = p.w + (long)r * H + k0;
    const bool arow = r < T, brow = r < V;
    for (int q = 0; q < ksl / 4; ++q) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
        if (arow) {
            a = *reinterpret_cast<const f32x4 *>(frow + 4 * q);
            if (p.drop.threshold) a = a * dropout_mult4(p.drop, (uint64_t)(((long)n * T + r) * H + k0 + 4 * q));
        }
        if (brow) b = *reinterpret_cast<const f32x4 *>(wrow + 4 * q);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(kh ? a[1] : a[0], kh ? b[1] : b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(kh ? a[3] : a[2], kh ? b[3] : b[2], acc, 0, 0, 0);
    }
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
        if (i < T && j < V) p.lp[((long)n * T + i) * V + j] = lpv;
    }
    __syncthreads();
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
        for (int t = 1; t < T; ++t) {
            const float p1 = __shfl_up(prev, 1, 64), p2 = __shfl_up(prev, 2, 64);
            float v = -INFINITY;
            if (t < il && lane < states) {
                if (lane == 0) v = prev + tile[t][0];
                else {
                    float a2 = log_add_exp_fast2(prev, p1);
                    if (lane >= 2 && can_skip) a2 = log_add_exp_fast2(a2, p2);
                    v = a2 + tile[t][lab];
                }
            }
            prev = v;
            if (lane < S_) alpha[(long)t * S_ + lane] = v;
            if (t == tlast) { ra = __shfl(prev, slast, 64); rb = sprev >= 0 ? __shfl(prev, sprev, 64) : -INFINITY; }
        }
        if (lane == 0) {
            const float out = il == 0 ? (tl == 0 ? 0.f : INFINITY) : -log_add_exp_fast2(ra, rb);
            // nll travels to the workgroup that finishes last: write-through store, drained, then the ticket (Guideline 16 R1)
            __hip_atomic_store(p.nll + n, out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            p.flen[n] = il;
            p.grad_out[n] = 1.0f / (fmaxf((float)p.tl[n], 1.0f) * (float)p.B);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned old = __hip_atomic_fetch_add(p.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old == (unsigned)p.B - 1u) {
                // reduction='mean' (ha/recognizer.py:71): mean_n(nll[n] / max(tl[n], 1)), summed in index order
                float s = 0.f;
                for (int i = 0; i < p.B; ++i)
                    s += __hip_atomic_load(p.nll + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) / fmaxf((float)p.tl[i], 1.0f);
                *p.loss = s / (float)p.B;
                __hip_atomic_store(p.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next call
            }
        }
    }
}

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

__global__ __launch_bounds__(1024) void ctc_head_bwd_kernel(const HeadBwdArgs p) {
    __shared__ float lps[32][LDT];        // log-probs
    __shared__ float dl[32][LDT];         // gradient at the logits (0 outside [T) x [V))
    __shared__ float ab[32][65];          // alpha, then alpha + beta
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = p.T, H = p.H, V = p.V, S_ = 2 * p.S + 1;
    const int64_t *tg = p.targets + (long)n * p.tg_stride;
    const int il = max(0, min((int)p.flen[n], T));
    const int tl = max(0, min((int)p.tl[n], p.S));
    const int states = 2 * tl + 1;
    {
        const int i = tid >> 5, j = tid & 31;
        lps[i][j] = (i < T && j < V) ? p.lp[((long)n * T + i) * V + j] : -INFINITY;
        dl[i][j] = 0.f;
    }
    for (int idx = tid; idx < T * S_; idx += 1024) ab[idx / S_][idx % S_] = p.alpha[(long)n * T * S_ + idx];
    __syncthreads();
    const float nll = p.nll[n], go = p.grad_out[n];
    if (wave == 0 && il > 0) {            // beta recursion, one state per lane (as ctc_beta_grad_wave_kernel)
        const int lab = lane < S_ ? ext_label2(tg, lane) : 0;
        const int labn2 = __shfl_down(lab, 2, 64);
        const bool can_skip = lane + 2 < states && labn2 != 0 && labn2 != lab;
        float nxt = -INFINITY;
        for (int t = il - 1; t >= 0; --t) {
            const float n1 = __shfl_down(nxt, 1, 64), n2 = __shfl_down(nxt, 2, 64);
            float v = -INFINITY;
            if (lane < states) {
                if (t == il - 1) {
                    if (lane == states - 1 || lane == states - 2) v = lps[t][lab];
                } else {
                    float a2 = log_add_exp_fast2(nxt, lane + 1 < states ? n1 : -INFINITY);
                    if (can_skip) a2 = log_add_exp_fast2(a2, n2);
                    v = a2 + lps[t][lab];
                }
                ab[t][lane] += v;
            }
            nxt = v;
        }
    }
    __syncthreads();
    {   // gradient at the log-probs in ATen's convention, then log_softmax backward: dlogit = g - exp(lp) * sum_c g
        const int t = tid >> 5, c = tid & 31;
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
    if (tid < V) {
        float s = 0.f;
        for (int t = 0; t < T; ++t) s += dl[t][tid];
        p.db_part[(long)n * V + tid] = s;
    }
    // ---- the two products, column tiles of 32 over H: wave w takes tiles w, w + NW, ... ----
    const int r = lane & 31, kh = lane >> 5;
    for (int nt = wave; nt < H / 32; nt += NW) {
        const int c0 = nt * 32;
        f32x16 dx, dwp;
#pragma unroll
        for (int e = 0; e < 16; ++e) { dx[e] = 0.f; dwp[e] = 0.f; }
#pragma unroll 4
        for (int s = 0; s < 16; ++s) {
            const int kk = 2 * s + kh;                      // class index (d features) / frame index (d W)
            // d features[t][c0 + j] += dl[t][kk] * W[kk][c0 + j]
            const float wv = kk < V ? p.w[(long)kk * H + c0 + r] : 0.f;
            dx = __builtin_amdgcn_mfma_f32_32x32x2f32(dl[r][kk], wv, dx, 0, 0, 0);
            // d W[v][c0 + j] += dl[kk][v] * fdrop[kk][c0 + j]
            float fv = 0.f;
            if (kk < T) {
                const long e = ((long)n * T + kk) * H + c0 + r;
                fv = p.feats[e];
                if (p.drop.threshold) fv *= dropout_mult(p.drop, (uint64_t)e);
            }
            dwp = __builtin_amdgcn_mfma_f32_32x32x2f32(dl[kk][r], fv, dwp, 0, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = (e & 3) + 8 * (e >> 2) + 4 * kh;
            if (row < T) {
                const long idx = ((long)n * T + row) * H + c0 + r;
                float v = dx[e];
                if (p.drop.threshold) v *= dropout_mult(p.drop, (uint64_t)idx);
                p.dfeats[idx] = v;
            }
            if (row < V) p.dw_part[((long)n * V + row) * H + c0 + r] = dwp[e];
        }
    }
}

// dW[v][k] = sum_n dw_part[n][v][k], db[v] = sum_n db_part[n][v], in index order
__global__ __launch_bounds__(256) void ctc_head_reduce_kernel(const float *__restrict__ dw_part, const float *__restrict__ db_part,
                                                              float *__restrict__ dw, float *__restrict__ db, int B, long VH, int V) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < VH) {
        float s = 0.f;
        for (int n = 0; n < B; ++n) s += dw_part[(long)n * VH + i];
        dw[i] = s;
    }
    if (i < V) {
        float s = 0.f;
        for (int n = 0; n < B; ++n) s += db_part[(long)n * V + i];
        db[i] = s;
    }
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
        if (hipFuncSetAttribute((const void *)ctc_head_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NW * 1024 * 4) != hipSuccess)
            return HALO_ELAUNCH;
        attr = true;
    }
    HeadFwdArgs a;
    a.feats = features; a.w = weight; a.bias = bias;
    a.drop = make_dropout(p_drop, seed, stream_id, offset, offset_dev);
    a.il = input_lengths; a.targets = targets; a.tl = target_lengths; a.tg_stride = tg_stride;
    a.lp = lp; a.alpha = alpha; a.nll = nll; a.grad_out = grad_out; a.loss = loss; a.flen = feature_lengths; a.ticket = ticket;
    a.B = B; a.T = T; a.H = H; a.V = V; a.S = S; a.ks = ks; a.stride = stride; a.pad = pad;
    hipLaunchKernelGGL(ctc_head_fwd_kernel, dim3(B), dim3(1024), NW * 1024 * 4, (hipStream_t)stream, a);
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
    hipLaunchKernelGGL(ctc_head_bwd_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, a);
    int rc = halo_launch_status();
    if (rc != HALO_OK) return rc;
    const long VH = (long)V * H;
    hipLaunchKernelGGL(ctc_head_reduce_kernel, dim3((unsigned)((VH + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a.dw_part, a.db_part,
                       dweight, dbias, B, VH, V);
    return halo_launch_status();
}

}  // extern "C"
