// CTC alpha/beta lattice and greedy decode for gfx950: one workgroup per utterance, threads own
// lattice states; the 2S+1-wide alpha row lives in LDS (double buffered), so for the config
// shapes (2S+1 <= 64) the whole recursion is a single wavefront with no cross-wave traffic.
#include <float.h>
#include "halo_common.h"

namespace {

struct CtcArgs {
    const float *lp;
    long stride_t, stride_n;
    int T, N, C;
    const int64_t *targets;
    long tg_stride;
    int S;
    const int64_t *il;
    const int64_t *tl;
    int flags;
    float *alpha;   // [N,T,2S+1]
    float *nll;     // [N]
};

__device__ __forceinline__ int ext_label(const int64_t *tg, int s) { return (s & 1) ? (int)tg[s >> 1] : 0; }

__global__ void ctc_alpha_kernel(const CtcArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int n = blockIdx.x;
    const int S_ = 2 * p.S + 1;
    float *rowA = smem, *rowB = smem + S_;
    const int64_t *tg = p.targets + (long)n * p.tg_stride;
    const bool full = p.flags & HALO_CTC_FULL_LATTICE;
    const float neg = (p.flags & HALO_CTC_FINITE_MIN) ? -FLT_MAX : -INFINITY;
    int il = p.il ? (int)p.il[n] : p.T;
    int tl = (int)p.tl[n];
    il = max(0, min(il, p.T));
    tl = max(0, min(tl, p.S));
    const int states = full ? S_ : 2 * tl + 1;
    const int frames = full ? p.T : il;
    const float *lp = p.lp + (long)n * p.stride_n;
    float *alpha = p.alpha + (long)n * p.T * S_;

    // t = 0
    for (int s = threadIdx.x; s < S_; s += blockDim.x) {
        float v = neg;
        if (frames > 0 && s < states && s < 2) v = lp[ext_label(tg, s)];
        rowA[s] = v;
        if (p.T > 0) alpha[s] = v;
    }
    __syncthreads();
    float *prev = rowA, *cur = rowB;
    for (int t = 1; t < p.T; ++t) {
        const float *lpt = lp + (long)t * p.stride_t;
        for (int s = threadIdx.x; s < S_; s += blockDim.x) {
            float v = neg;
            if (t < frames && s < states) {
                const int lab = ext_label(tg, s);
                if (s == 0) {
                    v = (p.flags & HALO_CTC_NO_LEAD_BLANK_LOOP) ? neg : prev[0] + lpt[0];
                } else {
                    float acc = log_add_exp(prev[s], prev[s - 1]);
                    if (s >= 2) {
                        if (lab != 0 && lab != ext_label(tg, s - 2)) acc = log_add_exp(acc, prev[s - 2]);
                    } else if (p.flags & HALO_CTC_WRAP_SKIP) {
                        acc = log_add_exp(acc, prev[states - 1]);
                    }
                    v = acc + lpt[lab];
                }
            }
            cur[s] = v;
            alpha[(long)t * S_ + s] = v;
        }
        __syncthreads();
        float *tmp = prev; prev = cur; cur = tmp;
    }
    if (threadIdx.x == 0) {
        float out;
        if (full) {
            const int tlast = max(0, min((p.il ? (int)p.il[n] : p.T) - 1, p.T - 1));
            const int slast = min(2 * (int)p.tl[n], S_ - 1);
            const int sprev = (slast - 1 + S_) % S_;
            out = -log_add_exp(alpha[(long)tlast * S_ + slast], alpha[(long)tlast * S_ + sprev]);
        } else if (il == 0) {
            out = tl == 0 ? 0.f : INFINITY;
        } else {
            const float a = alpha[(long)(il - 1) * S_ + 2 * tl];
            const float b = tl > 0 ? alpha[(long)(il - 1) * S_ + 2 * tl - 1] : -INFINITY;
            out = -log_add_exp(a, b);
        }
        p.nll[n] = out;
    }
}

struct CtcBwdArgs {
    const float *lp;
    long stride_t, stride_n;
    int T, N, C;
    const int64_t *targets;
    long tg_stride;
    int S;
    const int64_t *il;
    const int64_t *tl;
    const float *alpha;
    const float *nll;
    const float *grad_out;
    float *beta;    // [N,T,2S+1]
    float *grad;
    long gstride_t, gstride_n;
};

__global__ void ctc_beta_grad_kernel(const CtcBwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int n = blockIdx.x;
    const int S_ = 2 * p.S + 1;
    float *rowA = smem, *rowB = smem + S_ + 2;   // two spare slots so s+1, s+2 never leave the row
    float *ab = smem + 2 * (S_ + 2);             // alpha+beta of the current frame [S_]
    const int64_t *tg = p.targets + (long)n * p.tg_stride;
    int il = p.il ? (int)p.il[n] : p.T;
    int tl = (int)p.tl[n];
    il = max(0, min(il, p.T));
    tl = max(0, min(tl, p.S));
    const int states = 2 * tl + 1;
    const float *lp = p.lp + (long)n * p.stride_n;
    const float *alpha = p.alpha + (long)n * p.T * S_;
    float *beta = p.beta + (long)n * p.T * S_;
    float *grad = p.grad + (long)n * p.gstride_n;
    const float nll = p.nll[n];
    const float go = p.grad_out[n];
    const float ninf = -INFINITY;

    // frames at and beyond the utterance's length carry no gradient
    for (int t = il; t < p.T; ++t)
        for (int c = threadIdx.x; c < p.C; c += blockDim.x) grad[(long)t * p.gstride_t + c] = 0.f;
    if (il == 0) return;

    float *next = rowA, *cur = rowB;
    for (int t = il - 1; t >= 0; --t) {
        const float *lpt = lp + (long)t * p.stride_t;
        for (int s = threadIdx.x; s < S_ + 2; s += blockDim.x) {
            float v = ninf;
            if (s < states) {
                const int lab = ext_label(tg, s);
                if (t == il - 1) {
                    if (s == states - 1 || (s == states - 2)) v = lpt[lab];
                } else {
                    float acc = log_add_exp(next[s], next[s + 1]);
                    if (s + 2 < states) {
                        const int lab2 = ext_label(tg, s + 2);
                        if (lab2 != 0 && lab2 != lab) acc = log_add_exp(acc, next[s + 2]);
                    }
                    v = acc + lpt[lab];
                }
                beta[(long)t * S_ + s] = v;
                ab[s] = alpha[(long)t * S_ + s] + v;
            }
            cur[s] = v;
        }
        __syncthreads();
        for (int c = threadIdx.x; c < p.C; c += blockDim.x) {
            // log-sum over the states that emit class c
            float m = ninf;
            if (c == 0) { for (int s = 0; s < states; s += 2) m = fmaxf(m, ab[s]); }
            else        { for (int s = 1; s < states; s += 2) if ((int)tg[s >> 1] == c) m = fmaxf(m, ab[s]); }
            float lcab = ninf;
            if (m > ninf) {
                float sum = 0.f;
                if (c == 0) { for (int s = 0; s < states; s += 2) sum += expf(ab[s] - m); }
                else        { for (int s = 1; s < states; s += 2) if ((int)tg[s >> 1] == c) sum += expf(ab[s] - m); }
                lcab = m + logf(sum);
            }
            const float l = lpt[c];
            grad[(long)t * p.gstride_t + c] = (expf(l) - expf(lcab + nll - l)) * go;
        }
        __syncthreads();
        float *tmp = next; next = cur; cur = tmp;
    }
}

// ---- single-wavefront fast paths (2S+1 <= 64 and the utterance's log-probs fit in LDS) ----------------
// The utterance's [T,C] log-prob rows are staged into LDS once; the lattice row lives in registers,
// one state per lane, and the s-1 / s-2 (s+1 / s+2) neighbours come from wave shuffles, so the
// T-step recursion has no memory round trips in its dependency chain.

// log(e^a + e^b) on the hardware exp2/log2 units (v_exp_f32 / v_log_f32): ~1e-7 absolute error per call,
// an order below the lattice's own fp32 rounding over T steps; the serial chain is ~10x shorter than
// with the libm-accurate expf/log1pf, which is what bounds a one-wave-per-utterance recursion.
__device__ __forceinline__ float log_add_exp_fast(float a, float b) {
    if (isinf(a) && a == b) return a;
    const float m = fmaxf(a, b);
    return m + __logf(1.0f + __expf(-fabsf(a - b)));
}

__global__ __launch_bounds__(64) void ctc_alpha_wave_kernel(const CtcArgs p) {
    extern __shared__ __attribute__((aligned(16))) float lp_s[];   // [T*C]
    const int n = blockIdx.x, lane = threadIdx.x;
    const int S_ = 2 * p.S + 1, T = p.T, C = p.C;
    const int64_t *tg = p.targets + (long)n * p.tg_stride;
    const bool full = p.flags & HALO_CTC_FULL_LATTICE;
    const float neg = (p.flags & HALO_CTC_FINITE_MIN) ? -FLT_MAX : -INFINITY;
    const int il_raw = p.il ? (int)p.il[n] : T;
    const int il = max(0, min(il_raw, T));
    const int tl = max(0, min((int)p.tl[n], p.S));
    const int states = full ? S_ : 2 * tl + 1;
    const int frames = full ? T : il;
    const float *lp = p.lp + (long)n * p.stride_n;
    float *alpha = p.alpha + (long)n * T * S_;
    for (int idx = lane; idx < T * C; idx += 64) lp_s[idx] = lp[(long)(idx / C) * p.stride_t + (idx % C)];
    __syncthreads();

    const int lab = lane < S_ ? ext_label(tg, lane) : 0;
    const int lab2 = __shfl_up(lab, 2, 64);
    const bool can_skip = lane >= 2 && lab != 0 && lab != lab2;
    int tlast, slast, sprev;
    if (full) { tlast = max(0, min(il_raw - 1, T - 1)); slast = min(2 * (int)p.tl[n], S_ - 1); sprev = (slast - 1 + S_) % S_; }
    else      { tlast = il - 1; slast = 2 * tl; sprev = tl > 0 ? 2 * tl - 1 : -1; }

    float prev = (frames > 0 && lane < states && lane < 2) ? lp_s[lab] : neg;
    if (lane < S_) alpha[lane] = prev;
    float ra = 0.f, rb = 0.f;
    if (tlast == 0) { ra = __shfl(prev, slast, 64); rb = sprev >= 0 ? __shfl(prev, sprev, 64) : -INFINITY; }
    for (int t = 1; t < T; ++t) {
        const float p1 = __shfl_up(prev, 1, 64), p2 = __shfl_up(prev, 2, 64);
        const float plast = __shfl(prev, states - 1, 64);
        float v = neg;
        if (t < frames && lane < states) {
            if (lane == 0) {
                v = (p.flags & HALO_CTC_NO_LEAD_BLANK_LOOP) ? neg : prev + lp_s[t * C];
            } else {
                float acc = log_add_exp_fast(prev, p1);
                if (lane >= 2) { if (can_skip) acc = log_add_exp_fast(acc, p2); }
                else if (p.flags & HALO_CTC_WRAP_SKIP) acc = log_add_exp_fast(acc, plast);
                v = acc + lp_s[t * C + lab];
            }
        }
        prev = v;
        if (lane < S_) alpha[(long)t * S_ + lane] = v;
        if (t == tlast) { ra = __shfl(prev, slast, 64); rb = sprev >= 0 ? __shfl(prev, sprev, 64) : -INFINITY; }
    }
    if (lane == 0) {
        float out;
        if (!full && il == 0) out = tl == 0 ? 0.f : INFINITY;
        else out = -log_add_exp_fast(ra, rb);
        p.nll[n] = out;
    }
}

__global__ __launch_bounds__(256) void ctc_beta_grad_wave_kernel(const CtcBwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int S_ = 2 * p.S + 1, T = p.T, C = p.C;
    float *lp_s = smem;               // [T*C]
    float *ab = smem + T * C;         // [T*S_] alpha, then alpha + beta
    const int64_t *tg = p.targets + (long)n * p.tg_stride;
    const int il = max(0, min(p.il ? (int)p.il[n] : T, T));
    const int tl = max(0, min((int)p.tl[n], p.S));
    const int states = 2 * tl + 1;
    const float *lp = p.lp + (long)n * p.stride_n;
    const float *alpha = p.alpha + (long)n * T * S_;
    float *beta = p.beta + (long)n * T * S_;
    float *grad = p.grad + (long)n * p.gstride_n;
    const float nll = p.nll[n], go = p.grad_out[n];
    const float ninf = -INFINITY;
    for (int idx = tid; idx < T * C; idx += 256) lp_s[idx] = lp[(long)(idx / C) * p.stride_t + (idx % C)];
    for (int idx = tid; idx < T * S_; idx += 256) ab[idx] = alpha[idx];
    __syncthreads();
    if (tid < 64 && il > 0) {         // wave 0 runs the recursion, one state per lane
        const int lab = lane < S_ ? ext_label(tg, lane) : 0;
        const int labn2 = __shfl_down(lab, 2, 64);
        const bool can_skip = lane + 2 < states && labn2 != 0 && labn2 != lab;
        float nxt = ninf;
        for (int t = il - 1; t >= 0; --t) {
            const float n1 = __shfl_down(nxt, 1, 64), n2 = __shfl_down(nxt, 2, 64);
            float v = ninf;
            if (lane < states) {
                if (t == il - 1) {
                    if (lane == states - 1 || lane == states - 2) v = lp_s[t * C + lab];
                } else {
                    float acc = log_add_exp_fast(nxt, lane + 1 < states ? n1 : ninf);
                    if (can_skip) acc = log_add_exp_fast(acc, n2);
                    v = acc + lp_s[t * C + lab];
                }
                beta[(long)t * S_ + lane] = v;
                ab[t * S_ + lane] += v;
            }
            nxt = v;
        }
    }
    __syncthreads();
    for (int idx = tid; idx < T * C; idx += 256) {
        const int t = idx / C, c = idx % C;
        float g = 0.f;
        if (t < il) {
            const float *row = ab + t * S_;
            float m = ninf;
            if (c == 0) { for (int s = 0; s < states; s += 2) m = fmaxf(m, row[s]); }
            else        { for (int s = 1; s < states; s += 2) if ((int)tg[s >> 1] == c) m = fmaxf(m, row[s]); }
            float lcab = ninf;
            if (m > ninf) {
                float sum = 0.f;
                if (c == 0) { for (int s = 0; s < states; s += 2) sum += __expf(row[s] - m); }
                else        { for (int s = 1; s < states; s += 2) if ((int)tg[s >> 1] == c) sum += __expf(row[s] - m); }
                lcab = m + __logf(sum);
            }
            const float l = lp_s[idx];
            g = (__expf(l) - __expf(lcab + nll - l)) * go;
        }
        grad[(long)t * p.gstride_t + c] = g;
    }
}

// Greedy decode: one wave per utterance, a lane per frame, 64 frames per pass.
__global__ __launch_bounds__(64) void ctc_greedy_kernel(const float *__restrict__ lp, int N, int T, int C,
                                                        int64_t *__restrict__ ali, float *__restrict__ scores,
                                                        int64_t *__restrict__ hyp, int64_t *__restrict__ hyp_len) {
    const int n = blockIdx.x, lane = threadIdx.x;
    int count = 0;
    int carry = -1;   // symbol of the last frame of the previous pass
    for (int t0 = 0; t0 < T; t0 += 64) {
        const int t = t0 + lane;
        int best = -1;
        float bv = -INFINITY;
        if (t < T) {
            const float *row = lp + ((long)n * T + t) * C;
            best = 0; bv = row[0];
            for (int c = 1; c < C; ++c) {
                const float v = row[c];
                if (v > bv) { bv = v; best = c; }
            }
            ali[(long)n * T + t] = best;
            scores[(long)n * T + t] = bv;
        }
        int left = __shfl_up(best, 1, 64);
        if (lane == 0) left = carry;
        const bool keep = (t < T) && best != left && best != 0;
        const unsigned long long mask = __ballot(keep);
        const int pos = count + __popcll(mask & ((1ull << lane) - 1ull));
        if (keep) hyp[(long)n * T + pos] = best;
        count += __popcll(mask);
        carry = __shfl(best, 63, 64);
    }
    if (lane == 0) hyp_len[n] = count;
}

inline int block_for_states(int S_) {
    int b = ((S_ + 2 + 63) / 64) * 64;
    return b > 1024 ? 1024 : b;
}

}  // namespace

extern "C" {

int halo_ctc_fwd(const float *lp, long stride_t, long stride_n, int T, int N, int C, const int64_t *targets,
                 long tg_stride, int S, const int64_t *input_lengths, const int64_t *target_lengths, int flags,
                 float *alpha, float *nll, halo_stream_t stream) {
    HALO_CHECK_ARG(lp && targets && target_lengths && alpha && nll);
    HALO_CHECK_ARG(T > 0 && N > 0 && C > 0 && S >= 0 && tg_stride >= S);
    const int S_ = 2 * S + 1;
    const size_t shmem = (size_t)2 * S_ * sizeof(float);
    if (shmem > 64 * 1024) return HALO_ENOTSUP;
    CtcArgs a;
    a.lp = lp; a.stride_t = stride_t; a.stride_n = stride_n; a.T = T; a.N = N; a.C = C;
    a.targets = targets; a.tg_stride = tg_stride; a.S = S; a.il = input_lengths; a.tl = target_lengths;
    a.flags = flags; a.alpha = alpha; a.nll = nll;
    const size_t wave_shmem = (size_t)T * C * sizeof(float);
    if (S_ <= 64 && wave_shmem <= 60 * 1024)
        hipLaunchKernelGGL(ctc_alpha_wave_kernel, dim3(N), dim3(64), wave_shmem, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(ctc_alpha_kernel, dim3(N), dim3(block_for_states(S_)), shmem, (hipStream_t)stream, a);
    return halo_launch_status();
}

int halo_ctc_bwd(const float *lp, long stride_t, long stride_n, int T, int N, int C, const int64_t *targets,
                 long tg_stride, int S, const int64_t *input_lengths, const int64_t *target_lengths, const float *alpha,
                 const float *nll, const float *grad_out, float *beta, float *grad, long gstride_t, long gstride_n,
                 halo_stream_t stream) {
    HALO_CHECK_ARG(lp && targets && target_lengths && alpha && nll && grad_out && beta && grad);
    HALO_CHECK_ARG(T > 0 && N > 0 && C > 0 && S >= 0 && tg_stride >= S);
    const int S_ = 2 * S + 1;
    const size_t shmem = (size_t)(3 * S_ + 4) * sizeof(float);
    if (shmem > 64 * 1024) return HALO_ENOTSUP;
    CtcBwdArgs a;
    a.lp = lp; a.stride_t = stride_t; a.stride_n = stride_n; a.T = T; a.N = N; a.C = C;
    a.targets = targets; a.tg_stride = tg_stride; a.S = S; a.il = input_lengths; a.tl = target_lengths;
    a.alpha = alpha; a.nll = nll; a.grad_out = grad_out; a.beta = beta; a.grad = grad;
    a.gstride_t = gstride_t; a.gstride_n = gstride_n;
    const size_t wave_shmem = ((size_t)T * C + (size_t)T * S_) * sizeof(float);
    if (S_ <= 64 && wave_shmem <= 60 * 1024) {
        hipLaunchKernelGGL(ctc_beta_grad_wave_kernel, dim3(N), dim3(256), wave_shmem, (hipStream_t)stream, a);
    } else {
        int block = block_for_states(S_ > C ? S_ : C);
        hipLaunchKernelGGL(ctc_beta_grad_kernel, dim3(N), dim3(block), shmem, (hipStream_t)stream, a);
    }
    return halo_launch_status();
}

int halo_ctc_greedy(const float *lp, int N, int T, int C, int64_t *alignments, float *scores, int64_t *hyp,
                    int64_t *hyp_len, halo_stream_t stream) {
    HALO_CHECK_ARG(lp && alignments && scores && hyp && hyp_len);
    HALO_CHECK_ARG(N > 0 && T > 0 && C > 0);
    hipLaunchKernelGGL(ctc_greedy_kernel, dim3(N), dim3(64), 0, (hipStream_t)stream, lp, N, T, C, alignments, scores, hyp,
                       hyp_len);
    return halo_launch_status();
}

}  // extern "C"
