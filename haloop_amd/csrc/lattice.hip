// The reference's two further sequence lattices, one workgroup per utterance each, forward score and alpha-beta gradient:
//
//   star-CTC (ha/star.py:65-166; [Pratap22] Star Temporal Classification).  4S+3 states for labels a b c:
//       blank <star\a> blank a blank <star\b> blank b blank <star\c> blank c blank <star> blank
//   <star> = logsumexp of all non-blank symbols, <star\s> = logsubexp(<star>, s) (ha/star.py:9-41; the emissions are never widened to
//   2V here: the kernel evaluates the two star values where a state needs them).  Into state i at frame t from frame t-1 (:112-135):
//       blank  i-1, i          star  i-1, i, i+1 (+ star_penalty)          label  i-3, i-2, i-1 and i-4 unless both labels are equal
//   (labels have no self loop; a star is also entered from the blank AFTER it -- both as the reference has them).  Four virtual
//   states before state 0 hold 0 at frame 0 (:93).  "log 0" is finfo(float32).min (:91).  The recursion covers all T frames and all
//   states; the lengths pick the read-out: frame emission_lengths[n], the four states 4*tl-1 .. 4*tl+2 (:153-162).
//
//   transducer (ha/transducer.py:175-207; [Graves12]): joint [N, T, U+1, K] log-probabilities, blank 0,
//       alpha[t, 0] = sum_{t'<t} joint[t', 0, 0];  alpha[t, u] = logaddexp(alpha[t, u-1] + joint[t, u-1, y[u-1]], alpha[t-1, u] + joint[t-1, u, 0])
//       loss = -(alpha[T_n-1, U_n] + joint[T_n-1, U_n, 0]).
//   The reference runs the t recurrence as a log-space parallel scan padded to 2**round(log2 T) (and raises when that is < T); the kernel
//   walks the anti-diagonals of the lattice, any T.
//
// Gradients: the reference differentiates these functions with autograd; the kernels compute the same derivatives as state / cell
// occupancies from a backward (beta) sweep over the saved alpha lattice (oracle/star_ref.py states both and is pinned to the
// reference's autograd gradients).  fp32 throughout; logaddexp as torch defines it (halo_common.h).
#include "halo_common.h"
#include "halo_internal.h"

namespace {

constexpr float VOID_F = -3.4028234663852886e38f;     // finfo(float32).min

struct StarArgs {
    const float *lp;            // [T][N][C] log-probabilities (strides in elements)
    long st, sn;
    int T, N, C, S;
    const int64_t *targets;     // [N][S]
    const int64_t *em_len, *tg_len;
    float penalty;
    float *alpha;               // [N][T][4S+3] (may be NULL in the forward)
    float *cstar;               // [N][T] the complete-star value of every frame (may be NULL in the forward)
    float *losses;              // [N]
    const float *gout;          // [N] upstream gradient (backward)
    float *grad;                // [T][N][C] (backward), same strides as lp
};

// emission of lattice state i (ha/star.py:46-49: ids into the widened emissions) from the frame's log-probabilities
__device__ __forceinline__ float star_emission(const float *lp, float cs, const int64_t *tg, int S, int i) {
    if ((i & 1) == 0) return lp[0];
    const int k = i >> 2;
    if ((i & 3) == 3) return lp[tg[k]];
    const int64_t l = k < S ? tg[k] : 0;                        // the last star is the complete one; so is <star\0> (index V + 0)
    if (l == 0) return cs;
    return cs + log1pf(-expf(lp[l] - cs));
}

// cs[t] = logsumexp(lp[t, 1:]) for every frame: wave w takes frames w, w + nwaves, ...
__device__ __forceinline__ void star_complete(const StarArgs &p, int n, float *cs) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int t = wave; t < p.T; t += nw) {
        const float *row = p.lp + (long)t * p.st + (long)n * p.sn;
        float m = -INFINITY;
        for (int c = 1 + lane; c < p.C; c += 64) m = fmaxf(m, row[c]);
        m = wave_max(m);
        float s = 0.f;
        for (int c = 1 + lane; c < p.C; c += 64) s += expf(row[c] - m);
        s = wave_sum(s);
        if (lane == 0) cs[t] = (m == -INFINITY) ? -INFINITY : m + logf(s);
    }
}

// dynamic LDS: cs[T] | a[S_] | b[S_]
__global__ __launch_bounds__(256) void star_alpha_kernel(const StarArgs p) {
    extern __shared__ float lds[];
    const int n = blockIdx.x, S_ = 4 * p.S + 3;
    float *cs = lds, *prev = lds + p.T, *cur = prev + S_;
    const int64_t *tg = p.targets + (long)n * p.S;
    star_complete(p, n, cs);
    for (int i = threadIdx.x; i < S_; i += blockDim.x) prev[i] = VOID_F;
    __syncthreads();
    if (p.cstar)
        for (int t = threadIdx.x; t < p.T; t += blockDim.x) p.cstar[(long)n * p.T + t] = cs[t];
    for (int t = 1; t <= p.T; ++t) {
        const float *row = p.lp + (long)(t - 1) * p.st + (long)n * p.sn;
        const float virt = t == 1 ? 0.f : VOID_F;                // the four states before state 0
        for (int i = threadIdx.x; i < S_; i += blockDim.x) {
            auto at = [&](int j) { return j < 0 ? virt : (j >= S_ ? -7007.7007f : prev[j]); };
            float tr;
            if ((i & 1) == 0) tr = log_add_exp(at(i - 1), at(i));
            else if ((i & 3) == 1) tr = log_add_exp(log_add_exp(at(i - 1), at(i)), at(i + 1)) + p.penalty;
            else {
                tr = log_add_exp(log_add_exp(at(i - 3), at(i - 1)), at(i - 2));
                const int k = i >> 2;
                const bool same = k >= 1 && tg[k] == tg[k - 1];
                if (!same) tr = log_add_exp(tr, at(i - 4));
            }
            const float v = tr + star_emission(row, cs[t - 1], tg, p.S, i);
            cur[i] = v;
            if (p.alpha) p.alpha[((long)n * p.T + (t - 1)) * S_ + i] = v;
        }
        __syncthreads();
        float *tmp = prev; prev = cur; cur = tmp;
        if (t == (int)p.em_len[n] && threadIdx.x == 0) {
            const int s_last = 4 * (int)p.tg_len[n] + 2;
            float z = prev[s_last];
#pragma unroll
            for (int d = 1; d <= 3; ++d) z = log_add_exp(z, s_last - d >= 0 ? prev[s_last - d] : VOID_F);
            p.losses[n] = -z;
        }
    }
}

// dynamic LDS: beta[S_] | nbeta[S_] | eb[S_] | occ_star[S + 1] | g[C]
__global__ __launch_bounds__(256) void star_beta_grad_kernel(const StarArgs p) {
    extern __shared__ float lds[];
    const int n = blockIdx.x, S_ = 4 * p.S + 3;
    float *beta = lds, *nbeta = beta + S_, *eb = nbeta + S_, *occ_star = eb + S_, *g = occ_star + p.S + 1;
    const int64_t *tg = p.targets + (long)n * p.S;
    const int Tn = (int)p.em_len[n], s_last = 4 * (int)p.tg_len[n] + 2;
    const float logz = -p.losses[n], go = p.gout ? p.gout[n] : 1.f;
    const bool live = logz > -1e30f && logz < INFINITY && Tn >= 1 && Tn <= p.T;
    // frames past the utterance's length (and everything of an unreachable lattice) carry no gradient
    for (int t = live ? Tn : 0; t < p.T; ++t)
        for (int c = threadIdx.x; c < p.C; c += blockDim.x) p.grad[(long)t * p.st + (long)n * p.sn + c] = 0.f;
    if (!live) return;
    for (int i = threadIdx.x; i < S_; i += blockDim.x) beta[i] = (i <= s_last && i >= s_last - 3) ? 0.f : -INFINITY;
    __syncthreads();
    for (int t = Tn; t >= 1; --t) {
        const float *row = p.lp + (long)(t - 1) * p.st + (long)n * p.sn;
        const float cs = p.cstar[(long)n * p.T + (t - 1)];
        const float *al = p.alpha + ((long)n * p.T + (t - 1)) * S_;
        for (int c = threadIdx.x; c < p.C; c += blockDim.x) g[c] = 0.f;
        __syncthreads();
        for (int i = threadIdx.x; i < S_; i += blockDim.x) {
            const float b = beta[i];
            const float occ = b == -INFINITY ? 0.f : expf(al[i] + b - logz);
            eb[i] = star_emission(row, cs, tg, p.S, i) + b;
            if ((i & 3) == 1) occ_star[i >> 2] = occ;
            else if (occ != 0.f) atomicAdd(&g[(i & 1) ? (int)tg[i >> 2] : 0], occ);
        }
        __syncthreads();
        // the star occupancies spread over the symbols they sum: d <star> / d lp[k] = exp(lp[k] - <star>),
        // d <star\s> / d lp[k] = exp(lp[k] - <star\s>) for k != s  (k >= 1)
        for (int c = 1 + threadIdx.x; c < p.C; c += blockDim.x) {
            float acc = g[c];
            const float x = row[c];
            for (int k = 0; k <= p.S; ++k) {
                const float o = occ_star[k];
                if (o == 0.f) continue;
                const int64_t l = k < p.S ? tg[k] : 0;
                if (l == c) continue;
                const float se = l == 0 ? cs : cs + log1pf(-expf(row[l] - cs));
                acc += o * expf(x - se);
            }
            g[c] = acc;
        }
        __syncthreads();
        for (int c = threadIdx.x; c < p.C; c += blockDim.x) p.grad[(long)(t - 1) * p.st + (long)n * p.sn + c] = -go * g[c];
        // beta of frame t-1: state j is followed by j (blank / star), j+1, j-1 (the star before a blank), and the labels at j+2 (from a
        // star), j+3 (from the blank before the star), j+4 (from the previous label, unless equal)
        for (int j = threadIdx.x; j < S_; j += blockDim.x) {
            float acc = -INFINITY;
            auto take = [&](int i, float w) { if (i >= 0 && i < S_ && eb[i] != -INFINITY) acc = log_add_exp(acc, eb[i] + w); };
            const int m = j & 3;
            if (m != 3) take(j, m == 1 ? p.penalty : 0.f);
            take(j + 1, ((j + 1) & 3) == 1 ? p.penalty : 0.f);
            if (m == 2) take(j - 1, p.penalty);
            if (m == 1) take(j + 2, 0.f);
            if (m == 0) take(j + 3, 0.f);
            if (m == 3 && j + 4 < S_) {
                const int k = (j + 4) >> 2;
                if (tg[k] != tg[k - 1]) take(j + 4, 0.f);
            }
            nbeta[j] = acc;
        }
        __syncthreads();
        float *tmp = beta; beta = nbeta; nbeta = tmp;
    }
}

// ------------------------------------------------------------------------------------------------ transducer
struct TransducerArgs {
    const float *joint;          // [N][T][U1][K] contiguous
    int N, T, U1, K;
    const int64_t *targets;      // [N][U1 - 1]
    const int *j_len, *t_len;    // [N]
    float *alpha;                // [N][T][U1] (may be NULL in the forward)
    float *losses;
    const float *gout;
    float *grad;                 // [N][T][U1][K]
};

// dynamic LDS: two diagonals of U1 floats
__global__ __launch_bounds__(256) void transducer_alpha_kernel(const TransducerArgs p) {
    extern __shared__ float lds[];
    const int n = blockIdx.x, T = p.T, U1 = p.U1;
    float *dprev = lds, *dcur = lds + U1;
    const float *jn = p.joint + (long)n * T * U1 * p.K;
    const int64_t *tg = p.targets + (long)n * (U1 - 1);
    const int Tn = p.j_len[n], Un = p.t_len[n];
    for (int d = 0; d <= T + U1 - 2; ++d) {
        for (int u = threadIdx.x; u < U1; u += blockDim.x) {
            const int t = d - u;
            if (t < 0 || t >= T) continue;
            float v;
            if (u == 0) v = t == 0 ? 0.f : dprev[0] + jn[((long)(t - 1) * U1) * p.K];
            else {
                const float bot = dprev[u - 1] + jn[((long)t * U1 + (u - 1)) * p.K + tg[u - 1]];
                v = t == 0 ? bot : log_add_exp(bot, dprev[u] + jn[((long)(t - 1) * U1 + u) * p.K]);
            }
            dcur[u] = v;
            if (p.alpha) p.alpha[((long)n * T + t) * U1 + u] = v;
            if (t == Tn - 1 && u == Un) p.losses[n] = -(v + jn[((long)t * U1 + u) * p.K]);
        }
        __syncthreads();
        float *tmp = dprev; dprev = dcur; dcur = tmp;
    }
}

__global__ __launch_bounds__(256) void transducer_beta_grad_kernel(const TransducerArgs p) {
    extern __shared__ float lds[];
    const int n = blockIdx.x, T = p.T, U1 = p.U1, K = p.K;
    float *dnext = lds, *dcur = lds + U1 + 1;
    const float *jn = p.joint + (long)n * T * U1 * K;
    float *gn = p.grad + (long)n * T * U1 * K;
    const float *an = p.alpha + (long)n * T * U1;
    const int64_t *tg = p.targets + (long)n * (U1 - 1);
    const int Tn = p.j_len[n], Un = p.t_len[n];
    const float logz = -p.losses[n], go = p.gout ? p.gout[n] : 1.f;
    for (long e = threadIdx.x; e < (long)T * U1 * K; e += blockDim.x) gn[e] = 0.f;
    if (!(logz > -1e30f && logz < INFINITY) || Tn < 1 || Tn > T || Un < 0 || Un >= U1) return;
    for (int u = threadIdx.x; u <= U1; u += blockDim.x) dnext[u] = dcur[u] = -INFINITY;
    __syncthreads();
    for (int d = Tn - 1 + Un; d >= 0; --d) {
        for (int u = threadIdx.x; u <= Un; u += blockDim.x) {
            const int t = d - u;
            if (t < 0 || t >= Tn) { dcur[u] = -INFINITY; continue; }
            const long cell = (long)t * U1 + u;
            const float jb = jn[cell * K], a = an[cell];
            float b;
            if (t == Tn - 1 && u == Un) {
                b = jb;
                gn[cell * K] -= go;                                      // occupancy 1: every path ends with this blank
            } else {
                const float via_blank = dnext[u] == -INFINITY ? -INFINITY : dnext[u] + jb;         // beta[t + 1, u]
                b = via_blank;
                if (via_blank != -INFINITY) gn[cell * K] -= go * expf(a + via_blank - logz);
                if (u < Un) {
                    const int64_t y = tg[u];
                    const float jy = jn[cell * K + y];
                    const float via_label = dnext[u + 1] == -INFINITY ? -INFINITY : dnext[u + 1] + jy;  // beta[t, u + 1]
                    if (via_label != -INFINITY) {
                        gn[cell * K + y] -= go * expf(a + via_label - logz);
                        b = log_add_exp(b, via_label);
                    }
                }
            }
            dcur[u] = b;
        }
        if (threadIdx.x == 0) dcur[Un + 1] = -INFINITY;
        __syncthreads();
        float *tmp = dnext; dnext = dcur; dcur = tmp;
    }
}

}  // namespace

extern "C" {

size_t halo_star_ctc_workspace_bytes(int T, int N, int S) {
    if (T <= 0 || N <= 0 || S <= 0) return 0;
    return ((size_t)N * T * (4 * (size_t)S + 3) + (size_t)N * T) * sizeof(float);
}

static bool star_args(StarArgs &p, const float *lp, long stride_t, long stride_n, int T, int N, int C, const int64_t *targets, int S,
                      const int64_t *em_len, const int64_t *tg_len, float penalty, void *workspace, float *losses) {
    if (!lp || !targets || !em_len || !tg_len || !losses || T <= 0 || N <= 0 || C < 2 || S <= 0) return false;
    p.lp = lp; p.st = stride_t; p.sn = stride_n; p.T = T; p.N = N; p.C = C; p.S = S; p.targets = targets; p.em_len = em_len;
    p.tg_len = tg_len; p.penalty = penalty; p.losses = losses;
    p.alpha = (float *)workspace;
    p.cstar = workspace ? p.alpha + (size_t)N * T * (4 * (size_t)S + 3) : nullptr;
    p.gout = nullptr; p.grad = nullptr;
    return true;
}

int halo_star_ctc_fwd(const float *log_probs, long stride_t, long stride_n, int T, int N, int C, const int64_t *targets, int S,
                      const int64_t *emission_lengths, const int64_t *target_lengths, float star_penalty, void *workspace, float *losses,
                      halo_stream_t stream) {
    StarArgs p;
    HALO_CHECK_ARG(star_args(p, log_probs, stride_t, stride_n, T, N, C, targets, S, emission_lengths, target_lengths, star_penalty,
                             workspace, losses));
    const size_t lds = ((size_t)T + 2 * (4 * (size_t)S + 3)) * sizeof(float);
    HALO_CHECK_ARG(lds <= 60 * 1024);
    hipLaunchKernelGGL(star_alpha_kernel, dim3(N), dim3(256), lds, (hipStream_t)stream, p);
    return halo_launch_status();
}

int halo_star_ctc_bwd(const float *log_probs, long stride_t, long stride_n, int T, int N, int C, const int64_t *targets, int S,
                      const int64_t *emission_lengths, const int64_t *target_lengths, float star_penalty, const void *workspace,
                      const float *losses, const float *grad_losses, float *grad_log_probs, halo_stream_t stream) {
    StarArgs p;
    HALO_CHECK_ARG(workspace && grad_log_probs);
    HALO_CHECK_ARG(star_args(p, log_probs, stride_t, stride_n, T, N, C, targets, S, emission_lengths, target_lengths, star_penalty,
                             (void *)workspace, (float *)losses));
    p.gout = grad_losses; p.grad = grad_log_probs;
    const size_t lds = (3 * (4 * (size_t)S + 3) + S + 1 + (size_t)C) * sizeof(float);
    HALO_CHECK_ARG(lds <= 60 * 1024);
    hipLaunchKernelGGL(star_beta_grad_kernel, dim3(N), dim3(256), lds, (hipStream_t)stream, p);
    return halo_launch_status();
}

size_t halo_transducer_workspace_bytes(int N, int T, int U1) {
    if (N <= 0 || T <= 0 || U1 <= 0) return 0;
    return (size_t)N * T * U1 * sizeof(float);
}

int halo_transducer_fwd(const float *joint, int N, int T, int U1, int K, const int64_t *targets, const int *joint_lengths,
                        const int *target_lengths, void *workspace, float *losses, halo_stream_t stream) {
    HALO_CHECK_ARG(joint && targets && joint_lengths && target_lengths && losses && N > 0 && T > 0 && U1 >= 2 && K > 0);
    HALO_CHECK_ARG(2 * (size_t)U1 * sizeof(float) <= 60 * 1024);
    TransducerArgs p;
    p.joint = joint; p.N = N; p.T = T; p.U1 = U1; p.K = K; p.targets = targets; p.j_len = joint_lengths; p.t_len = target_lengths;
    p.alpha = (float *)workspace; p.losses = losses; p.gout = nullptr; p.grad = nullptr;
    hipLaunchKernelGGL(transducer_alpha_kernel, dim3(N), dim3(256), 2 * (size_t)U1 * sizeof(float), (hipStream_t)stream, p);
    return halo_launch_status();
}

int halo_transducer_bwd(const float *joint, int N, int T, int U1, int K, const int64_t *targets, const int *joint_lengths,
                        const int *target_lengths, const void *workspace, const float *losses, const float *grad_losses,
                        float *grad_joint, halo_stream_t stream) {
    HALO_CHECK_ARG(joint && targets && joint_lengths && target_lengths && losses && workspace && grad_joint);
    HALO_CHECK_ARG(N > 0 && T > 0 && U1 >= 2 && K > 0 && 2 * ((size_t)U1 + 1) * sizeof(float) <= 60 * 1024);
    TransducerArgs p;
    p.joint = joint; p.N = N; p.T = T; p.U1 = U1; p.K = K; p.targets = targets; p.j_len = joint_lengths; p.t_len = target_lengths;
    p.alpha = (float *)workspace; p.losses = (float *)losses; p.gout = grad_losses; p.grad = grad_joint;
    hipLaunchKernelGGL(transducer_beta_grad_kernel, dim3(N), dim3(256), 2 * ((size_t)U1 + 1) * sizeof(float), (hipStream_t)stream, p);
    return halo_launch_status();
}

}  // extern "C"
