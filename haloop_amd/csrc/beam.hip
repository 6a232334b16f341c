// Beam search over CTC emissions with the reference's exact semantics (ha/beam.py:71-137 and its
// probability-domain twin beam.py:5-68), one 256-thread workgroup per utterance.
//
// What is reproduced: extension candidates enter with blank score 0.0
// in the log domain; blank (k = 0) is proposed as an output symbol; equal prefixes are never
// merged; the parent look-up takes the FIRST kept prefix equal to seq[:-1] and sees the parent's
// blank score already advanced to this frame iff parent index < s; candidates are ordered
// [kept prefixes..., prefix0+0, prefix0+1, ...]; ranking is a descending top-k.
//
// The reference's per-prefix Python loop is order dependent only through "blank[p] already
// updated iff p < s"; since the updated value total[p] + e[blank] does not depend on the loop,
// all prefixes are processed in parallel with that rule applied explicitly.
#include "halo_common.h"

namespace {

struct BeamArgs {
    const float *em;      // [N,T,V]
    int N, T, V, beam, log_domain;
    int64_t *seqs;        // [N,beam,T]
    int32_t *lens;        // [N,beam]
    float *scores;        // [N,beam]
    int32_t *ws_seq;      // [N,2,beam,T]
    float *ws_cand;       // [N,beam*(1+V)]
    int32_t *ws_taken;    // [N,beam*(1+V)]
};

struct Best {
    float v;
    int i;
};

__device__ __forceinline__ bool better(float v2, int i2, float v, int i) { return v2 > v || (v2 == v && i2 < i); }

__global__ __launch_bounds__(256) void beam_kernel(const BeamArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int n = blockIdx.x, tid = threadIdx.x, beam = p.beam, V = p.V, T = p.T;
    // small per-prefix state, double buffered where the update needs the old values
    float *const total0 = smem;              // [2][beam]
    float *const blank0 = smem + 2 * beam;   // [2][beam]
    float *const label0 = smem + 4 * beam;   // [2][beam]
    float *const blank_new = smem + 6 * beam;
    float *const label_new = smem + 7 * beam;
    int *const len0 = (int *)(smem + 8 * beam);   // [2][beam]
    int *const parent = (int *)(smem + 10 * beam);
    int *const sel = (int *)(smem + 11 * beam);
    __shared__ float red_v[4];
    __shared__ int red_i[4];

    int32_t *const seq0 = p.ws_seq + (long)n * 2 * beam * T;   // [2][beam][T]
    float *cand = p.ws_cand + (long)n * beam * (1 + V);
    int32_t *taken = p.ws_taken + (long)n * beam * (1 + V);
    const bool logd = p.log_domain != 0;
    const float zero_score = logd ? 0.f : 1.f;      // probability 1
    const float none_score = logd ? -INFINITY : 0.f; // probability 0

    int cur = 0, nb = 1;
    if (tid == 0) {
        total0[0] = zero_score; blank0[0] = zero_score; label0[0] = none_score; len0[0] = 0;
    }
    __syncthreads();

    for (int t = 0; t < T; ++t) {
        const float *e = p.em + ((long)n * T + t) * V;
        const int nxt = cur ^ 1;
        const float *tot = total0 + cur * beam, *blk = blank0 + cur * beam, *lab = label0 + cur * beam;
        const int *ln = len0 + cur * beam;
        const int32_t *sq = seq0 + (long)cur * beam * T;
        int32_t *sq_next = seq0 + (long)nxt * beam * T;

        for (int s = tid; s < nb; s += 256) {
            blank_new[s] = logd ? tot[s] + e[0] : tot[s] * e[0];
            parent[s] = 0x7fffffff;
        }
        __syncthreads();
        // first kept prefix equal to seq[:-1]
        for (int pair = tid; pair < nb * nb; pair += 256) {
            const int s = pair / nb, q = pair % nb;
            const int ls = ln[s];
            if (ls > 0 && q != s && ln[q] == ls - 1) {
                bool same = true;
                for (int k = 0; k < ls - 1 && same; ++k) same = sq[(long)s * T + k] == sq[(long)q * T + k];
                if (same) atomicMin(&parent[s], q);
            }
        }
        __syncthreads();
        for (int s = tid; s < nb; s += 256) {
            float l = lab[s];
            if (ln[s] > 0) {
                const float el = e[sq[(long)s * T + ln[s] - 1]];
                l = logd ? l + el : l * el;
                const int q = parent[s];
                if (q != 0x7fffffff) {
                    const float bq = q < s ? blank_new[q] : blk[q];
                    l = logd ? log_add_exp(l, el + bq) : l + el * bq;
                }
            }
            label_new[s] = l;
        }
        __syncthreads();
        const int ncand = nb * (1 + V);
        for (int i = tid; i < ncand; i += 256) {
            float b, l;
            if (i < nb) { b = blank_new[i]; l = label_new[i]; }
            else {
                const int s = (i - nb) / V, k = (i - nb) % V;
                const int pivot = ln[s] > 0 ? sq[(long)s * T + ln[s] - 1] : 0;
                const float base = k == pivot ? blank_new[s] : tot[s];
                b = logd ? 0.f : 0.f;
                l = logd ? e[k] + base : e[k] * base;
            }
            cand[i] = logd ? log_add_exp(b, l) : b + l;
            taken[i] = 0;
        }
        __syncthreads();
        // descending top-k by repeated arg-max (lowest index wins ties)
        for (int j = 0; j < beam; ++j) {
            Best b = {-INFINITY, 0x7fffffff};
            for (int i = tid; i < ncand; i += 256) {
                if (!taken[i]) {
                    const float v = cand[i];
                    if (b.i == 0x7fffffff || better(v, i, b.v, b.i)) { b.v = v; b.i = i; }
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float v2 = __shfl_xor(b.v, o, 64);
                const int i2 = __shfl_xor(b.i, o, 64);
                if (i2 != 0x7fffffff && (b.i == 0x7fffffff || better(v2, i2, b.v, b.i))) { b.v = v2; b.i = i2; }
            }
            if ((tid & 63) == 0) { red_v[tid >> 6] = b.v; red_i[tid >> 6] = b.i; }
            __syncthreads();
            if (tid == 0) {
                Best r = {red_v[0], red_i[0]};
                for (int w = 1; w < 4; ++w)
                    if (red_i[w] != 0x7fffffff && (r.i == 0x7fffffff || better(red_v[w], red_i[w], r.v, r.i))) {
                        r.v = red_v[w]; r.i = red_i[w];
                    }
                sel[j] = r.i;
                taken[r.i] = 1;
            }
            __syncthreads();
        }
        // new per-prefix state
        for (int j = tid; j < beam; j += 256) {
            const int i = sel[j];
            float b, l;
            int nl;
            if (i < nb) { b = blank_new[i]; l = label_new[i]; nl = ln[i]; }
            else {
                const int s = (i - nb) / V, k = (i - nb) % V;
                const int pivot = ln[s] > 0 ? sq[(long)s * T + ln[s] - 1] : 0;
                const float base = k == pivot ? blank_new[s] : tot[s];
                b = 0.f;
                l = logd ? e[k] + base : e[k] * base;
                nl = ln[s] + 1;
            }
            total0[nxt * beam + j] = cand[i];
            blank0[nxt * beam + j] = b;
            label0[nxt * beam + j] = l;
            len0[nxt * beam + j] = nl;
        }
        // new sequences: copy the parent's symbols, append k for extensions
        for (int w = tid; w < beam * T; w += 256) {
            const int j = w / T, k = w % T;
            const int i = sel[j];
            const int s = i < nb ? i : (i - nb) / V;
            const int pl = ln[s];
            int sym = 0;
            if (k < pl) sym = sq[(long)s * T + k];
            else if (k == pl && i >= nb) sym = (i - nb) % V;
            sq_next[(long)j * T + k] = sym;
        }
        __syncthreads();
        cur = nxt;
        nb = beam;
    }
    for (int w = tid; w < beam * T; w += 256) p.seqs[(long)n * beam * T + w] = seq0[(long)cur * beam * T + w];
    for (int j = tid; j < beam; j += 256) {
        p.lens[(long)n * beam + j] = len0[cur * beam + j];
        p.scores[(long)n * beam + j] = total0[cur * beam + j];
    }
}

}  // namespace

extern "C" {

size_t halo_ctc_beam_workspace_bytes(int N, int T, int V, int beam) {
    if (N <= 0 || T <= 0 || V <= 0 || beam <= 0) return 0;
    const size_t seq = (size_t)N * 2 * beam * T * sizeof(int32_t);
    const size_t cand = (size_t)N * beam * (1 + V) * (sizeof(float) + sizeof(int32_t));
    return seq + cand;
}

int halo_ctc_beam(const float *em, int N, int T, int V, int beam, int log_domain, int64_t *seqs, int32_t *lens,
                  float *scores, void *workspace, halo_stream_t stream) {
    HALO_CHECK_ARG(em && seqs && lens && scores && workspace);
    HALO_CHECK_ARG(N > 0 && T > 0 && V > 0 && beam > 0);
    HALO_CHECK_ARG(beam <= 1 + V);    // the reference's topk raises at t = 0 otherwise
    const size_t shmem = (size_t)12 * beam * sizeof(float);
    if (shmem > 48 * 1024) return HALO_ENOTSUP;
    BeamArgs a;
    a.em = em; a.N = N; a.T = T; a.V = V; a.beam = beam; a.log_domain = log_domain;
    a.seqs = seqs; a.lens = lens; a.scores = scores;
    a.ws_seq = (int32_t *)workspace;
    a.ws_cand = (float *)(a.ws_seq + (size_t)N * 2 * beam * T);
    a.ws_taken = (int32_t *)(a.ws_cand + (size_t)N * beam * (1 + V));
    hipLaunchKernelGGL(beam_kernel, dim3(N), dim3(256), shmem, (hipStream_t)stream, a);
    return halo_launch_status();
}

}  // extern "C"
