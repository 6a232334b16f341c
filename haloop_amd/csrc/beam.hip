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
    int32_t *ws_taken;    // [N,beam*(1+V)]  queue indices when the queue does not fit in LDS
    float *ws_qv;         // [N,beam*(1+V)]  queue values, same case
    int queue_in_lds;
};


// ---- torch.topk(largest=True, sorted=True) on CPU, replicated -----------------------------------
// The reference ranks candidates with torch.topk (ha/beam.py:129).  On flat emission distributions
// hypotheses that differ only in early symbols converge to EXACTLY equal fp32 scores, so which ones
// survive is decided by topk's order among equal keys.  ATen's CPU kernel (TopKImpl.h) runs, on
// (value, index) pairs with comparator  x > y  (NaN first):
//     k*64 <= n :  std::partial_sort(first, first+k, last)
//     otherwise :  std::nth_element(first, first+k-1, last); std::sort(first, first+k-1)
// Below is libstdc++'s algorithm for each (introselect / introsort / heap), executed by one thread,
// exported on its own as halo_topk_f32 and checked against torch.topk on tie-heavy inputs in
// tests/test_gpu_parity.py::test_topk_replica_matches_torch_cpu.
struct TopQ {
    float *v;
    int *i;
};
__device__ __forceinline__ bool qcmp(float x, float y) { return (x != x && y == y) || x > y; }
__device__ __forceinline__ void qswap(const TopQ &q, int a, int b) {
    const float tv = q.v[a]; q.v[a] = q.v[b]; q.v[b] = tv;
    const int ti = q.i[a]; q.i[a] = q.i[b]; q.i[b] = ti;
}
__device__ __forceinline__ void qmove(const TopQ &q, int dst, int src) { q.v[dst] = q.v[src]; q.i[dst] = q.i[src]; }
__device__ inline int ilog2(int n) { return 31 - __clz(n); }

__device__ void q_median_to_first(const TopQ &q, int result, int a, int b, int c) {
    const float va = q.v[a], vb = q.v[b], vc = q.v[c];
    if (qcmp(va, vb)) {
        if (qcmp(vb, vc)) qswap(q, result, b);
        else if (qcmp(va, vc)) qswap(q, result, c);
        else qswap(q, result, a);
    } else if (qcmp(va, vc)) qswap(q, result, a);
    else if (qcmp(vb, vc)) qswap(q, result, c);
    else qswap(q, result, b);
}
__device__ int q_unguarded_partition(const TopQ &q, int first, int last, int pivot) {
    while (true) {
        while (qcmp(q.v[first], q.v[pivot])) ++first;
        --last;
        while (qcmp(q.v[pivot], q.v[last])) --last;
        if (!(first < last)) return first;
        qswap(q, first, last);
        ++first;
    }
}
__device__ int q_partition_pivot(const TopQ &q, int first, int last) {
    const int mid = first + (last - first) / 2;
    q_median_to_first(q, first, first + 1, mid, last - 1);
    return q_unguarded_partition(q, first + 1, last, first);
}
__device__ void q_unguarded_linear_insert(const TopQ &q, int last) {
    const float vv = q.v[last]; const int vi = q.i[last];
    int next = last - 1;
    while (qcmp(vv, q.v[next])) { qmove(q, last, next); last = next; --next; }
    q.v[last] = vv; q.i[last] = vi;
}
__device__ void q_insertion_sort(const TopQ &q, int first, int last) {
    if (first == last) return;
    for (int i = first + 1; i != last; ++i) {
        if (qcmp(q.v[i], q.v[first])) {
            const float vv = q.v[i]; const int vi = q.i[i];
            for (int j = i; j > first; --j) qmove(q, j, j - 1);
            q.v[first] = vv; q.i[first] = vi;
        } else {
            q_unguarded_linear_insert(q, i);
        }
    }
}
__device__ void q_push_heap(const TopQ &q, int first, int hole, int top, float vv, int vi) {
    int parent = (hole - 1) / 2;
    while (hole > top && qcmp(q.v[first + parent], vv)) {
        qmove(q, first + hole, first + parent);
        hole = parent;
        parent = (hole - 1) / 2;
    }
    q.v[first + hole] = vv; q.i[first + hole] = vi;
}
__device__ void q_adjust_heap(const TopQ &q, int first, int hole, int len, float vv, int vi) {
    const int top = hole;
    int second = hole;
    while (second < (len - 1) / 2) {
        second = 2 * (second + 1);
        if (qcmp(q.v[first + second], q.v[first + second - 1])) --second;
        qmove(q, first + hole, first + second);
        hole = second;
    }
    if ((len & 1) == 0 && second == (len - 2) / 2) {
        second = 2 * (second + 1);
        qmove(q, first + hole, first + second - 1);
        hole = second - 1;
    }
    q_push_heap(q, first, hole, top, vv, vi);
}
__device__ void q_make_heap(const TopQ &q, int first, int last) {
    const int len = last - first;
    if (len < 2) return;
    for (int parent = (len - 2) / 2;; --parent) {
        q_adjust_heap(q, first, parent, len, q.v[first + parent], q.i[first + parent]);
        if (parent == 0) return;
    }
}
__device__ void q_pop_heap(const TopQ &q, int first, int last, int result) {
    const float vv = q.v[result]; const int vi = q.i[result];
    qmove(q, result, first);
    q_adjust_heap(q, first, 0, last - first, vv, vi);
}
__device__ void q_heap_select(const TopQ &q, int first, int middle, int last) {
    q_make_heap(q, first, middle);
    for (int i = middle; i < last; ++i)
        if (qcmp(q.v[i], q.v[first])) q_pop_heap(q, first, middle, i);
}
__device__ void q_sort_heap(const TopQ &q, int first, int last) {
    while (last - first > 1) { --last; q_pop_heap(q, first, last, last); }
}
__device__ void q_introselect(const TopQ &q, int first, int nth, int last, int depth) {
    while (last - first > 3) {
        if (depth == 0) {
            q_heap_select(q, first, nth + 1, last);
            qswap(q, first, nth);
            return;
        }
        --depth;
        const int cut = q_partition_pivot(q, first, last);
        if (cut <= nth) first = cut; else last = cut;
    }
    q_insertion_sort(q, first, last);
}
// std::__introsort_loop without recursion: pending right-hand ranges go on a small stack
__device__ void q_introsort_loop(const TopQ &q, int first, int last, int depth) {
    int stk_first[64], stk_last[64], stk_depth[64], sp = 0;
    while (true) {
        while (last - first > 16) {
            if (depth == 0) {
                q_heap_select(q, first, last, last);
                q_sort_heap(q, first, last);
                break;
            }
            --depth;
            const int cut = q_partition_pivot(q, first, last);
            // recurse into [cut, last) first (as the library does), then continue with [first, cut)
            stk_first[sp] = first; stk_last[sp] = cut; stk_depth[sp] = depth; ++sp;
            first = cut;
        }
        if (sp == 0) return;
        --sp;
        first = stk_first[sp]; last = stk_last[sp]; depth = stk_depth[sp];
    }
}
__device__ void q_std_sort(const TopQ &q, int first, int last) {
    if (first == last) return;
    q_introsort_loop(q, first, last, ilog2(last - first) * 2);
    if (last - first > 16) {
        q_insertion_sort(q, first, first + 16);
        for (int i = first + 16; i != last; ++i) q_unguarded_linear_insert(q, i);
    } else {
        q_insertion_sort(q, first, last);
    }
}
// after the call q.i[0..k) are the selected candidate indices, best first
__device__ void torch_topk_serial(const TopQ &q, int n, int k) {
    if ((long)k * 64 <= n) {
        q_heap_select(q, 0, k, n);
        q_sort_heap(q, 0, k);
    } else {
        if (n > 0 && k - 1 != n) q_introselect(q, 0, k - 1, n, ilog2(n) * 2);
        q_std_sort(q, 0, k - 1);
    }
}

// logaddexp with each libm call evaluated in double and rounded once to float: the value a
// correctly rounded expf/log1pf would give, which is what glibc returns in all but rare cases.
__device__ __forceinline__ float log_add_exp_cr(float a, float b) {
    if (isinf(a) && a == b) return a;
    const float m = fmaxf(a, b);
    const float e = (float)exp((double)(-fabsf(a - b)));
    return m + (float)log1p((double)e);
}

// one workgroup per row: values -> (top-k values, indices) in torch.topk's CPU order
__global__ __launch_bounds__(256) void topk_rows_kernel(const float *__restrict__ v, int n, int k, float *qv_all,
                                                        int *qi_all, float *__restrict__ vals, int64_t *__restrict__ idx) {
    const int row = blockIdx.x;
    float *qv = qv_all + (long)row * n;
    int *qi = qi_all + (long)row * n;
    for (int i = threadIdx.x; i < n; i += 256) { qv[i] = v[(long)row * n + i]; qi[i] = i; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const TopQ q = {qv, qi};
        torch_topk_serial(q, n, k);
        for (int j = 0; j < k; ++j) { vals[(long)row * k + j] = qv[j]; idx[(long)row * k + j] = qi[j]; }
    }
}

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
    // top-k queue: in LDS when it fits, else in the global workspace
    const int ncand_max = beam * (1 + V);
    float *qv;
    int *qi;
    if (p.queue_in_lds) { qv = smem + 12 * beam; qi = (int *)(qv + ncand_max); }
    else { qv = p.ws_qv + (long)n * ncand_max; qi = p.ws_taken + (long)n * ncand_max; }

    int32_t *const seq0 = p.ws_seq + (long)n * 2 * beam * T;   // [2][beam][T]
    float *cand = p.ws_cand + (long)n * beam * (1 + V);
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
                    l = logd ? log_add_exp_cr(l, el + bq) : l + el * bq;
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
            cand[i] = logd ? log_add_exp_cr(b, l) : b + l;
        }
        __syncthreads();
        // ranking: torch.topk's CPU algorithm, serially, on a (value, index) queue
        for (int i = tid; i < ncand; i += 256) { qv[i] = cand[i]; qi[i] = i; }
        __syncthreads();
        if (tid == 0) {
            const TopQ q = {qv, qi};
            torch_topk_serial(q, ncand, beam);
            for (int j = 0; j < beam; ++j) sel[j] = qi[j];
        }
        __syncthreads();
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

int halo_topk_f32(const float *values, int rows, int n, int k, float *out_values, int64_t *out_indices, void *workspace,
                  halo_stream_t stream) {
    HALO_CHECK_ARG(values && out_values && out_indices && workspace);
    HALO_CHECK_ARG(rows > 0 && n > 0 && k > 0 && k <= n);
    float *qv = (float *)workspace;
    int *qi = (int *)(qv + (size_t)rows * n);
    hipLaunchKernelGGL(topk_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, values, n, k, qv, qi, out_values,
                       out_indices);
    return halo_launch_status();
}

size_t halo_ctc_beam_workspace_bytes(int N, int T, int V, int beam) {
    if (N <= 0 || T <= 0 || V <= 0 || beam <= 0) return 0;
    const size_t seq = (size_t)N * 2 * beam * T * sizeof(int32_t);
    const size_t cand = (size_t)N * beam * (1 + V) * (2 * sizeof(float) + sizeof(int32_t));
    return seq + cand;
}

int halo_ctc_beam(const float *em, int N, int T, int V, int beam, int log_domain, int64_t *seqs, int32_t *lens,
                  float *scores, void *workspace, halo_stream_t stream) {
    HALO_CHECK_ARG(em && seqs && lens && scores && workspace);
    HALO_CHECK_ARG(N > 0 && T > 0 && V > 0 && beam > 0);
    HALO_CHECK_ARG(beam <= 1 + V);    // the reference's topk raises at t = 0 otherwise
    size_t shmem = (size_t)12 * beam * sizeof(float);
    if (shmem > 16 * 1024) return HALO_ENOTSUP;
    const size_t qbytes = (size_t)beam * (1 + V) * (sizeof(float) + sizeof(int32_t));
    const int queue_in_lds = shmem + qbytes <= 60 * 1024;
    if (queue_in_lds) shmem += qbytes;
    BeamArgs a;
    a.em = em; a.N = N; a.T = T; a.V = V; a.beam = beam; a.log_domain = log_domain;
    a.seqs = seqs; a.lens = lens; a.scores = scores;
    a.ws_seq = (int32_t *)workspace;
    a.ws_cand = (float *)(a.ws_seq + (size_t)N * 2 * beam * T);
    a.ws_taken = (int32_t *)(a.ws_cand + (size_t)N * beam * (1 + V));
    a.ws_qv = (float *)(a.ws_taken + (size_t)N * beam * (1 + V));
    a.queue_in_lds = queue_in_lds;
    hipLaunchKernelGGL(beam_kernel, dim3(N), dim3(256), shmem, (hipStream_t)stream, a);
    return halo_launch_status();
}

}  // extern "C"
