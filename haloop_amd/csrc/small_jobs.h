// Small reductions that ride in another launch.  A training step ends in several fixed-order sums of a few hundred KiB to a few MiB -- the
// CTC head's per-utterance weight-gradient partials, the LSTM bias gradients' batch-tile partials -- whose results nothing reads before the
// optimizer; each was a ~5 us launch of its own.  With halo_set_defer_small_jobs(1) their producers queue them on the context instead
// (host bookkeeping only) and the next launch that can carry extra workgroups (the conv backward's reduce, or halo_flush_small_jobs)
// runs them from its tail blocks: same arithmetic, same order, one launch.
#pragma once
#include "halo_common.h"

constexpr int HALO_SMALL_JOBS_MAX = 4;
struct HaloSmallJob {
    // kind 1: o1[v][k] = sum_n a[n][v][k] (len = V*H elements), o2[v] = sum_n b[n][v] (m = V): 64 elements per block, its four waves a
    //         quarter of the n each (fixed order inside a quarter, then quarter 0 + 1 + 2 + 3) -- the CTC head's reduce
    // kind 2: o1[c] (and o2[c]) = sum over the n rows of a [n][len], rows in order, one thread per column -- a bias gradient's partials
    int kind, n, m, blocks;
    long len;
    const float *a, *b;
    float *o1, *o2;
    float *ss;       // kind 2, optional: block i also writes the sum of the squares of what it stored (both outputs) to ss[i]
};
struct HaloSmallJobs {
    HaloSmallJob job[HALO_SMALL_JOBS_MAX];
    int n, blocks;
};

inline int halo_small_job_blocks(const HaloSmallJob &j) { return j.kind == 1 ? (int)(j.len / 64 + (j.m + 63) / 64) : (int)((j.len + 255) / 256); }

// block `local` (0 .. blocks-1 over all queued jobs) of the queue; 256 threads; part: 4 x 64 floats of LDS
__device__ __forceinline__ void halo_small_jobs_block(const HaloSmallJobs &q, int local, float (*part)[64]) {
    int ji = 0;
#pragma unroll
    for (int i = 0; i < HALO_SMALL_JOBS_MAX - 1; ++i)
        if (ji < q.n - 1 && local >= q.job[ji].blocks) { local -= q.job[ji].blocks; ++ji; }
    const HaloSmallJob &j = q.job[ji];
    if (j.kind == 1) {
        const int e = threadIdx.x & 63, w = threadIdx.x >> 6;
        const long i = (long)local * 64 + e;
        const int per = (j.n + 3) / 4, n0 = w * per, n1 = min(j.n, n0 + per);
        const bool is_b = i >= j.len;                    // the blocks behind the weight elements sum the bias partials
        const long k = is_b ? i - j.len : i, count = is_b ? (long)j.m : j.len;
        const float *src = is_b ? j.b : j.a;
        float s = 0.f;
        if (k < count) {
#pragma unroll 8
            for (int n = n0; n < n1; ++n) s += src[(long)n * count + k];
        }
        part[w][e] = s;
        __syncthreads();
        if (w == 0 && k < count) (is_b ? j.o2 : j.o1)[k] = ((part[0][e] + part[1][e]) + part[2][e]) + part[3][e];
    } else {
        const long c = (long)local * 256 + threadIdx.x;
        float s = 0.f;
        if (c < j.len) {
            for (int r = 0; r < j.n; ++r) s += j.a[(long)r * j.len + c];
            j.o1[c] = s;
            if (j.o2) j.o2[c] = s;
        }
        if (j.ss) {      // (uniform over the block)
            float q = wave_sum(s * s);
            if ((threadIdx.x & 63) == 0) part[0][threadIdx.x >> 6] = q;
            __syncthreads();
            if (threadIdx.x == 0) j.ss[local] = ((part[0][0] + part[0][1]) + (part[0][2] + part[0][3])) * (j.o2 ? 2.f : 1.f);
        }
    }
}
