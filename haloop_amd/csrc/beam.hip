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
#include "halo_internal.h"

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
    int vec_chunk;        // elements per vector iteration of ATen's CPU logaddexp on the reference machine (32: AVX-512)
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

// ---- torch.logaddexp on CPU, replicated bit for bit -----------------------------------------------------------------------
// The reference scores with torch.logaddexp (ha/beam.py:107,127).  On flat emissions hypotheses converge to EXACTLY equal fp32
// scores and the survivor is decided by the last ulp of that function, so the kernel evaluates it the way ATen's CPU kernel does
// (aten/src/ATen/native/cpu/BinaryOpsKernel.cpp, logaddexp_kernel; torch 2.10, x86-64):
//   * whole chunks of 2 * Vectorized<float>::size() elements (32 on AVX-512, 16 on AVX2) of a contiguous operand go through the
//     vector lambda  maximum(a, b) + log1p(exp(-|a - b|))  with Sleef_expf*_u10 and Sleef_log1pf*_u10 (Sleef 3.x, FMA build);
//   * the remaining n mod chunk elements, and every 0-dim call (beam.py:107), go through the scalar lambda with glibc's expf
//     (ARM optimized routines: double arithmetic, 32-entry table) and log1pf (FDLIBM, float arithmetic).
// Each function below restates the published algorithm of that routine; together they reproduce torch.logaddexp bit for bit on
// 3.2 M (vector) + 0.4 M (scalar) random operand pairs (checked on the CPU against torch and glibc 2.35 while this was written).
// All of it runs with contraction off: an fma appears exactly where the library has one.
__device__ __forceinline__ float f_from_bits(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t bits_from_f(float f) { return __builtin_bit_cast(uint32_t, f); }

// Sleef xexpf (u10)
__device__ float sleef_expf_u10(float d) {
#pragma clang fp contract(off)
    const int q = (int)rintf(d * 1.442695040888963407359924681001892137426645954152985934135449406931f);
    float s = __builtin_fmaf((float)q, -0.693145751953125f, d);
    s = __builtin_fmaf((float)q, -1.428606765330187045e-06f, s);
    float u = 0.000198527617612853646278381f;
    u = __builtin_fmaf(u, s, 0.00139304355252534151077271f);
    u = __builtin_fmaf(u, s, 0.00833336077630519866943359f);
    u = __builtin_fmaf(u, s, 0.0416664853692054748535156f);
    u = __builtin_fmaf(u, s, 0.166666671633720397949219f);
    u = __builtin_fmaf(u, s, 0.5f);
    u = 1.0f + __builtin_fmaf(s * s, u, s);
    const int qh = q >> 1;
    u = u * f_from_bits((uint32_t)((qh + 0x7f) << 23)) * f_from_bits((uint32_t)((q - qh + 0x7f) << 23));     // vldexp2
    if (d < -104.0f) u = 0.0f;
    if (d > 100.0f) u = INFINITY;
    return u;
}

struct F2 { float x, y; };   // Sleef's double-float
// Sleef xlog1pf (u10); the AVX-512 (vgetexp) and AVX2 (vilogb2k) bodies compute the same values for 1 + d >= FLT_MIN
__device__ float sleef_log1pf_u10(float d) {
#pragma clang fp contract(off)
    const float dp1 = d + 1.0f;
    const float e = (float)((int)((bits_from_f(dp1 * (1.0f / 0.75f)) >> 23) & 0xff) - 127);
    const float t = f_from_bits(bits_from_f(1.0f) + ((uint32_t)(-(int)e) << 23));                           // vldexp3(1, -e)
    const float m = __builtin_fmaf(d, t, t - 1.0f);
    // s = dfmul((ln2_hi, ln2_lo), e)
    const float l2h = 0.69314718246459960938f, l2l = -1.904654323148236017e-09f;
    F2 sv;
    sv.x = l2h * e;
    sv.y = __builtin_fmaf(l2l, e, __builtin_fmaf(l2h, e, -sv.x));
    // x = dfdiv((m, 0), dfadd(2, m))
    F2 den;
    den.x = 2.0f + m;
    den.y = (2.0f - den.x) + m;
    const float rt = 1.0f / den.x;
    F2 x;
    x.x = m * rt;
    const float uu = __builtin_fmaf(rt, m, -x.x);
    const float vv = __builtin_fmaf(-den.y, rt, __builtin_fmaf(-den.x, rt, 1.0f));
    x.y = __builtin_fmaf(x.x, vv, __builtin_fmaf(0.0f, rt, uu));
    const float x2 = x.x * x.x;
    float tt = 0.3027294874e+0f;
    tt = __builtin_fmaf(tt, x2, 0.3996108174e+0f);
    tt = __builtin_fmaf(tt, x2, 0.6666694880e+0f);
    // s = dfadd(s, dfscale(x, 2)); s = dfadd(s, x2 * x.x * t)
    const float xs_x = x.x * 2.0f, xs_y = x.y * 2.0f;
    float r0 = sv.x + xs_x;
    F2 s1;
    s1.x = r0;
    s1.y = (((sv.x - r0) + xs_x) + sv.y) + xs_y;
    const float add = (x2 * x.x) * tt;
    r0 = s1.x + add;
    F2 s2;
    s2.x = r0;
    s2.y = ((s1.x - r0) + add) + s1.y;
    float r = s2.x + s2.y;
    if (d > 1e+38f) r = INFINITY;
    if (d == -1.0f) r = -INFINITY;
    if (bits_from_f(d) == 0x80000000u) r = -0.0f;
    return r;
}

// glibc 2.35 __expf (sysdeps/ieee754/flt-32/e_expf.c; table 2^(i/32) - (i << 47) as in e_exp2f_data.c)
__device__ const uint64_t EXP2F_T[32] = {0x3ff0000000000000ULL, 0x3fefd9b0d3158574ULL, 0x3fefb5586cf9890fULL, 0x3fef9301d0125b51ULL, 0x3fef72b83c7d517bULL, 0x3fef54873168b9aaULL, 0x3fef387a6e756238ULL, 0x3fef1e9df51fdee1ULL, 0x3fef06fe0a31b715ULL, 0x3feef1a7373aa9cbULL, 0x3feedea64c123422ULL, 0x3feece086061892dULL, 0x3feebfdad5362a27ULL, 0x3feeb42b569d4f82ULL, 0x3feeab07dd485429ULL, 0x3feea47eb03a5585ULL, 0x3feea09e667f3bcdULL, 0x3fee9f75e8ec5f74ULL, 0x3feea11473eb0187ULL, 0x3feea589994cce13ULL, 0x3feeace5422aa0dbULL, 0x3feeb737b0cdc5e5ULL, 0x3feec49182a3f090ULL, 0x3feed503b23e255dULL, 0x3feee89f995ad3adULL, 0x3feeff76f2fb5e47ULL, 0x3fef199bdd85529cULL, 0x3fef3720dcef9069ULL, 0x3fef5818dcfba487ULL, 0x3fef7c97337b9b5fULL, 0x3fefa4afa2a490daULL, 0x3fefd0765b6e4540ULL};
__device__ float glibc_expf(float x) {
#pragma clang fp contract(off)
    const double InvLn2N = 0x1.71547652b82fep+0 * 32, SHIFT = 0x1.8p+52;
    const double C0 = 0x1.c6af84b912394p-5 / 32 / 32 / 32, C1 = 0x1.ebfce50fac4f3p-3 / 32 / 32, C2 = 0x1.62e42ff0c52d6p-1 / 32;
    const uint32_t abstop = (bits_from_f(x) >> 20) & 0x7ff;
    if (abstop >= (0x42b00000u >> 20)) {                       // |x| >= 88
        if (bits_from_f(x) == 0xff800000u) return 0.0f;
        if (abstop >= (0x7f800000u >> 20)) return x + x;
        if (x > 0x1.62e42ep6f) return INFINITY;
        if (x < -0x1.9fe368p6f) return 0.0f;
    }
    double z = InvLn2N * (double)x;
    double kd = z + SHIFT;
    const uint64_t ki = __builtin_bit_cast(uint64_t, kd);
    kd -= SHIFT;
    const double r = z - kd;
    uint64_t t = EXP2F_T[ki % 32];
    t += ki << (52 - 5);
    const double sc = __builtin_bit_cast(double, t);
    z = C0 * r + C1;
    const double r2 = r * r;
    double y = C2 * r + 1.0;
    y = z * r2 + y;
    y = y * sc;
    return (float)y;
}

// glibc 2.35 __log1pf (sysdeps/ieee754/flt-32/s_log1pf.c, FDLIBM), for the arguments logaddexp produces: 0 <= x <= 1 (and inf/nan pass-through)
__device__ float fdlibm_log1pf(float x) {
#pragma clang fp contract(off)
    const float ln2_hi = 6.9313812256e-01f, ln2_lo = 9.0580006145e-06f;
    const float Lp1 = 6.6666668653e-01f, Lp2 = 4.0000000596e-01f, Lp3 = 2.8571429849e-01f, Lp4 = 2.2222198546e-01f,
                Lp5 = 1.8183572590e-01f, Lp6 = 1.5313838422e-01f, Lp7 = 1.4798198640e-01f;
    float hfsq, f = 0.f, c = 0.f, sq, z, R, u;
    int32_t k, hx, hu = 0, ax;
    hx = (int32_t)bits_from_f(x);
    ax = hx & 0x7fffffff;
    k = 1;
    if (hx < 0x3ed413d7) {                                     // x < 0.41422
        if (ax >= 0x3f800000) return x == -1.0f ? -INFINITY : NAN;
        if (ax < 0x31000000) return ax < 0x24800000 ? x : x - x * x * 0.5f;
        if (hx > 0 || hx <= (int32_t)0xbe95f61f) { k = 0; f = x; hu = 1; }
    }
    if (hx >= 0x7f800000) return x + x;
    if (k != 0) {
        if (hx < 0x5a000000) {
            u = 1.0f + x;
            hu = (int32_t)bits_from_f(u);
            k = (hu >> 23) - 127;
            c = (k > 0) ? 1.0f - (u - x) : x - (u - 1.0f);
            c /= u;
        } else {
            u = x;
            hu = (int32_t)bits_from_f(u);
            k = (hu >> 23) - 127;
            c = 0.f;
        }
        hu &= 0x007fffff;
        if (hu < 0x3504f7) {
            u = f_from_bits((uint32_t)hu | 0x3f800000u);
        } else {
            k += 1;
            u = f_from_bits((uint32_t)hu | 0x3f000000u);
            hu = (0x00800000 - hu) >> 2;
        }
        f = u - 1.0f;
    }
    hfsq = 0.5f * f * f;
    if (hu == 0) {
        if (f == 0.0f) {
            if (k == 0) return 0.0f;
            c += k * ln2_lo;
            return k * ln2_hi + c;
        }
        R = hfsq * (1.0f - 0.66666666666666666f * f);
        if (k == 0) return f - R;
        return k * ln2_hi - ((R - (k * ln2_lo + c)) - f);
    }
    sq = f / (2.0f + f);
    z = sq * sq;
    R = z * (Lp1 + z * (Lp2 + z * (Lp3 + z * (Lp4 + z * (Lp5 + z * (Lp6 + z * Lp7))))));
    if (k == 0) return f - (hfsq - sq * (hfsq + R));
    return k * ln2_hi - ((hfsq - (sq * (hfsq + R) + (k * ln2_lo + c))) - f);
}

// vectorised: the element belongs to a whole vector chunk of the operand; otherwise the scalar lambda
__device__ float log_add_exp_aten(float a, float b, bool vectorised) {
#pragma clang fp contract(off)
    if (isinf(a) && a == b) return a;
    const float m = fmaxf(a, b);
    const float d = -fabsf(a - b);
    return vectorised ? m + sleef_log1pf_u10(sleef_expf_u10(d)) : m + fdlibm_log1pf(glibc_expf(d));
}

__global__ void logaddexp_probe_kernel(const float *a, const float *b, float *o, long n, int chunk) {
    const long nvec = chunk > 0 ? n / chunk * chunk : 0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        o[i] = log_add_exp_aten(a[i], b[i], i < nvec);
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
                    l = logd ? log_add_exp_aten(l, el + bq, false) : l + el * bq;       // a 0-dim call: scalar lambda
                }
            }
            label_new[s] = l;
        }
        __syncthreads();
        const int ncand = nb * (1 + V);
        const int nvec = p.vec_chunk > 0 ? ncand / p.vec_chunk * p.vec_chunk : 0;       // handled by ATen's vector loop
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
            cand[i] = logd ? log_add_exp_aten(b, l, i < nvec) : b + l;
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

int halo_set_beam_vector_chunk(int elements) {
    if (elements != 0 && elements != 16 && elements != 32) return HALO_EINVAL;
    halo_ctx_cur().beam_vec_chunk = elements;
    return HALO_OK;
}

int halo_logaddexp_aten(const float *a, const float *b, float *out, size_t n, halo_stream_t stream) {
    HALO_CHECK_ARG(a && b && out);
    if (n == 0) return HALO_OK;
    size_t g = (n + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(logaddexp_probe_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, a, b, out, (long)n, halo_ctx_cur().beam_vec_chunk);
    return halo_launch_status();
}

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
    a.vec_chunk = halo_ctx_cur().beam_vec_chunk;
    hipLaunchKernelGGL(beam_kernel, dim3(N), dim3(256), shmem, (hipStream_t)stream, a);
    return halo_launch_status();
}

}  // extern "C"
