// GPT forward kernels for gfx950 (ha/attention.py: GPT.forward_all, Block, MonitoredSelfAttention,
// LayerNorm): embedding gather, LayerNorm, causal self-attention (flash-style, exact-f32 MFMA with
// online softmax), per-token cross-entropy.  The Linear layers run on the GEMMs of gemm_*.hip
// (bias / tanh-GELU / residual-accumulate epilogues).
#include "halo_common.h"

namespace {

// x[n,:] = wte[ids[n],:] + wpe[pos0 + n % T,:]
__global__ __launch_bounds__(256) void embed_kernel(const int64_t *__restrict__ ids, const float *__restrict__ wte,
                                                    const float *__restrict__ wpe, float *__restrict__ x, int n_tok,
                                                    int T, int C, int pos0, int vocab) {
    const int n = blockIdx.x;
    long id = ids[n];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const float *te = wte + id * C, *pe = wpe + (long)(pos0 + n % T) * C;
    float *o = x + (long)n * C;
    for (int c = threadIdx.x; c < C; c += 256) o[c] = te[c] + pe[c];
}

// F.layer_norm over the last dimension, one wave per row (biased variance, eps inside the sqrt)
__global__ __launch_bounds__(256) void layernorm_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                        const float *__restrict__ b, float *__restrict__ y, int rows,
                                                        int C, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float *xr = x + (long)row * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += xr[c];
    const float mean = wave_sum(s) / (float)C;
    float v = 0.f;
    for (int c = lane; c < C; c += 64) { const float d = xr[c] - mean; v += d * d; }
    const float rstd = rsqrtf(wave_sum(v) / (float)C + eps);
    float *yr = y + (long)row * C;
    for (int c = lane; c < C; c += 64) {
        float o = (xr[c] - mean) * rstd * w[c];
        if (b) o += b[c];
        yr[c] = o;
    }
}

// loss[n] = logsumexp(logits[n,:]) - logits[n,target[n]]   (0 where target == ignore_index)
__global__ __launch_bounds__(256) void cross_entropy_kernel(const float *__restrict__ logits, const int64_t *__restrict__ target,
                                                            float *__restrict__ loss, int V, long ld, long ignore_index) {
    __shared__ float red[4];
    const int n = blockIdx.x;
    const long tgt = target[n];
    if (tgt == ignore_index) { if (threadIdx.x == 0) loss[n] = 0.f; return; }
    const float *row = logits + (long)n * ld;
    float m = -INFINITY;
    for (int c = threadIdx.x; c < V; c += 256) m = fmaxf(m, row[c]);
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int c = threadIdx.x; c < V; c += 256) s += expf(row[c] - m);
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) loss[n] = m + logf((red[0] + red[1]) + (red[2] + red[3])) - row[tgt];
}

// ---- causal self-attention ----------------------------------------------------------------------
// One workgroup = 64 query rows of one (batch, head); 4 waves x 16 rows.  Keys/values stream through
// LDS in tiles of 64; S = Q K^T and O += P V run on v_mfma_f32_16x16x4_f32 (exact f32), softmax is the
// online (running max / running sum) form.  LDS images are chosen for conflict-free MFMA operand reads:
//   Ks [key][dim]  stride 66  (B operand of Q K^T: lane (key = l&15, dim-group = l>>4); 2*key + group is
//                              distinct over a 32-lane read group, and staging writes are 8-byte aligned)
//   Vs [key][dim]  stride 80  (B operand of P V  : lane (dim = l&15, key-group = l>>4))
//   Ps [row][key]  stride 66, per wave (P re-laid from the MFMA D layout to the A layout)
template <int HD>
__global__ __launch_bounds__(256) void attention_causal_kernel(const float *__restrict__ qkv, float *__restrict__ y, int T,
                                                               int n_head, int C, float scale) {
    constexpr int KS_STRIDE = HD + 2, VS_STRIDE = HD + 16, PS_STRIDE = 66;
    __shared__ __attribute__((aligned(16))) float Ks[64 * KS_STRIDE];
    __shared__ __attribute__((aligned(16))) float Vs[64 * VS_STRIDE];
    __shared__ float Ps[4][16 * PS_STRIDE];
    const int qt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int lr = lane & 15, lq = lane >> 4;
    const long row_stride = 3L * C;
    const float *base = qkv + (long)b * T * row_stride + (long)h * HD;
    const int q0 = qt * 64 + wave * 16;                   // this wave's first query row

    // Q fragments, pre-scaled: A[row = lr][k = 4s + lq]
    float qa[HD / 4];
    {
        const int qrow = min(q0 + lr, T - 1);
        const float *qp = base + (long)qrow * row_stride;
#pragma unroll
        for (int s = 0; s < HD / 4; ++s) qa[s] = qp[4 * s + lq] * scale;
    }
    f32x4 o[HD / 16];
#pragma unroll
    for (int m = 0; m < HD / 16; ++m) o[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    float mrow[4], lrow[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { mrow[r] = -INFINITY; lrow[r] = 0.f; }

    const int n_ktiles = qt + 1;                          // causal: keys up to the diagonal tile
    // K/V tiles are fetched one tile ahead into registers (issue early, write to LDS late), so the global
    // latency of tile kt+1 hides under the MFMAs of tile kt
    constexpr int UNITS = 64 * (HD / 4) / 256;            // float4 units of K (and of V) per thread and tile
    f32x4 kreg[UNITS], vreg[UNITS];
    auto fetch = [&](int kt) {
#pragma unroll
        for (int i = 0; i < UNITS; ++i) {
            const int u = threadIdx.x + 256 * i;
            const int key = u / (HD / 4), d4 = (u % (HD / 4)) * 4;
            const int krow = min(kt * 64 + key, T - 1);
            const float *kp = base + (long)krow * row_stride + C + d4;
            kreg[i] = *reinterpret_cast<const f32x4 *>(kp);
            vreg[i] = *reinterpret_cast<const f32x4 *>(kp + C);
        }
    };
    fetch(0);
    for (int kt = 0; kt < n_ktiles; ++kt) {
        __syncthreads();                                  // previous tile fully consumed
#pragma unroll
        for (int i = 0; i < UNITS; ++i) {
            const int u = threadIdx.x + 256 * i;
            const int key = u / (HD / 4), d4 = (u % (HD / 4)) * 4;
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            *reinterpret_cast<f32x2 *>(&Ks[key * KS_STRIDE + d4]) = f32x2{kreg[i][0], kreg[i][1]};
            *reinterpret_cast<f32x2 *>(&Ks[key * KS_STRIDE + d4 + 2]) = f32x2{kreg[i][2], kreg[i][3]};
            *reinterpret_cast<f32x4 *>(&Vs[key * VS_STRIDE + d4]) = vreg[i];
        }
        __syncthreads();
        if (kt + 1 < n_ktiles) fetch(kt + 1);
        // S = Q K^T : 4 key sub-tiles of 16
        f32x4 sacc[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            sacc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < HD / 4; ++s)
                sacc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[s], Ks[(16 * n + lr) * KS_STRIDE + 4 * s + lq], sacc[n], 0, 0, 0);
        }
        // causal mask + online softmax; element (row = 4*lq + r, key = kt*64 + 16n + lr)
        float alpha[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qrow = q0 + 4 * lq + r;
            float mx = -INFINITY;
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int key = kt * 64 + 16 * n + lr;
                if (key > qrow || key >= T) sacc[n][r] = -INFINITY;
                mx = fmaxf(mx, sacc[n][r]);
            }
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
            const float mnew = fmaxf(mrow[r], mx);
            const float msafe = mnew == -INFINITY ? 0.f : mnew;
            alpha[r] = __expf(mrow[r] - msafe);           // exp(-inf) = 0 on the first tile (hardware exp2: ~1e-7 rel.)
            float ps = 0.f;
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const float pv = __expf(sacc[n][r] - msafe);
                sacc[n][r] = pv;
                ps += pv;
            }
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) ps += __shfl_xor(ps, off, 64);
            lrow[r] = lrow[r] * alpha[r] + ps;
            mrow[r] = mnew;
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
    // y[b, row, h*HD + dim] = O / l
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int qrow = q0 + 4 * lq + r;
        if (qrow >= T) continue;
        const float inv = 1.0f / lrow[r];
        float *yp = y + ((long)b * T + qrow) * C + (long)h * HD;
#pragma unroll
        for (int m = 0; m < HD / 16; ++m) yp[16 * m + lr] = o[m][r] * inv;
    }
}

}  // namespace

extern "C" {

int halo_embed_fwd(const int64_t *ids, const float *wte, const float *wpe, float *x, int n_tokens, int T, int C, int pos0,
                   int vocab, halo_stream_t stream) {
    HALO_CHECK_ARG(ids && wte && wpe && x && n_tokens > 0 && T > 0 && C > 0 && pos0 >= 0 && vocab > 0);
    hipLaunchKernelGGL(embed_kernel, dim3(n_tokens), dim3(256), 0, (hipStream_t)stream, ids, wte, wpe, x, n_tokens, T, C, pos0,
                       vocab);
    return halo_launch_status();
}

int halo_layernorm_fwd(const float *x, const float *weight, const float *bias, float *y, int rows, int C, float eps,
                       halo_stream_t stream) {
    HALO_CHECK_ARG(x && weight && y && rows > 0 && C > 0);
    hipLaunchKernelGGL(layernorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, weight, bias, y, rows, C,
                       eps);
    return halo_launch_status();
}

int halo_cross_entropy_fwd(const float *logits, const int64_t *targets, float *loss, int rows, int V, long ld,
                           long ignore_index, halo_stream_t stream) {
    HALO_CHECK_ARG(logits && targets && loss && rows > 0 && V > 0 && ld >= V);
    hipLaunchKernelGGL(cross_entropy_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, logits, targets, loss, V, ld,
                       ignore_index);
    return halo_launch_status();
}

int halo_attention_causal_fwd(const float *qkv, float *y, int B, int T, int n_head, int C, halo_stream_t stream) {
    HALO_CHECK_ARG(qkv && y && B > 0 && T > 0 && n_head > 0 && C > 0 && C % n_head == 0);
    HALO_CHECK_ARG((uintptr_t)qkv % 16 == 0 && C % 4 == 0);
    const int hd = C / n_head;
    const float scale = 1.0f / sqrtf((float)hd);
    dim3 grid((T + 63) / 64, n_head, B);
    if (hd == 64) hipLaunchKernelGGL(attention_causal_kernel<64>, grid, dim3(256), 0, (hipStream_t)stream, qkv, y, T, n_head, C, scale);
    else if (hd == 32) hipLaunchKernelGGL(attention_causal_kernel<32>, grid, dim3(256), 0, (hipStream_t)stream, qkv, y, T, n_head, C, scale);
    else return HALO_ENOTSUP;
    return halo_launch_status();
}

}  // extern "C"
