// GPT forward kernels for gfx950 (ha/attention.py: GPT.forward_all, Block, MonitoredSelfAttention,
// LayerNorm): embedding gather, LayerNorm, per-token cross-entropy (the attention kernels live in attn.hip).  The Linear layers run on the GEMMs of gemm_*.hip
// (bias / tanh-GELU / residual-accumulate epilogues).
#include "halo_common.h"

namespace {

// x[n,:] = wte[ids[n],:] + wpe[pos0 + n % T,:]   (wpe may be NULL: ha/transformer.py:105 has no position table)
__global__ __launch_bounds__(256) void embed_kernel(const int64_t *__restrict__ ids, const float *__restrict__ wte,
                                                    const float *__restrict__ wpe, float *__restrict__ x, int n_tok,
                                                    int T, int C, int pos0, int vocab) {
    const int n = blockIdx.x;
    long id = ids[n];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const float *te = wte + id * C, *pe = wpe ? wpe + (long)(pos0 + n % T) * C : nullptr;
    float *o = x + (long)n * C;
    for (int c = threadIdx.x; c < C; c += 256) o[c] = pe ? te[c] + pe[c] : te[c];
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

}  // namespace

extern "C" {

int halo_embed_fwd(const int64_t *ids, const float *wte, const float *wpe, float *x, int n_tokens, int T, int C, int pos0,
                   int vocab, halo_stream_t stream) {
    HALO_CHECK_ARG(ids && wte && x && n_tokens > 0 && T > 0 && C > 0 && pos0 >= 0 && vocab > 0);
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

}  // extern "C"
