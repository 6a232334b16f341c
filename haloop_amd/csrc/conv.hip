// Convolutional front-end of the attention ASR encoder (ha/conv.py:25-47 ConvEncoder / DWConv1d) for gfx950,
// channels-last: activations stay [N, T, C] (the layout the collator produces and the transformer blocks
// consume), so the reference's two .mT transposes (ha/transformer.py:235-237) disappear.
//   dense conv      = im2col_cl + GEMM (bias + exact-GELU epilogue)
//   depthwise conv  = dwconv1d_cl (streaming, one thread per 4 channels)
//   pointwise conv  = GEMM on the [N*T', C] rows (bias + exact-GELU epilogue)
#include "halo_common.h"

namespace {

// col[(n, t'), cin*ks + k] = x[n, t'*stride + k - pad, cin]  (0 outside [0, T)); matches weight[Cout, Cin, ks] flattened
__global__ __launch_bounds__(256) void im2col_cl_kernel(const float *__restrict__ x, float *__restrict__ col, int N, int T, int Cin,
                                                        int To, int ks, int stride, int pad) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long K = (long)Cin * ks;
    if (idx >= (long)N * To * K) return;
    const int kk = (int)(idx % K);
    const long row = idx / K;
    const int to = (int)(row % To), n = (int)(row / To);
    const int cin = kk / ks, k = kk % ks;
    const int t = to * stride + k - pad;
    col[idx] = (t >= 0 && t < T) ? x[((long)n * T + t) * Cin + cin] : 0.f;
}

// y[n, t', c] = bias[c] + sum_k w[c, k] * x[n, t'*stride + k - pad, c]
__global__ __launch_bounds__(256) void dwconv1d_cl_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                          const float *__restrict__ bias, float *__restrict__ y, int N, int T, int C,
                                                          int To, int ks, int stride, int pad) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)N * To * C) return;
    const int c = (int)(idx % C);
    const long row = idx / C;
    const int to = (int)(row % To), n = (int)(row / To);
    float acc = 0.f;
    for (int k = 0; k < ks; ++k) {
        const int t = to * stride + k - pad;
        if (t >= 0 && t < T) acc = fmaf(w[c * ks + k], x[((long)n * T + t) * C + c], acc);
    }
    y[idx] = acc + (bias ? bias[c] : 0.f);
}

}  // namespace

extern "C" {

int halo_im2col_cl(const float *x, float *col, int N, int T, int Cin, int ks, int stride, int pad, halo_stream_t stream) {
    HALO_CHECK_ARG(x && col && N > 0 && T > 0 && Cin > 0 && ks > 0 && stride > 0 && pad >= 0);
    const int To = (T + 2 * pad - ks) / stride + 1;
    HALO_CHECK_ARG(To > 0);
    const long n = (long)N * To * Cin * ks;
    hipLaunchKernelGGL(im2col_cl_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, col, N, T, Cin, To, ks,
                       stride, pad);
    return halo_launch_status();
}

int halo_dwconv1d_cl(const float *x, const float *weight, const float *bias, float *y, int N, int T, int C, int ks, int stride,
                     int pad, halo_stream_t stream) {
    HALO_CHECK_ARG(x && weight && y && N > 0 && T > 0 && C > 0 && ks > 0 && stride > 0 && pad >= 0);
    const int To = (T + 2 * pad - ks) / stride + 1;
    HALO_CHECK_ARG(To > 0);
    const long n = (long)N * To * C;
    hipLaunchKernelGGL(dwconv1d_cl_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, weight, bias, y, N,
                       T, C, To, ks, stride, pad);
    return halo_launch_status();
}

}  // extern "C"
