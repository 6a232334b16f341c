// Convolutional front-end of the attention ASR encoder (ha/conv.py:25-47 ConvEncoder / DWConv1d) for gfx950,
// channels-last: activations stay [N, T, C] (the layout the collator produces and the transformer blocks
// consume), so the reference's two .mT transposes (ha/transformer.py:235-237) disappear.
//   dense conv      = im2col_cl + GEMM (bias + exact-GELU epilogue)
//   depthwise conv  = dwconv1d_cl (streaming, one thread per 4 channels)
//   pointwise conv  = GEMM on the [N*T', C] rows (bias + exact-GELU epilogue)
#include "halo_common.h"
#include "halo_internal.h"

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

// transpose of im2col_cl (the gradient of the unfold): dx[n, t, cin] = sum over (t', k) with t'*stride + k - pad == t of dcol[(n, t'), cin*ks + k]
__global__ __launch_bounds__(256) void col2im_cl_kernel(const float *__restrict__ dcol, float *__restrict__ dx, int N, int T, int Cin,
                                                        int To, int ks, int stride, int pad) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)N * T * Cin) return;
    const int cin = (int)(idx % Cin);
    const long row = idx / Cin;
    const int t = (int)(row % T), n = (int)(row / T);
    const long K = (long)Cin * ks;
    float acc = 0.f;
    for (int k = 0; k < ks; ++k) {
        const int u = t + pad - k;
        if (u < 0 || u % stride) continue;
        const int to = u / stride;
        if (to < To) acc += dcol[((long)n * To + to) * K + (long)cin * ks + k];
    }
    dx[idx] = acc;
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

// dx[n, t, c] = sum_k w[c, k] * dy[n, (t + pad - k) / stride, c]   over the k with (t + pad - k) % stride == 0 and a valid output index
__global__ __launch_bounds__(256) void dwconv1d_cl_bwd_dx_kernel(const float *__restrict__ dy, const float *__restrict__ w,
                                                                 float *__restrict__ dx, int N, int T, int C, int To, int ks, int stride,
                                                                 int pad) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)N * T * C) return;
    const int c = (int)(idx % C);
    const long row = idx / C;
    const int t = (int)(row % T), n = (int)(row / T);
    float acc = 0.f;
    for (int k = 0; k < ks; ++k) {
        const int u = t + pad - k;
        if (u < 0 || u % stride) continue;
        const int to = u / stride;
        if (to < To) acc = fmaf(w[c * ks + k], dy[((long)n * To + to) * C + c], acc);
    }
    dx[idx] = acc;
}

// partial[chunk, c*ks + k] = sum over the chunk's output rows of dy[row, c] * x[n, to*stride + k - pad, c];  pb[chunk, c] = sum dy
__global__ __launch_bounds__(256) void dwconv1d_cl_bwd_dw_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                                 float *__restrict__ pw, float *__restrict__ pb, int N, int T, int C, int To,
                                                                 int ks, int stride, int pad, int rows_per_chunk) {
    const int c = blockIdx.x * 256 + threadIdx.x, chunk = blockIdx.y;
    if (c >= C) return;
    const int rows = N * To, r0 = chunk * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
    float ab = 0.f;
    float aw[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) aw[k] = 0.f;
    for (int r = r0; r < r1; ++r) {
        const int n = r / To, to = r % To;
        const float d = dy[(long)r * C + c];
        ab += d;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int t = to * stride + k - pad;
            if (k < ks && t >= 0 && t < T) aw[k] = fmaf(d, x[((long)n * T + t) * C + c], aw[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (k < ks) pw[(long)chunk * C * ks + c * ks + k] = aw[k];
    pb[(long)chunk * C + c] = ab;
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

int halo_col2im_cl(const float *dcol, float *dx, int N, int T, int Cin, int ks, int stride, int pad, halo_stream_t stream) {
    HALO_CHECK_ARG(dcol && dx && N > 0 && T > 0 && Cin > 0 && ks > 0 && stride > 0 && pad >= 0);
    const int To = (T + 2 * pad - ks) / stride + 1;
    HALO_CHECK_ARG(To > 0);
    const long n = (long)N * T * Cin;
    hipLaunchKernelGGL(col2im_cl_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dcol, dx, N, T, Cin, To, ks,
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

#define HALO_DWCONV_BWD_CHUNKS 64

size_t halo_dwconv1d_cl_bwd_workspace_bytes(int C, int ks) {
    if (C <= 0 || ks <= 0) return 0;
    return (size_t)HALO_DWCONV_BWD_CHUNKS * C * (ks + 1) * sizeof(float);
}

int halo_dwconv1d_cl_bwd(const float *dy, const float *x, const float *weight, float *dx, float *dweight, float *dbias,
                         void *workspace, int N, int T, int C, int ks, int stride, int pad, halo_stream_t stream) {
    HALO_CHECK_ARG(dy && x && weight && dweight && workspace && N > 0 && T > 0 && C > 0 && ks > 0 && ks <= 8 && stride > 0 && pad >= 0);
    const int To = (T + 2 * pad - ks) / stride + 1;
    HALO_CHECK_ARG(To > 0);
    hipStream_t st = (hipStream_t)stream;
    if (dx) {
        const long n = (long)N * T * C;
        hipLaunchKernelGGL(dwconv1d_cl_bwd_dx_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dy, weight, dx, N, T, C, To, ks,
                           stride, pad);
    }
    const int rows = N * To;
    const int chunks = rows < HALO_DWCONV_BWD_CHUNKS ? rows : HALO_DWCONV_BWD_CHUNKS, rpc = (rows + chunks - 1) / chunks;
    const int used = (rows + rpc - 1) / rpc;
    float *pw = (float *)workspace, *pb = pw + (size_t)HALO_DWCONV_BWD_CHUNKS * C * ks;
    hipLaunchKernelGGL(dwconv1d_cl_bwd_dw_kernel, dim3((C + 255) / 256, used), dim3(256), 0, st, dy, x, pw, pb, N, T, C, To, ks, stride,
                       pad, rpc);
    if (halo_launch_status() != HALO_OK) return HALO_ELAUNCH;
    int rc = halo_colsum2(pw, used, C * ks, C * ks, dweight, nullptr, st);
    if (rc == HALO_OK && dbias) rc = halo_colsum2(pb, used, C, C, dbias, nullptr, st);
    return rc;
}

}  // extern "C"
