// Small memory-bound kernels around the GEMMs: dropout, im2col for the stride-4 subsample conv,
// relu/dropout backward, row-wise log-softmax, column sums, transpose.
#include <stdlib.h>
#include "halo_common.h"
#include "halo_internal.h"

namespace {

__global__ __launch_bounds__(256) void dropout_fwd_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                          size_t n4, size_t n, DropoutCfg d) {
    // 4 elements per thread: one Philox call covers exactly one float4
    for (size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x; q < n4; q += (size_t)gridDim.x * blockDim.x) {
        const size_t e = q * 4;
        const f32x4 m = dropout_mult4(d, e);
        if (e + 3 < n) {
            f32x4 v = *reinterpret_cast<const f32x4 *>(x + e);
            v[0] *= m[0]; v[1] *= m[1]; v[2] *= m[2]; v[3] *= m[3];
            *reinterpret_cast<f32x4 *>(y + e) = v;
        } else {
            for (int i = 0; i < 4 && e + i < n; ++i) y[e + i] = x[e + i] * m[i];
        }
    }
}

// col[(t*B + b)][c*ks + kk] = x[b][t*stride - pad + kk][c]
__global__ __launch_bounds__(256) void im2col_kernel(const float *__restrict__ x, float *__restrict__ col, int B,
                                                     int T, int F, int Tp, int ks, int stride, int pad) {
    const int row = blockIdx.x;               // t*B + b
    const int t = row / B, b = row % B;
    const int K = F * ks;
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        const int c = k / ks, kk = k % ks;
        const int tt = t * stride - pad + kk;
        float v = 0.f;
        if (tt >= 0 && tt < T) v = x[((long)b * T + tt) * F + c];
        col[(long)row * K + k] = v;
    }
}

// The whole subsample convolution of the forward in ONE launch: a workgroup builds the im2col tile of 16 output rows in LDS (and, once per
// row tile, writes it to col for the backward), then its four waves each multiply it with 16 output channels of the weight on the
// exact-f32 MFMA (16x16x4; a lane's 16-byte load of W and 16-byte read of the tile feed four MFMAs, the k order permuted the same way
// on both sides) and apply bias, relu and the inverted dropout.  Replaces im2col + product + split-K reduce (17 us at B=64, T=80, F=80,
// C=128) for K = F*ks a multiple of 16 and C a multiple of 64.  grid (row tiles, C / 64).
template <int NS>        // NS = K / 16 sixteen-deep k groups (25 at F = 80, ks = 5)
__global__ __launch_bounds__(256) void subsample_fused_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                              const float *__restrict__ bias, float *__restrict__ y,
                                                              float *__restrict__ col, int B, int T, int F, int C, int Tp, int ks, int stride,
                                                              int pad, DropoutCfg drop) {
    extern __shared__ __attribute__((aligned(16))) float tile[];      // [16][LDA], LDA = K + 4 = 4 * odd: 16 rows x 16 bytes land on 16 bank groups
    constexpr int K = 16 * NS, LDA = K + 4;
    const int rows = Tp * B, row0 = blockIdx.x * 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 15, lq = lane >> 4;
    // this lane's weight fragments first: they are in flight while the tile is built
    const int n = blockIdx.y * 64 + wave * 16 + lr;
    const float *wr = w + (long)n * K + 4 * lq;
    f32x4 bw[NS];
#pragma unroll
    for (int u = 0; u < NS; ++u) bw[u] = *reinterpret_cast<const f32x4 *>(wr + 16 * u);
    // im2col tile: unit = (row r, frame kk, four channels c4 .. c4+3), one 16-byte load each (F % 4 == 0), scattered into the
    // col order k = c*ks + kk; all of a thread's loads are issued before the first is used
    const int F4 = F / 4, units = 16 * ks * F4;
    constexpr int MAXU = 8;                                         // 16 * 5 * 20 / 256 = 6.25 at the LC shape
    f32x4 xv[MAXU];
#pragma unroll
    for (int i = 0; i < MAXU; ++i) {
        const int u = threadIdx.x + 256 * i;
        xv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (u < units) {
            const int r = u / (ks * F4), rem = u - r * (ks * F4), kk = rem / F4, c4 = (rem - kk * F4) * 4, row = row0 + r;
            if (row < rows) {
                const int t = row / B, b = row - t * B, tt = t * stride - pad + kk;
                if (tt >= 0 && tt < T) xv[i] = *reinterpret_cast<const f32x4 *>(x + ((long)b * T + tt) * F + c4);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < MAXU; ++i) {
        const int u = threadIdx.x + 256 * i;
        if (u < units) {
            const int r = u / (ks * F4), rem = u - r * (ks * F4), kk = rem / F4, c4 = (rem - kk * F4) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) tile[r * LDA + (c4 + j) * ks + kk] = xv[i][j];
        }
    }
    __syncthreads();
    if (blockIdx.y == 0) {              // the tile, row by row, to col (16-byte stores; K % 16 == 0)
        for (int u = threadIdx.x; u < 16 * (K / 4); u += 256) {
            const int r = u / (K / 4), k4 = (u - r * (K / 4)) * 4;
            if (row0 + r < rows) *reinterpret_cast<f32x4 *>(col + (long)(row0 + r) * K + k4) = *reinterpret_cast<const f32x4 *>(tile + r * LDA + k4);
        }
    }
    const float *ar = tile + lr * LDA + 4 * lq;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < NS; ++u) {
        const f32x4 a = *reinterpret_cast<const f32x4 *>(ar + 16 * u);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], bw[u][j], acc, 0, 0, 0);
    }
    const int colc = blockIdx.y * 64 + wave * 16 + lr;             // D layout: column = lane % 16, rows 4 * (lane / 16) + e
    const float bv = bias ? bias[colc] : 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int row = row0 + 4 * lq + e;
        if (row >= rows) continue;
        const long o = (long)row * C + colc;
        float v = fmaxf(acc[e] + bv, 0.f);
        if (drop.threshold) v *= dropout_mult(drop, (uint64_t)o);
        y[o] = v;
    }
}

// dpre = dy * (y > 0 ? scale : 0): backward of relu followed by inverted dropout, read off y
__global__ __launch_bounds__(256) void relu_dropout_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ y,
                                                               float *__restrict__ dpre, size_t n, float scale) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dpre[i] = y[i] > 0.f ? dy[i] * scale : 0.f;
}

// dy slices summed in place into slice 0 (the unfused fallback of halo_subsample_bwd_slabs)
__global__ __launch_bounds__(256) void sum_slabs_kernel(float *__restrict__ dy, size_t n, int slabs, long slab_stride) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float g = dy[i];
        for (int q = 1; q < slabs; ++q) g += dy[i + q * slab_stride];
        dy[i] = g;
    }
}

// The front-end convolution's backward in two launches (relu/dropout mask + weight-gradient product + bias column sums, then the
// reduction of the row chunks) instead of four (mask, exact-f32 GEMM, split-K reduce, column sum).  Workgroup (chunk, c-tile, n-half):
// rows [chunk * CH, +CH) of dy / y / col, 16 output channels, half of the K = F * ks columns.  dpre = dy * (y > 0 ? scale : 0) of its
// rows x 16 channels goes to LDS (it is the A operand, read transposed: A[m = channel][k = row]); the B fragments B[k = row][n] are
// read straight from col (4 rows x 16 columns = 4 x 64 contiguous bytes per wave instruction); fp32 MFMA 16x16x4, rows in order.
// part[chunk][c][k] and bias_part[chunk][c] are summed over the chunks, in order, by subsample_bwd_reduce_kernel.
// slabs > 1: dy is the sum of that many [rows][C] matrices slab_stride floats apart (halo_subsample_bwd_slabs).
__global__ __launch_bounds__(256) void subsample_bwd_partial_kernel(const float *__restrict__ dy, const float *__restrict__ y,
                                                                    const float *__restrict__ col, float *__restrict__ part,
                                                                    float *__restrict__ bias_part, int rows, int C, int K, int CH, float scale,
                                                                    int slabs, long slab_stride) {
    extern __shared__ float dp[];                      // [CH rounded up to 32][17], zero beyond the chunk's rows
    const int chunk = blockIdx.x, c0 = blockIdx.y * 16, nh = blockIdx.z;
    const int row0 = chunk * CH, nrows = min(CH, rows - row0), CHP = (CH + 31) / 32 * 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 15, lq = lane >> 4;
    for (int u = threadIdx.x; u < CHP * 16; u += 256) {
        const int r = u >> 4, c = u & 15;
        float v = 0.f;
        if (r < nrows) {
            const long o = (long)(row0 + r) * C + c0 + c;
            // dy left as K-slices by its producer: added in order, eight independent loads at a time
            float g = dy[o];
            for (int q0 = 1; q0 < slabs; q0 += 8) {
                float gq[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) gq[q] = dy[o + min(q0 + q, slabs - 1) * slab_stride];
#pragma unroll
                for (int q = 0; q < 8; ++q) g += q0 + q < slabs ? gq[q] : 0.f;
            }
            v = y[o] > 0.f ? g * scale : 0.f;
        }
        dp[r * 17 + c] = v;
    }
    __syncthreads();
    if (nh == 0 && threadIdx.x < 16) {
        float sum = 0.f;
        for (int r = 0; r < nrows; ++r) sum += dp[r * 17 + threadIdx.x];
        bias_part[(long)chunk * C + c0 + threadIdx.x] = sum;
    }
    const int ntiles = K / 16, half = (ntiles + 1) / 2;
    const int t0 = nh == 0 ? 0 : half, t1 = nh == 0 ? half : ntiles;
    // this wave's n-tiles: t0 + wave, + 4, ... (at most 4 of them: K / 16 <= 32); a tile past the half is computed on the last valid
    // tile's columns and not stored, rows past the chunk read the last row against a zero A -- so that every load is unconditional
    // and the 32 loads of eight k-steps are in flight together
    constexpr int MAXT = 4, KU = 8;
    f32x4 acc[MAXT];
    const float *cb[MAXT];
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
        acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        cb[i] = col + (long)row0 * K + min(t0 + wave + 4 * i, t1 - 1) * 16 + lr;
    }
    for (int k0 = 0; k0 < CHP; k0 += 4 * KU) {
        float a[KU], b[KU][MAXT];
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const int r = k0 + 4 * u + lq;
            a[u] = dp[r * 17 + lr];
            const long ro = (long)min(r, nrows - 1) * K;
#pragma unroll
            for (int i = 0; i < MAXT; ++i) b[u][i] = cb[i][ro];
        }
#pragma unroll
        for (int u = 0; u < KU; ++u)
#pragma unroll
            for (int i = 0; i < MAXT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u][i], acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
        const int t = t0 + wave + 4 * i;
        if (t >= t1) continue;
#pragma unroll
        for (int e = 0; e < 4; ++e) part[((long)chunk * C + c0 + 4 * lq + e) * K + t * 16 + lr] = acc[i][e];    // D: row 4 lq + e, column lr
    }
}

// ... and, from the blocks behind its own (own_blocks of them), the small reductions queued on the context (small_jobs.h)
__global__ __launch_bounds__(256) void subsample_bwd_reduce_kernel(const float *__restrict__ part, const float *__restrict__ bias_part,
                                                                   float *__restrict__ dw, float *__restrict__ dbias, int chunks, long CK, int C,
                                                                   int own_blocks, const HaloSmallJobs jobs, float *__restrict__ ss) {
    __shared__ float red[4][64];
    if ((int)blockIdx.x >= own_blocks) {
        halo_small_jobs_block(jobs, blockIdx.x - own_blocks, red);
        return;
    }
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    float sum = 0.f;
    if (i < CK) {
        for (int q = 0; q < chunks; ++q) sum += part[(long)q * CK + i];
        dw[i] = sum;
    } else if (i - CK < C) {
        for (int q = 0; q < chunks; ++q) sum += bias_part[(long)q * C + (i - CK)];
        dbias[i - CK] = sum;
    }
    if (ss) {           // the squared-norm partial of this block's 256 gradient elements (halo_set_grad_sumsq)
        const float q = wave_sum(sum * sum);
        if ((threadIdx.x & 63) == 0) red[0][threadIdx.x >> 6] = q;
        __syncthreads();
        if (threadIdx.x == 0) ss[blockIdx.x] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    }
}

__global__ __launch_bounds__(256) void small_jobs_kernel(const HaloSmallJobs jobs) {
    __shared__ float red[4][64];
    halo_small_jobs_block(jobs, blockIdx.x, red);
}

// one wave per row
__global__ __launch_bounds__(256) void log_softmax_fwd_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                              int rows, int cols) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float *xr = x + (long)row * cols;
    float m = -INFINITY;
    for (int c = lane; c < cols; c += 64) m = fmaxf(m, xr[c]);
    m = wave_max(m);
    float s = 0.f;
    for (int c = lane; c < cols; c += 64) s += expf(xr[c] - m);
    s = wave_sum(s);
    const float lse = m + logf(s);
    float *yr = y + (long)row * cols;
    for (int c = lane; c < cols; c += 64) yr[c] = xr[c] - lse;
}

// dx = dy - exp(y) * sum(dy)
__global__ __launch_bounds__(256) void log_softmax_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ y,
                                                              float *__restrict__ dx, int rows, int cols) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float *dr = dy + (long)row * cols, *yr = y + (long)row * cols;
    float s = 0.f;
    for (int c = lane; c < cols; c += 64) s += dr[c];
    s = wave_sum(s);
    float *xr = dx + (long)row * cols;
    for (int c = lane; c < cols; c += 64) xr[c] = dr[c] - expf(yr[c]) * s;
}

// out[n] = sum_m x[m*ld + n]; block = 32 columns x 32 row slices, fixed summation order
__global__ __launch_bounds__(1024) void colsum_kernel(const float *__restrict__ x, int rows, int cols, int ld,
                                                      float *__restrict__ out, float *__restrict__ out2) {
    __shared__ float part[32][33];
    const int lane_c = threadIdx.x & 31;
    const int c = blockIdx.x * 32 + lane_c;
    const int slice = threadIdx.x >> 5;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < cols) {
        int r = slice;
        for (; r + 96 < rows; r += 128) {
            s0 += x[(long)r * ld + c];
            s1 += x[(long)(r + 32) * ld + c];
            s2 += x[(long)(r + 64) * ld + c];
            s3 += x[(long)(r + 96) * ld + c];
        }
        for (; r < rows; r += 32) s0 += x[(long)r * ld + c];
    }
    part[slice][lane_c] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (slice == 0 && c < cols) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) t += part[i][lane_c];
        out[c] = t;
        if (out2) out2[c] = t;
    }
}

// out[c][r] = in[r][c]
__global__ __launch_bounds__(256) void transpose_kernel(const float *__restrict__ in, float *__restrict__ out, int rows,
                                                        int cols) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8)
        if (r0 + i < rows && c0 + tx < cols) tile[i][tx] = in[(long)(r0 + i) * cols + c0 + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (c0 + i < cols && r0 + tx < rows) out[(long)(c0 + i) * rows + r0 + tx] = tile[tx][i];
}

__global__ void counter_inc_kernel(uint32_t *c) { *c += 1u; }

// subsampled lengths (float floor like ha/rnn.py:13-18) and the per-utterance weights of the CTC mean
__global__ __launch_bounds__(256) void ctc_prepare_kernel(const int64_t *__restrict__ il, const int64_t *__restrict__ tl,
                                                          int n, int ks, int stride, int pad, int64_t *__restrict__ flen,
                                                          float *__restrict__ grad_out) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float o = (float)(il[i] + 2 * pad - ks);
        flen[i] = (int64_t)floorf(o / (float)stride + 1.0f);
        const float t = fmaxf((float)tl[i], 1.0f);
        grad_out[i] = 1.0f / (t * (float)n);
    }
}

// loss = mean_n(nll[n] / max(tl[n], 1)), summed in a fixed order
__global__ __launch_bounds__(256) void ctc_mean_loss_kernel(const float *__restrict__ nll, const int64_t *__restrict__ tl,
                                                            int n, float *__restrict__ loss) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += nll[i] / fmaxf((float)tl[i], 1.0f);
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) *loss = ((red[0] + red[1]) + (red[2] + red[3])) / (float)n;
}

__global__ __launch_bounds__(256) void fill_kernel(float *p, size_t n, float v) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

inline unsigned grid_for(size_t n, unsigned per_block = 256, unsigned cap = 4096) {
    size_t g = (n + per_block - 1) / per_block;
    if (g < 1) g = 1;
    return (unsigned)(g > cap ? cap : g);
}

}  // namespace

// ---- internal helpers used by lstm.hip --------------------------------------------------------
int halo_transpose(const float *in, float *out, int rows, int cols, hipStream_t st) {
    hipLaunchKernelGGL(transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, st, in, out, rows, cols);
    return halo_launch_status();
}
int halo_colsum2(const float *x, int rows, int cols, int ld, float *out, float *out2, hipStream_t st) {
    hipLaunchKernelGGL(colsum_kernel, dim3((cols + 31) / 32), dim3(1024), 0, st, x, rows, cols, ld, out, out2);
    return halo_launch_status();
}
int halo_fill(float *p, size_t n, float v, hipStream_t st) {
    if (n == 0) return HALO_OK;
    hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n)), dim3(256), 0, st, p, n, v);
    return halo_launch_status();
}

static inline int subsampled_len(int T, int ks, int stride, int pad) { return (T + 2 * pad - ks) / stride + 1; }

bool halo_defer_small_job(const HaloSmallJob &job) {
    HaloCtx &ctx = halo_ctx_cur();
    if (!ctx.defer_small_jobs || ctx.small_jobs.n >= HALO_SMALL_JOBS_MAX) return false;
    HaloSmallJob &j = ctx.small_jobs.job[ctx.small_jobs.n++];
    j = job;
    j.blocks = halo_small_job_blocks(job);
    ctx.small_jobs.blocks += j.blocks;
    return true;
}

extern "C" {

int halo_dropout_fwd(const float *x, float *y, size_t n, float p, uint64_t seed, uint32_t stream_id, uint32_t offset,
                     const uint32_t *offset_dev, halo_stream_t stream) {
    HALO_CHECK_ARG(x && y);
    HALO_CHECK_ARG(((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0));
    if (n == 0) return HALO_OK;
    const size_t n4 = (n + 3) / 4;
    hipLaunchKernelGGL(dropout_fwd_kernel, dim3(grid_for(n4)), dim3(256), 0, (hipStream_t)stream, x, y, n4, n,
                       make_dropout(p, seed, stream_id, offset, offset_dev));
    return halo_launch_status();
}

int halo_ctc_prepare(const int64_t *input_lengths, const int64_t *target_lengths, int n, int ks, int stride, int pad,
                     int64_t *feature_lengths, float *grad_out, halo_stream_t stream) {
    HALO_CHECK_ARG(input_lengths && target_lengths && feature_lengths && grad_out && n > 0 && ks > 0 && stride > 0 && pad >= 0);
    hipLaunchKernelGGL(ctc_prepare_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, input_lengths,
                       target_lengths, n, ks, stride, pad, feature_lengths, grad_out);
    return halo_launch_status();
}

int halo_ctc_mean_loss(const float *nll, const int64_t *target_lengths, int n, float *loss, halo_stream_t stream) {
    HALO_CHECK_ARG(nll && target_lengths && loss && n > 0);
    hipLaunchKernelGGL(ctc_mean_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, nll, target_lengths, n, loss);
    return halo_launch_status();
}

int halo_counter_inc(uint32_t *counter, halo_stream_t stream) {
    HALO_CHECK_ARG(counter);
    hipLaunchKernelGGL(counter_inc_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, counter);
    return halo_launch_status();
}

size_t halo_subsample_col_bytes(int B, int T, int F, int ks, int stride, int pad) {
    if (B <= 0 || T <= 0 || F <= 0 || ks <= 0 || stride <= 0 || T + 2 * pad < ks) return 0;
    return (size_t)subsampled_len(T, ks, stride, pad) * B * F * ks * sizeof(float);
}

int halo_subsample_fwd(const float *x, const float *w, const float *bias, float *y, float *col, int B, int T, int F,
                       int C, int ks, int stride, int pad, float p_drop, uint64_t seed, uint32_t offset,
                       const uint32_t *offset_dev, halo_stream_t stream) {
    HALO_CHECK_ARG(x && w && y && col);
    HALO_CHECK_ARG(B > 0 && T > 0 && F > 0 && C > 0 && ks > 0 && stride > 0 && pad >= 0 && T + 2 * pad >= ks);
    const int Tp = subsampled_len(T, ks, stride, pad);
    hipStream_t st = (hipStream_t)stream;
    const int K = F * ks;
    static const int fused = getenv("HALO_SUBSAMPLE_FUSED") ? atoi(getenv("HALO_SUBSAMPLE_FUSED")) : 1;
    // the LC front-end's shape (K = 400): one fused launch; the loads of a workgroup's 16 x ks x F/4 tile units must fit 8 per thread
    if (fused && K == 400 && F % 4 == 0 && C % 64 == 0 && 16 * ks * (F / 4) <= 8 * 256 &&
        (((uintptr_t)w | (uintptr_t)x | (uintptr_t)col) % 16 == 0)) {
        const DropoutCfg d = make_dropout(p_drop, seed, HALO_STREAM_SUBSAMPLE, offset, offset_dev);
        hipLaunchKernelGGL(subsample_fused_kernel<25>, dim3((Tp * B + 15) / 16, C / 64), dim3(256), (size_t)16 * (K + 4) * sizeof(float), st,
                           x, w, bias, y, col, B, T, F, C, Tp, ks, stride, pad, d);
        return halo_launch_status();
    }
    hipLaunchKernelGGL(im2col_kernel, dim3(Tp * B), dim3(256), 0, st, x, col, B, T, F, Tp, ks, stride, pad);
    int rc = halo_launch_status();
    if (rc) return rc;
    // y[T'B, C] = col[T'B, F*ks] * w[C, F*ks]^T + bias, relu, dropout on the time-major index
    return halo_gemm_f32(1, 1, Tp * B, C, F * ks, col, F * ks, w, F * ks, y, C, bias, nullptr, HALO_GEMM_RELU, p_drop,
                         seed, HALO_STREAM_SUBSAMPLE, offset, offset_dev, stream);
}

int halo_subsample_bwd(const float *dy, const float *y, const float *col, float *dpre, float *dw, float *dbias, int B,
                       int T, int F, int C, int ks, int stride, int pad, float p_drop, halo_stream_t stream) {
    return halo_subsample_bwd_slabs((float *)dy, 1, y, col, dpre, dw, dbias, B, T, F, C, ks, stride, pad, p_drop, stream);
}

int halo_subsample_bwd_slabs(float *dy, int slabs, const float *y, const float *col, float *dpre, float *dw, float *dbias, int B,
                             int T, int F, int C, int ks, int stride, int pad, float p_drop, halo_stream_t stream) {
    HALO_CHECK_ARG(dy && y && col && dpre && dw && dbias && slabs >= 1);
    HALO_CHECK_ARG(B > 0 && T > 0 && F > 0 && C > 0 && ks > 0 && stride > 0 && pad >= 0 && T + 2 * pad >= ks);
    const int Tp = subsampled_len(T, ks, stride, pad);
    const size_t n = (size_t)Tp * B * C;
    hipStream_t st = (hipStream_t)stream;
    const float scale = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
    {
        // two launches when the shape fits them and the caller has lent scratch for the row chunks' partial sums
        const int rows = Tp * B, K = F * ks;
        // sixteen row chunks, more when they would be longer than 168 rows (a workgroup builds its chunk's masked gradient tile first: 1.076 ->
        // 1.056 ms per step at B = 256 with 32 chunks of 168 instead of 16 of 336; more chunks at B = 64 cost the reduce more than they save)
        const int nch = rows > 16 * 168 ? (rows + 167) / 168 : 16;
        int CH = ((rows + nch - 1) / nch + 3) / 4 * 4;                 // whole 4-row k-steps
        const int chunks = (rows + CH - 1) / CH;
        void *scratch; size_t bytes;
        halo_get_scratch(&scratch, &bytes);
        static const int fused = getenv("HALO_SUBSAMPLE_BWD_FUSED") ? atoi(getenv("HALO_SUBSAMPLE_BWD_FUSED")) : 1;
        const size_t need = ((size_t)chunks * C * K + (size_t)chunks * C) * sizeof(float);
        if (fused && scratch && bytes >= need && K % 16 == 0 && K / 16 <= 32 && C % 16 == 0 && rows >= 64 && CH <= 512) {
            float *part = (float *)scratch, *bias_part = part + (size_t)chunks * C * K;
            hipLaunchKernelGGL(subsample_bwd_partial_kernel, dim3(chunks, C / 16, 2), dim3(256), (size_t)((CH + 31) / 32 * 32) * 17 * sizeof(float), st, dy, y,
                               col, part, bias_part, rows, C, K, CH, scale, slabs, (long)n);
            int rc = halo_launch_status();
            if (rc) return rc;
            const long CK = (long)C * K;
            const int own = (int)((CK + C + 255) / 256);
            HaloCtx &ctx = halo_ctx_cur();
            HaloSmallJobs &q = ctx.small_jobs;                       // queued small reductions ride in this launch's tail blocks
            float *ss = ctx.grad_sumsq && ctx.grad_sumsq_n + own <= ctx.grad_sumsq_cap ? ctx.grad_sumsq + ctx.grad_sumsq_n : nullptr;
            hipLaunchKernelGGL(subsample_bwd_reduce_kernel, dim3((unsigned)(own + q.blocks)), dim3(256), 0, st, part, bias_part, dw,
                               dbias, chunks, CK, C, own, q, ss);
            if (ss) { ctx.grad_sumsq_n += own; ctx.grad_sumsq_cover |= 16u; }
            q.n = 0; q.blocks = 0;
            return halo_launch_status();
        }
    }
    if (slabs > 1) {
        hipLaunchKernelGGL(sum_slabs_kernel, dim3(grid_for(n)), dim3(256), 0, st, dy, n, slabs, (long)n);
        int rc = halo_launch_status();
        if (rc) return rc;
    }
    hipLaunchKernelGGL(relu_dropout_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, st, dy, y, dpre, n, scale);
    int rc = halo_launch_status();
    if (rc) return rc;
    // dw[C, F*ks] = dpre[T'B, C]^T * col[T'B, F*ks]
    rc = halo_gemm_f32(0, 0, C, F * ks, Tp * B, dpre, C, col, F * ks, dw, F * ks, nullptr, nullptr, 0, 0.f, 0, 0, 0,
                       nullptr, stream);
    if (rc) return rc;
    rc = halo_colsum(dpre, Tp * B, C, C, dbias, stream);
    if (rc) return rc;
    return halo_flush_small_jobs(stream);
}

int halo_set_defer_small_jobs(int on) {
    HaloCtx &ctx = halo_ctx_cur();
    ctx.defer_small_jobs = on ? 1 : 0;
    if (on) { ctx.small_jobs.n = 0; ctx.small_jobs.blocks = 0; }
    return HALO_OK;
}

int halo_flush_small_jobs(halo_stream_t stream) {
    HaloSmallJobs &q = halo_ctx_cur().small_jobs;
    if (q.n == 0) return HALO_OK;
    hipLaunchKernelGGL(small_jobs_kernel, dim3((unsigned)q.blocks), dim3(256), 0, (hipStream_t)stream, q);
    q.n = 0; q.blocks = 0;
    return halo_launch_status();
}

int halo_log_softmax_fwd(const float *x, float *y, int rows, int cols, halo_stream_t stream) {
    HALO_CHECK_ARG(x && y && rows > 0 && cols > 0);
    hipLaunchKernelGGL(log_softmax_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, y, rows, cols);
    return halo_launch_status();
}

int halo_log_softmax_bwd(const float *dy, const float *y, float *dx, int rows, int cols, halo_stream_t stream) {
    HALO_CHECK_ARG(dy && y && dx && rows > 0 && cols > 0);
    hipLaunchKernelGGL(log_softmax_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, dy, y, dx, rows,
                       cols);
    return halo_launch_status();
}

int halo_colsum(const float *x, int rows, int cols, int ld, float *out, halo_stream_t stream) {
    HALO_CHECK_ARG(x && out && rows > 0 && cols > 0 && ld >= cols);
    return halo_colsum2(x, rows, cols, ld, out, nullptr, (hipStream_t)stream);
}

}  // extern "C"
