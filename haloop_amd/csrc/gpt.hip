// GPT forward kernels for gfx950 (ha/attention.py: GPT.forward_all, Block, MonitoredSelfAttention,
// LayerNorm): embedding gather, LayerNorm, per-token cross-entropy (the attention kernels live in attn.hip).  The Linear layers run on the GEMMs of gemm_*.hip
// (bias / tanh-GELU / residual-accumulate epilogues).
#include "halo_common.h"
#include "halo_internal.h"

#define HALO_LN_BWD_CHUNKS 512   // row chunks (workgroups) of the LayerNorm weight/bias gradient partial sums

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
// One sweep over the row: every thread keeps a running (max, sum of exp relative to it) over its 16-byte pieces and the
// pairs are merged at the end (online softmax), so the 200 KB row of a 50k vocabulary is read once, not twice.
__device__ __forceinline__ void lse_merge(float &m, float &s, float om, float os) {
    const float nm = fmaxf(m, om);
    if (nm == -INFINITY) return;
    s = s * __expf(m - nm) + os * __expf(om - nm);
    m = nm;
}

__global__ __launch_bounds__(256) void cross_entropy_kernel(const float *__restrict__ logits, const int64_t *__restrict__ target,
                                                            float *__restrict__ loss, float *__restrict__ lse_out, int V, long ld,
                                                            long ignore_index) {
    __shared__ float redm[4], reds[4];
    const int n = blockIdx.x;
    const long tgt = target[n];
    if (tgt == ignore_index) { if (threadIdx.x == 0) { loss[n] = 0.f; if (lse_out) lse_out[n] = 0.f; } return; }
    const float *row = logits + (long)n * ld;
    float m = -INFINITY, s = 0.f;
    const bool vec = (ld % 4 == 0) && ((uintptr_t)logits % 16 == 0);
    const int V4 = vec ? V / 4 : 0;
    for (int c = threadIdx.x; c < V4; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(row + 4 * c);
        const float vm = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
        if (vm > m) { s *= __expf(m - vm); m = vm; }
        s += (__expf(v[0] - m) + __expf(v[1] - m)) + (__expf(v[2] - m) + __expf(v[3] - m));
    }
    for (int c = 4 * V4 + threadIdx.x; c < V; c += 256) {
        const float v = row[c];
        if (v > m) { s *= __expf(m - v); m = v; }
        s += __expf(v - m);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lse_merge(m, s, __shfl_xor(m, o, 64), __shfl_xor(s, o, 64));
    if ((threadIdx.x & 63) == 0) { redm[threadIdx.x >> 6] = m; reds[threadIdx.x >> 6] = s; }
    __syncthreads();
    if (threadIdx.x == 0) {
        m = redm[0]; s = reds[0];
#pragma unroll
        for (int w = 1; w < 4; ++w) lse_merge(m, s, redm[w], reds[w]);
        const float l = m + logf(s);
        loss[n] = l - row[tgt];
        if (lse_out) lse_out[n] = l;
    }
}

// ---- backward kernels of the GPT training path -----------------------------------------------------------
// dx[n,:] = dres[n,:] + rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * w,  xhat = (x - mean) * rstd;
// stats[n] = (mean, rstd) for the column pass.  One wave per row; the row (<= a few KB) is re-read from L1.
__global__ __launch_bounds__(256) void layernorm_bwd_dx_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                               const float *__restrict__ w, const float *__restrict__ dres,
                                                               float *__restrict__ dx, float *__restrict__ stats, int rows, int C,
                                                               float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float *xr = x + (long)row * C, *dyr = dy + (long)row * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += xr[c];
    const float mean = wave_sum(s) / (float)C;
    float v = 0.f;
    for (int c = lane; c < C; c += 64) { const float d = xr[c] - mean; v += d * d; }
    const float rstd = rsqrtf(wave_sum(v) / (float)C + eps);
    float sg = 0.f, sgx = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float g = dyr[c] * w[c];
        sg += g;
        sgx += g * (xr[c] - mean) * rstd;
    }
    const float mg = wave_sum(sg) / (float)C, mgx = wave_sum(sgx) / (float)C;
    float *dxr = dx + (long)row * C;
    const float *rr = dres ? dres + (long)row * C : nullptr;
    for (int c = lane; c < C; c += 64) {
        const float xh = (xr[c] - mean) * rstd;
        const float d = rstd * (dyr[c] * w[c] - mg - xh * mgx);
        dxr[c] = rr ? rr[c] + d : d;
    }
    if (lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
}

// The same backward in ONE pass over dy and x for C % 4 == 0, C <= 256 * MAXV: a wave holds its row in registers (float4 per lane),
// so x and dy are read once, the four row statistics are DPP / permlane reductions, and the weight / bias gradient partial sums
// of the rows a workgroup walks (rows b*4 + wave, stepping by 4 * gridDim) stay in registers until the end: pw[b, :], pb[b, :]
// (fixed order -> reproducible), summed over b by the column-sum kernel.  Replaces the dx kernel + the column pass that
// re-read dy and x.
template <int MAXV>
__global__ __launch_bounds__(256) void layernorm_bwd_fused_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                                  const float *__restrict__ w, const float *__restrict__ dres,
                                                                  float *__restrict__ dx, float *__restrict__ pw, float *__restrict__ pb,
                                                                  int rows, int C, float eps, __bf16 *__restrict__ dxb = nullptr,
                                                                  const __bf16 *__restrict__ dy16 = nullptr) {
    // (dy16: the incoming gradient as row-major bf16 instead of fp32 dy -- the input-gradient product's bf16 result, halo_gemm_rows)
    extern __shared__ float red[];                       // [2][C]: cross-wave sums of the partials
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int C4 = C >> 2;
    f32x4 wv[MAXV], aw[MAXV], ab[MAXV];
#pragma unroll
    for (int v = 0; v < MAXV; ++v) {
        const int q = v * 64 + lane;
        wv[v] = q < C4 ? *reinterpret_cast<const f32x4 *>(w + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
        aw[v] = f32x4{0.f, 0.f, 0.f, 0.f};
        ab[v] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float inv_c = 1.0f / (float)C;
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        const float *xr = x + (long)row * C, *dyr = dy16 ? nullptr : dy + (long)row * C;
        const __bf16 *dyr16 = dy16 ? dy16 + (long)row * C : nullptr;
        f32x4 xv[MAXV], dv[MAXV];
        float s = 0.f;
#pragma unroll
        for (int v = 0; v < MAXV; ++v) {
            const int q = v * 64 + lane;
            const bool in = q < C4;
            xv[v] = in ? *reinterpret_cast<const f32x4 *>(xr + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
            if (dy16) {
                typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                const bf16x4 d4 = in ? *reinterpret_cast<const bf16x4 *>(dyr16 + 4 * q) : bf16x4{(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
                dv[v] = f32x4{(float)d4[0], (float)d4[1], (float)d4[2], (float)d4[3]};
            } else {
                dv[v] = in ? *reinterpret_cast<const f32x4 *>(dyr + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            s += (xv[v][0] + xv[v][1]) + (xv[v][2] + xv[v][3]);
        }
        const float mean = wave_sum(s) * inv_c;
        float var = 0.f;
#pragma unroll
        for (int v = 0; v < MAXV; ++v) {
            if (v * 64 + lane < C4) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { xv[v][e] -= mean; var += xv[v][e] * xv[v][e]; }
            }
        }
        const float rstd = rsqrtf(wave_sum(var) * inv_c + eps);
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int v = 0; v < MAXV; ++v)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xv[v][e] *= rstd;                        // xhat (0 in the padding lanes)
                const float g = dv[v][e] * wv[v][e];
                sg += g;
                sgx += g * xv[v][e];
                aw[v][e] += dv[v][e] * xv[v][e];
                ab[v][e] += dv[v][e];
            }
        const float mg = wave_sum(sg) * inv_c, mgx = wave_sum(sgx) * inv_c;
        float *dxr = dx + (long)row * C;
        const float *rr = dres ? dres + (long)row * C : nullptr;
#pragma unroll
        for (int v = 0; v < MAXV; ++v) {
            const int q = v * 64 + lane;
            if (q >= C4) continue;
            f32x4 d;
#pragma unroll
            for (int e = 0; e < 4; ++e) d[e] = rstd * (dv[v][e] * wv[v][e] - mg - xv[v][e] * mgx);
            if (rr) {
                const f32x4 r4 = *reinterpret_cast<const f32x4 *>(rr + 4 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) d[e] += r4[e];
            }
            *reinterpret_cast<f32x4 *>(dxr + 4 * q) = d;
            if (dxb) {          // the same rows as row-major bf16: the operand of the next Linear's two gradient products
                typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                *reinterpret_cast<bf16x4 *>(dxb + (long)row * C + 4 * q) = bf16x4{(__bf16)d[0], (__bf16)d[1], (__bf16)d[2], (__bf16)d[3]};
            }
        }
    }
    // the four waves' partials, added in wave order
    for (int k = 0; k < 4; ++k) {
        if (wave == k) {
#pragma unroll
            for (int v = 0; v < MAXV; ++v) {
                const int q = v * 64 + lane;
                if (q >= C4) continue;
                f32x4 *rw = reinterpret_cast<f32x4 *>(red + 4 * q), *rb = reinterpret_cast<f32x4 *>(red + C + 4 * q);
                if (k == 0) { *rw = aw[v]; *rb = ab[v]; }
                else {
                    f32x4 a4 = *rw, b4 = *rb;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { a4[e] += aw[v][e]; b4[e] += ab[v][e]; }
                    *rw = a4; *rb = b4;
                }
            }
        }
        __syncthreads();
    }
    for (int c = threadIdx.x; c < C; c += 256) {
        pw[(long)blockIdx.x * C + c] = red[c];
        pb[(long)blockIdx.x * C + c] = red[C + c];
    }
}

// partial_w[chunk, c] = sum over the chunk's rows of dy * xhat, partial_b[chunk, c] = sum dy (fixed order -> reproducible)
__global__ __launch_bounds__(256) void layernorm_bwd_dw_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                               const float *__restrict__ stats, float *__restrict__ pw,
                                                               float *__restrict__ pb, int rows, int C, int rows_per_chunk) {
    const int c = blockIdx.x * 256 + threadIdx.x, chunk = blockIdx.y;
    if (c >= C) return;
    const int r0 = chunk * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
    float aw = 0.f, ab = 0.f;
    for (int r = r0; r < r1; ++r) {
        const float d = dy[(long)r * C + c];
        aw += d * (x[(long)r * C + c] - stats[2 * r]) * stats[2 * r + 1];
        ab += d;
    }
    pw[(long)chunk * C + c] = aw;
    pb[(long)chunk * C + c] = ab;
}

// kind 0: tanh-GELU (ha/attention.py:12-17), kind 1: exact GELU.  fwd: y = gelu(a);  bwd: da = dy * gelu'(a)
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const float *__restrict__ a, float *__restrict__ y, size_t n, int kind) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = gemm_activation(a[i], kind ? 8 : 2);
}
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ a, float *__restrict__ da,
                                                       size_t n, int kind) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    da[i] = dy[i] * gelu_grad(a[i], kind);
}

// logits[n,:] <- (softmax(logits[n,:]) - onehot(target[n])) * grad[n]   (0 where target == ignore_index), in place
__global__ __launch_bounds__(256) void cross_entropy_bwd_kernel(float *__restrict__ logits, const int64_t *__restrict__ target,
                                                                const float *__restrict__ lse, const float *__restrict__ grad,
                                                                long grad_stride, int V, long ld, long ignore_index) {
    const int n = blockIdx.x;
    const long tgt = target[n];
    float *row = logits + (long)n * ld;
    if (tgt == ignore_index) {
        for (int c = threadIdx.x; c < V; c += 256) row[c] = 0.f;
        return;
    }
    const float l = lse[n], g = grad[(long)n * grad_stride];
    for (int c = threadIdx.x; c < V; c += 256) row[c] = (expf(row[c] - l) - (c == tgt ? 1.0f : 0.f)) * g;
}

// dwte[ids[n], :] += dx[n, :]  (float atomics: tokens repeat)
__global__ __launch_bounds__(256) void embed_bwd_wte_kernel(const int64_t *__restrict__ ids, const float *__restrict__ dx,
                                                            float *__restrict__ dwte, int C, int vocab) {
    const int n = blockIdx.x;
    long id = ids[n];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    for (int c = threadIdx.x; c < C; c += 256) atomicAdd(dwte + id * C + c, dx[(long)n * C + c]);
}
// x[n, :] += p[n % T, :]   (StableEmbedding: the normalised position rows are shared by the batch, ha/attention.py:222-224)
__global__ __launch_bounds__(256) void add_rows_bcast_kernel(float *__restrict__ x, const float *__restrict__ p, long n_elem, int T, int C) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_elem) return;
    const long row = idx / C;
    x[idx] += p[(row % T) * C + idx % C];
}

// dwpe[pos0 + t, c] (+)= sum_b dx[b*T + t, c]  (fixed order)
__global__ __launch_bounds__(256) void embed_bwd_wpe_kernel(const float *__restrict__ dx, float *__restrict__ dwpe, int B, int T, int C,
                                                            int pos0, int accumulate) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)T * C) return;
    const int t = (int)(idx / C), c = (int)(idx % C);
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += dx[((long)b * T + t) * C + c];
    float *o = dwpe + (long)(pos0 + t) * C + c;
    *o = accumulate ? *o + s : s;
}

// fp32 -> row-major bf16 (operands of halo_gemm_split_io / halo_gemm_tn_bf16 that no producing launch wrote as bf16)
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float *__restrict__ x, __bf16 *__restrict__ y, long n8) {
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        const f32x4 a = *reinterpret_cast<const f32x4 *>(x + 8 * i), b = *reinterpret_cast<const f32x4 *>(x + 8 * i + 4);
        *reinterpret_cast<bf16x8 *>(y + 8 * i) =
            bf16x8{(__bf16)a[0], (__bf16)a[1], (__bf16)a[2], (__bf16)a[3], (__bf16)b[0], (__bf16)b[1], (__bf16)b[2], (__bf16)b[3]};
    }
}


// the MLP's elementwise passes with a bf16 result: y = gelu(a), da = dy * gelu'(a)
template <bool BWD>
__global__ __launch_bounds__(256) void gelu_bf16_kernel(const float *__restrict__ dy, const float *__restrict__ a, __bf16 *__restrict__ y, long n8,
                                                        int exact) {
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        const f32x4 a0 = *reinterpret_cast<const f32x4 *>(a + 8 * i), a1 = *reinterpret_cast<const f32x4 *>(a + 8 * i + 4);
        float v[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        if (BWD) {
            const f32x4 d0 = *reinterpret_cast<const f32x4 *>(dy + 8 * i), d1 = *reinterpret_cast<const f32x4 *>(dy + 8 * i + 4);
            const float d[8] = {d0[0], d0[1], d0[2], d0[3], d1[0], d1[1], d1[2], d1[3]};
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = d[e] * gelu_grad(v[e], exact);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = gemm_activation(v[e], exact ? 8 : 2);
        }
        *reinterpret_cast<bf16x8 *>(y + 8 * i) = bf16x8{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3], (__bf16)v[4], (__bf16)v[5],
                                                        (__bf16)v[6], (__bf16)v[7]};
    }
}

// ... and with bf16 on both sides (the c_fc product left its result as row-major bf16: halo_gemm_rows): y = gelu(a), da = dy * gelu'(a);
// the arithmetic in fp32 on the bf16 values
template <bool BWD>
__global__ __launch_bounds__(256) void gelu_b16_kernel(const __bf16 *__restrict__ dy, const __bf16 *__restrict__ a, __bf16 *__restrict__ y, long n8,
                                                       int exact) {
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        const bf16x8 av = *reinterpret_cast<const bf16x8 *>(a + 8 * i);
        bf16x8 o;
        if (BWD) {
            const bf16x8 dv = *reinterpret_cast<const bf16x8 *>(dy + 8 * i);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (__bf16)((float)dv[e] * gelu_grad((float)av[e], exact));
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (__bf16)gemm_activation((float)av[e], exact ? 8 : 2);
        }
        *reinterpret_cast<bf16x8 *>(y + 8 * i) = o;
    }
}

int layernorm_bwd_impl(const float *dy, const float *x, const float *weight, const float *dres, float *dx, __bf16 *dxb, float *dweight,
                              float *dbias, void *workspace, int rows, int C, float eps, hipStream_t st, const __bf16 *dy16 = nullptr) {
    float *stats = (float *)workspace, *pw = stats + (size_t)rows * 2, *pb = pw + (size_t)HALO_LN_BWD_CHUNKS * C;
    const bool aligned = (((uintptr_t)dy | (uintptr_t)x | (uintptr_t)weight | (uintptr_t)dres | (uintptr_t)dx) % 16) == 0 && (uintptr_t)dy16 % 8 == 0;
    if (dy16 && !(C % 4 == 0 && C <= 2048 && aligned)) return HALO_ENOTSUP;
    if (C % 4 == 0 && C <= 2048 && aligned) {
        const int wgs = min(HALO_LN_BWD_CHUNKS, (rows + 3) / 4);
        const size_t lds = (size_t)2 * C * sizeof(float);
        if (C <= 1024)
            hipLaunchKernelGGL(layernorm_bwd_fused_kernel<4>, dim3(wgs), dim3(256), lds, st, dy, x, weight, dres, dx, pw, pb, rows, C, eps, dxb, dy16);
        else
            hipLaunchKernelGGL(layernorm_bwd_fused_kernel<8>, dim3(wgs), dim3(256), lds, st, dy, x, weight, dres, dx, pw, pb, rows, C, eps, dxb, dy16);
        if (halo_launch_status() != HALO_OK) return HALO_ELAUNCH;
        return halo_colsum2(pw, wgs, C, C, dweight, nullptr, st) || (dbias ? halo_colsum2(pb, wgs, C, C, dbias, nullptr, st) : HALO_OK);
    }
    hipLaunchKernelGGL(layernorm_bwd_dx_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, dy, x, weight, dres, dx, stats, rows, C, eps);
    const int chunks = rows < HALO_LN_BWD_CHUNKS ? rows : HALO_LN_BWD_CHUNKS, rpc = (rows + chunks - 1) / chunks;
    const int used = (rows + rpc - 1) / rpc;
    hipLaunchKernelGGL(layernorm_bwd_dw_kernel, dim3((C + 255) / 256, used), dim3(256), 0, st, dy, x, stats, pw, pb, rows, C, rpc);
    if (halo_launch_status() != HALO_OK) return HALO_ELAUNCH;
    return halo_colsum2(pw, used, C, C, dweight, nullptr, st) || (dbias ? halo_colsum2(pb, used, C, C, dbias, nullptr, st) : HALO_OK);
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
    hipLaunchKernelGGL(cross_entropy_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, logits, targets, loss, (float *)nullptr, V,
                       ld, ignore_index);
    return halo_launch_status();
}

int halo_cross_entropy_fwd_lse(const float *logits, const int64_t *targets, float *loss, float *lse, int rows, int V, long ld,
                               long ignore_index, halo_stream_t stream) {
    HALO_CHECK_ARG(logits && targets && loss && lse && rows > 0 && V > 0 && ld >= V);
    hipLaunchKernelGGL(cross_entropy_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, logits, targets, loss, lse, V, ld,
                       ignore_index);
    return halo_launch_status();
}

int halo_cross_entropy_bwd(float *logits, const int64_t *targets, const float *lse, const float *grad, long grad_stride, int rows,
                           int V, long ld, long ignore_index, halo_stream_t stream) {
    HALO_CHECK_ARG(logits && targets && lse && grad && rows > 0 && V > 0 && ld >= V && (grad_stride == 0 || grad_stride == 1));
    hipLaunchKernelGGL(cross_entropy_bwd_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, logits, targets, lse, grad, grad_stride,
                       V, ld, ignore_index);
    return halo_launch_status();
}

size_t halo_layernorm_bwd_workspace_bytes(int rows, int C) {
    if (rows <= 0 || C <= 0) return 0;
    return ((size_t)rows * 2 + (size_t)2 * HALO_LN_BWD_CHUNKS * C) * sizeof(float);
}

int halo_cast_bf16(const float *x, void *y, size_t n, halo_stream_t stream) {
    HALO_CHECK_ARG(x && y && n > 0 && n % 8 == 0 && (((uintptr_t)x | (uintptr_t)y) % 16) == 0);
    const long n8 = (long)(n / 8);
    const long want = (n8 + 255) / 256;
    hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)(want > 8192 ? 8192 : want)), dim3(256), 0, (hipStream_t)stream, x, (__bf16 *)y, n8);
    return halo_launch_status();
}

int halo_gelu_bf16(const float *a, void *y_bf16, size_t n, int exact, halo_stream_t stream) {
    HALO_CHECK_ARG(a && y_bf16 && n > 0 && n % 8 == 0 && (((uintptr_t)a | (uintptr_t)y_bf16) % 16) == 0);
    const long n8 = (long)(n / 8), want = (n8 + 255) / 256;
    hipLaunchKernelGGL(gelu_bf16_kernel<false>, dim3((unsigned)(want > 16384 ? 16384 : want)), dim3(256), 0, (hipStream_t)stream, nullptr, a,
                       (__bf16 *)y_bf16, n8, exact);
    return halo_launch_status();
}

int halo_gelu_bwd_bf16(const float *dy, const float *a, void *da_bf16, size_t n, int exact, halo_stream_t stream) {
    HALO_CHECK_ARG(dy && a && da_bf16 && n > 0 && n % 8 == 0 && (((uintptr_t)dy | (uintptr_t)a | (uintptr_t)da_bf16) % 16) == 0);
    const long n8 = (long)(n / 8), want = (n8 + 255) / 256;
    hipLaunchKernelGGL(gelu_bf16_kernel<true>, dim3((unsigned)(want > 16384 ? 16384 : want)), dim3(256), 0, (hipStream_t)stream, dy, a,
                       (__bf16 *)da_bf16, n8, exact);
    return halo_launch_status();
}

int halo_gelu_b16(const void *a_bf16, void *y_bf16, size_t n, int exact, halo_stream_t stream) {
    HALO_CHECK_ARG(a_bf16 && y_bf16 && n > 0 && n % 8 == 0 && (((uintptr_t)a_bf16 | (uintptr_t)y_bf16) % 16) == 0);
    const size_t want = (n / 8 + 255) / 256;
    hipLaunchKernelGGL(gelu_b16_kernel<false>, dim3((unsigned)(want > 16384 ? 16384 : want)), dim3(256), 0, (hipStream_t)stream, nullptr,
                       (const __bf16 *)a_bf16, (__bf16 *)y_bf16, (long)(n / 8), exact);
    return halo_launch_status();
}

int halo_gelu_bwd_b16(const void *dy_bf16, const void *a_bf16, void *da_bf16, size_t n, int exact, halo_stream_t stream) {
    HALO_CHECK_ARG(dy_bf16 && a_bf16 && da_bf16 && n > 0 && n % 8 == 0 && (((uintptr_t)dy_bf16 | (uintptr_t)a_bf16 | (uintptr_t)da_bf16) % 16) == 0);
    const size_t want = (n / 8 + 255) / 256;
    hipLaunchKernelGGL(gelu_b16_kernel<true>, dim3((unsigned)(want > 16384 ? 16384 : want)), dim3(256), 0, (hipStream_t)stream,
                       (const __bf16 *)dy_bf16, (const __bf16 *)a_bf16, (__bf16 *)da_bf16, (long)(n / 8), exact);
    return halo_launch_status();
}

int halo_layernorm_bwd(const float *dy, const float *x, const float *weight, const float *dres, float *dx, float *dweight,
                       float *dbias, void *workspace, int rows, int C, float eps, halo_stream_t stream) {
    HALO_CHECK_ARG(dy && x && weight && dx && dweight && workspace && rows > 0 && C > 0);
    return layernorm_bwd_impl(dy, x, weight, dres, dx, nullptr, dweight, dbias, workspace, rows, C, eps, (hipStream_t)stream);
}

int halo_layernorm_bwd_bf16(const float *dy, const float *x, const float *weight, const float *dres, float *dx, void *dx_bf16, float *dweight,
                            float *dbias, void *workspace, int rows, int C, float eps, halo_stream_t stream) {
    HALO_CHECK_ARG(dy && x && weight && dx && dx_bf16 && dweight && workspace && rows > 0 && C > 0);
    HALO_CHECK_ARG(C % 4 == 0 && C <= 2048 && (((uintptr_t)dy | (uintptr_t)x | (uintptr_t)weight | (uintptr_t)dres | (uintptr_t)dx | (uintptr_t)dx_bf16) % 16) == 0);
    return layernorm_bwd_impl(dy, x, weight, dres, dx, (__bf16 *)dx_bf16, dweight, dbias, workspace, rows, C, eps, (hipStream_t)stream);
}


int halo_layernorm_bwd_b16(const void *dy_bf16, const float *x, const float *weight, const float *dres, float *dx, void *dx_bf16, float *dweight,
                           float *dbias, void *workspace, int rows, int C, float eps, halo_stream_t stream) {
    HALO_CHECK_ARG(dy_bf16 && x && weight && dx && dweight && workspace && rows > 0 && C > 0);
    HALO_CHECK_ARG(C % 4 == 0 && C <= 2048 && (((uintptr_t)x | (uintptr_t)weight | (uintptr_t)dres | (uintptr_t)dx | (uintptr_t)dx_bf16) % 16) == 0 &&
                   (uintptr_t)dy_bf16 % 8 == 0);
    return layernorm_bwd_impl(nullptr, x, weight, dres, dx, (__bf16 *)dx_bf16, dweight, dbias, workspace, rows, C, eps, (hipStream_t)stream,
                              (const __bf16 *)dy_bf16);
}

int halo_gelu_fwd(const float *a, float *y, size_t n, int exact, halo_stream_t stream) {
    HALO_CHECK_ARG(a && y && n > 0);
    hipLaunchKernelGGL(gelu_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, y, n, exact);
    return halo_launch_status();
}

int halo_gelu_bwd(const float *dy, const float *a, float *da, size_t n, int exact, halo_stream_t stream) {
    HALO_CHECK_ARG(dy && a && da && n > 0);
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy, a, da, n, exact);
    return halo_launch_status();
}

int halo_add_rows_bcast(float *x, const float *p, int rows, int T, int C, halo_stream_t stream) {
    HALO_CHECK_ARG(x && p && rows > 0 && T > 0 && C > 0);
    const long n = (long)rows * C;
    hipLaunchKernelGGL(add_rows_bcast_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, p, n, T, C);
    return halo_launch_status();
}

int halo_embed_bwd(const int64_t *ids, const float *dx, float *dwte, float *dwpe, int B, int T, int C, int pos0, int vocab,
                   int accumulate_wpe, halo_stream_t stream) {
    HALO_CHECK_ARG(ids && dx && (dwte || dwpe) && B > 0 && T > 0 && C > 0 && pos0 >= 0 && vocab > 0);
    hipStream_t st = (hipStream_t)stream;
    if (dwte) hipLaunchKernelGGL(embed_bwd_wte_kernel, dim3(B * T), dim3(256), 0, st, ids, dx, dwte, C, vocab);
    if (dwpe) {
        const long n = (long)T * C;
        hipLaunchKernelGGL(embed_bwd_wpe_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dx, dwpe, B, T, C, pos0,
                           accumulate_wpe);
    }
    return halo_launch_status();
}

}  // extern "C"
