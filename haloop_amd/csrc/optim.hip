// Gradient-norm clipping and AdamW on flat fp32 buffers (one pass each over the parameters).
#include <math.h>
#include <stdlib.h>
#include "halo_common.h"
#include "halo_internal.h"

namespace {

__global__ __launch_bounds__(256) void sumsq_kernel(const float *__restrict__ x, size_t n, float *__restrict__ partials) {
    __shared__ float red[4];
    float s = 0.f;
    const size_t n4 = n / 4;
    const f32x4 *x4 = reinterpret_cast<const f32x4 *>(x);
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const f32x4 v = x4[i];
        s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    if (blockIdx.x == 0)
        for (size_t i = n4 * 4 + threadIdx.x; i < n; i += 256) s += x[i] * x[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// the same over up to 8 ranges of one buffer, as if they were one concatenated array (the sharded data-parallel update: a rank's
// chunks of the reduce-scattered spans); every range begins and ends on a multiple of 4 elements
struct Ranges8 {
    size_t begin[8], len4[8];          // element offset, length in float4
    int n;
};
__global__ __launch_bounds__(256) void sumsq_ranges_kernel(const float *__restrict__ x, Ranges8 r, float *__restrict__ partials) {
    __shared__ float red[4];
    float s = 0.f;
    size_t base = 0;
    for (int k = 0; k < r.n; ++k) {
        const f32x4 *x4 = reinterpret_cast<const f32x4 *>(x + r.begin[k]);
        // (the concatenated index space is dealt to the threads round-robin, as sumsq_kernel deals one array)
        const size_t stride = (size_t)gridDim.x * 256, me = blockIdx.x * (size_t)256 + threadIdx.x;
        size_t i = (me + stride - base % stride) % stride;
        for (; i < r.len4[k]; i += stride) {
            const f32x4 v = x4[i];
            s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
        }
        base += r.len4[k];
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// dst (bf16, contiguous) <- the concatenation of up to 8 ranges of src (fp32), 4 elements per thread and pass
__global__ __launch_bounds__(256) void pack_ranges_bf16_kernel(const float *__restrict__ src, Ranges8 r, __bf16 *__restrict__ dst) {
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    size_t base = 0;
    for (int k = 0; k < r.n; ++k) {
        const f32x4 *s4 = reinterpret_cast<const f32x4 *>(src + r.begin[k]);
        bf16x4 *d4 = reinterpret_cast<bf16x4 *>(dst) + base;
        for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < r.len4[k]; i += (size_t)gridDim.x * 256) {
            const f32x4 v = s4[i];
            bf16x4 o;
            o[0] = (__bf16)v[0]; o[1] = (__bf16)v[1]; o[2] = (__bf16)v[2]; o[3] = (__bf16)v[3];
            d4[i] = o;
        }
        base += r.len4[k];
    }
}

// the inverse on the gathered buffer: stage [world][sum of chunks] bf16, rank-major; span k of dst begins at r.begin[k] and is cut into
// `world` chunks of r.len4[k] float4 each; chunk `skip` (the caller's own: its fp32 master values stay) is left alone
__global__ __launch_bounds__(256) void expand_ranges_bf16_kernel(const __bf16 *__restrict__ stage, Ranges8 r, int world, int skip,
                                                                 float *__restrict__ dst) {
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    size_t per_rank = 0;
    for (int k = 0; k < r.n; ++k) per_rank += r.len4[k];
    size_t off = 0;
    for (int k = 0; k < r.n; ++k) {
        const size_t c = r.len4[k], total = c * (size_t)world;
        f32x4 *d4 = reinterpret_cast<f32x4 *>(dst + r.begin[k]);
        for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
            const size_t rk = i / c, j = i - rk * c;
            if ((int)rk == skip) continue;
            const bf16x4 v = reinterpret_cast<const bf16x4 *>(stage)[rk * per_rank + off + j];
            d4[i] = f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
        }
        off += c;
    }
}

__global__ __launch_bounds__(256) void clip_coef_kernel(const float *__restrict__ partials, int count, float max_norm,
                                                        float *coef, float *norm_out, uint32_t *applied_steps, const uint32_t *status) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < count; i += 256) s += partials[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float norm = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
        const float c = max_norm / (norm + 1e-6f);
        // a non-finite norm poisons both scales: halo_adamw skips the update when its scale is NaN
        // (the reference skips the batch on a NaN/Inf loss or gradient norm, ha/loop.py:167-189); so does a raised status word
        // (halo_set_status_word: a persistent recurrence of this step gave up a bounded wait, its gradients are not to be applied)
        const bool finite = isfinite(norm) && !(status && *status != 0u);
        coef[0] = finite ? (c > 1.0f ? 1.0f : c) : NAN;
        coef[1] = finite ? 1.0f : NAN;
        if (norm_out) *norm_out = norm;
        if (applied_steps && finite) *applied_steps += 1u;      // the Adam step count advances only with an applied update
    }
}

// One AdamW element update (torch.optim.AdamW, _single_tensor).  Every kernel below calls this, with contraction pinned, so that the
// per-tensor, ranged and multi-tensor launches (and their vector bodies and scalar tails) produce the same bits.
__device__ __forceinline__ void adam_update(float &p, float &m, float &v, float g, float decay_mul, float beta1_w, float beta2,
                                            float beta2_w, float step_size, float bc2_sqrt, float eps) {
#pragma clang fp contract(off)
    p = p * decay_mul;
    m = __builtin_fmaf(g - m, beta1_w, m);
    v = __builtin_fmaf(beta2_w * g, g, v * beta2);
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    p = p - step_size * (m / denom);
}

struct AdamArgs {
    float *p;
    const float *g;
    float *m;
    float *v;
    size_t n;
    float decay_mul;     // 1 - lr*wd
    float beta1_w;       // 1 - beta1
    float beta2;
    float beta2_w;       // 1 - beta2
    float step_size;     // lr / (1 - beta1^t)
    float bc2_sqrt;      // sqrt(1 - beta2^t)
    float eps;
    const float *grad_scale;
};

__global__ __launch_bounds__(256) void adamw_kernel(const AdamArgs a) {
    const float gs = a.grad_scale ? *a.grad_scale : 1.0f;
    if (gs != gs) return;   // NaN scale: skip this update entirely
    const size_t n4 = a.n / 4;
    f32x4 *p4 = reinterpret_cast<f32x4 *>(a.p), *m4 = reinterpret_cast<f32x4 *>(a.m), *v4 = reinterpret_cast<f32x4 *>(a.v);
    const f32x4 *g4 = reinterpret_cast<const f32x4 *>(a.g);
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        f32x4 p = p4[i], m = m4[i], v = v4[i];
        const f32x4 g = g4[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float ge = a.grad_scale ? g[e] * gs : g[e];
            float pe = p[e], me = m[e], ve = v[e];
            adam_update(pe, me, ve, ge, a.decay_mul, a.beta1_w, a.beta2, a.beta2_w, a.step_size, a.bc2_sqrt, a.eps);
            p[e] = pe; m[e] = me; v[e] = ve;
        }
        p4[i] = p; m4[i] = m; v4[i] = v;
    }
    if (blockIdx.x == 0) {
        for (size_t i = n4 * 4 + threadIdx.x; i < a.n; i += 256) {
            const float ge = a.grad_scale ? a.g[i] * gs : a.g[i];
            float pe = a.p[i], me = a.m[i], ve = a.v[i];
            adam_update(pe, me, ve, ge, a.decay_mul, a.beta1_w, a.beta2, a.beta2_w, a.step_size, a.bc2_sqrt, a.eps);
            a.p[i] = pe; a.m[i] = me; a.v[i] = ve;
        }
    }
}


// AdamW over several contiguous ranges of the same flat buffers in ONE launch (each range has its own weight decay and
// gradient scale: clipped encoder / unclipped recognizer, decayed weights / undecayed biases), plus the dropout step counter's
// increment: the per-step optimizer was 4 AdamW launches + 1 counter launch, three of them over a few hundred elements.
constexpr int ADAM_MAX_RANGES = 8;
struct AdamRangesArgs {
    float *p;
    const float *g;
    float *m;
    float *v;
    size_t begin4[ADAM_MAX_RANGES], end4[ADAM_MAX_RANGES];      // in float4 units
    float decay_mul[ADAM_MAX_RANGES];
    const float *grad_scale[ADAM_MAX_RANGES];
    int n_ranges;
    float beta1_w, beta2, beta2_w, step_size, bc2_sqrt, eps;
    uint32_t *counter;
    const uint32_t *step_dev;     // optional: 1-based update count on the device (halo_clip_coef_step); overrides step_size / bc2_sqrt
    const float *lr_dev;          // optional (with step_dev): the learning rate read from the device, so a schedule reaches a captured launch
    float weight_decay[ADAM_MAX_RANGES];
    float lr, beta1;
};

template <bool NT>
__global__ __launch_bounds__(256) void adamw_ranges_kernel(const AdamRangesArgs a) {
    f32x4 *p4 = reinterpret_cast<f32x4 *>(a.p), *m4 = reinterpret_cast<f32x4 *>(a.m), *v4 = reinterpret_cast<f32x4 *>(a.v);
    const f32x4 *g4 = reinterpret_cast<const f32x4 *>(a.g);
    float step_size = a.step_size, bc2_sqrt = a.bc2_sqrt;
    const float lr = a.lr_dev ? *a.lr_dev : a.lr;
    if (a.step_dev) {
        // the same scalar preparation as the host path (double, like torch's python-side bias corrections), from the device counter
        const double t = (double)max(*a.step_dev, 1u);
        step_size = (float)((double)lr / (1.0 - pow((double)a.beta1, t)));
        bc2_sqrt = (float)sqrt(1.0 - pow((double)a.beta2, t));
    }
    for (int r = 0; r < a.n_ranges; ++r) {
        const float gs = a.grad_scale[r] ? *a.grad_scale[r] : 1.0f;
        if (gs != gs) continue;                                   // NaN scale: this range's update is skipped
        const float decay_mul = a.lr_dev ? (float)(1.0 - (double)lr * (double)a.weight_decay[r]) : a.decay_mul[r];
        for (size_t i = a.begin4[r] + blockIdx.x * (size_t)256 + threadIdx.x; i < a.end4[r]; i += (size_t)gridDim.x * 256) {
            // the moments and the gradient are touched once per step: streamed past the caches (non-temporal; HALO_ADAMW_NT=0: plain).  Same
            // kernel time (62 us), but the step around it is 9-10 us shorter on the same box (0.428-0.434 -> 0.419-0.423 ms): what the next
            // launches read is still cached.  The parameters stay temporal (the packing launch reads them back; non-temporal: no gain)
            f32x4 p = p4[i], m = NT ? __builtin_nontemporal_load(m4 + i) : m4[i], v = NT ? __builtin_nontemporal_load(v4 + i) : v4[i];
            const f32x4 g = NT ? __builtin_nontemporal_load(g4 + i) : g4[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float ge = a.grad_scale[r] ? g[e] * gs : g[e];
                float pe = p[e], me = m[e], ve = v[e];
                adam_update(pe, me, ve, ge, decay_mul, a.beta1_w, a.beta2, a.beta2_w, step_size, bc2_sqrt, a.eps);
                p[e] = pe; m[e] = me; v[e] = ve;
            }
            p4[i] = p;
            if (NT) { __builtin_nontemporal_store(m, m4 + i); __builtin_nontemporal_store(v, v4 + i); }
            else { m4[i] = m; v4[i] = v; }
        }
    }
    if (a.counter && blockIdx.x == 0 && threadIdx.x == 0) *a.counter += 1u;
}

// One launch over MANY tensors (torch.optim.AdamW(fused / foreach) on a model's parameter list): the device table names each
// tensor's p / g / m / v and its decay; a workgroup takes one 16384-element chunk (chunk table: tensor index, chunk index).
// Same per-element arithmetic as adamw_kernel, so the results are bit-identical to per-tensor launches.
struct AdamTensor {      // static per parameter: lives in a device table built once
    float *p;
    float *m;
    float *v;
    unsigned long long n;
    float weight_decay;  // of this tensor; the decay factor 1 - lr*wd is formed in the kernel from the CURRENT lr
    int pad;
};
constexpr unsigned ADAM_CHUNK = 16384;
constexpr int ADAM_MULTI_MAX = 256;
// the gradient tensors are new allocations every backward: their addresses travel in the kernel arguments (copied at launch,
// so the host may run ahead and prepare the next step's list while this one is still queued)
struct AdamGrads { const float *g[ADAM_MULTI_MAX]; };

__global__ __launch_bounds__(256) void adamw_multi_kernel(const AdamTensor *__restrict__ tensors, const uint2 *__restrict__ chunks,
                                                          const AdamGrads grads, float lr, float beta1_w, float beta2, float beta2_w,
                                                          float step_size, float bc2_sqrt, float eps,
                                                          const float *__restrict__ grad_scale) {
    const float gs = grad_scale ? *grad_scale : 1.0f;
    if (gs != gs) return;   // NaN scale: skip this update entirely
    const uint2 c = chunks[blockIdx.x];
    const AdamTensor t = tensors[c.x];
    const float decay_mul = (float)(1.0 - (double)lr * (double)t.weight_decay);      // as halo_adamw forms it on the host
    const float *tg = grads.g[c.x];
    const size_t begin = (size_t)c.y * ADAM_CHUNK, end = min(begin + (size_t)ADAM_CHUNK, (size_t)t.n);
    auto update = [&](float &pe, float &me, float &ve, float g) {
        adam_update(pe, me, ve, grad_scale ? g * gs : g, decay_mul, beta1_w, beta2, beta2_w, step_size, bc2_sqrt, eps);
    };
    const bool vec = (((uintptr_t)t.p | (uintptr_t)tg | (uintptr_t)t.m | (uintptr_t)t.v) % 16) == 0;
    size_t done = begin;
    if (vec) {
        const size_t e4 = end / 4;
        f32x4 *p4 = reinterpret_cast<f32x4 *>(t.p), *m4 = reinterpret_cast<f32x4 *>(t.m), *v4 = reinterpret_cast<f32x4 *>(t.v);
        const f32x4 *g4 = reinterpret_cast<const f32x4 *>(tg);
        for (size_t i = begin / 4 + threadIdx.x; i < e4; i += 256) {
            // (moments and gradient past the caches, as adamw_ranges_kernel: touched once per step)
            f32x4 p = p4[i], m = __builtin_nontemporal_load(m4 + i), v = __builtin_nontemporal_load(v4 + i);
            const f32x4 g = __builtin_nontemporal_load(g4 + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float pe = p[e], me = m[e], ve = v[e];
                update(pe, me, ve, g[e]);
                p[e] = pe; m[e] = me; v[e] = ve;
            }
            p4[i] = p;
            __builtin_nontemporal_store(m, m4 + i);
            __builtin_nontemporal_store(v, v4 + i);
        }
        done = e4 * 4;
    }
    for (size_t i = done + threadIdx.x; i < end; i += 256) {
        float pe = t.p[i], me = t.m[i], ve = t.v[i];
        update(pe, me, ve, tg[i]);
        t.p[i] = pe; t.m[i] = me; t.v[i] = ve;
    }
}

}  // namespace

namespace {
// y = alpha * y + beta * x : gradient accumulation over micro-batches (ha/loop.py:176-181: loss / accumulate, backward, ...).
// alpha == 0 never reads y (0 * NaN would keep a poisoned sum alive: the first micro-step of a cycle must be a plain scaled copy);
// guard != NULL with a non-finite *guard drops x's contribution: the reference skips a micro-batch whose loss is NaN/Inf
// (ha/loop.py:167-174) and the cycle recovers.
__global__ __launch_bounds__(256) void scale_add_kernel(float *__restrict__ y, const float *__restrict__ x, float alpha, float beta, size_t n4,
                                                        size_t n, const float *__restrict__ guard) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool take_x = !guard || isfinite(*guard);
    const bool take_y = alpha != 0.f;
    if (i < n4) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
        if (take_y) a = alpha * reinterpret_cast<f32x4 *>(y)[i];
        if (take_x) b = beta * reinterpret_cast<const f32x4 *>(x)[i];
        reinterpret_cast<f32x4 *>(y)[i] = a + b;
    } else if (i == n4) {
        for (size_t e = 4 * n4; e < n; ++e) y[e] = (take_y ? alpha * y[e] : 0.f) + (take_x ? beta * x[e] : 0.f);
    }
}
}  // namespace

namespace {
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
// gradient wire format of the data-parallel all-reduce (haloop_amd/dp.py, wire_dtype='bf16'): fp32 -> bf16 before, bf16 -> fp32 * scale after
__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float *__restrict__ x, __bf16 *__restrict__ y, size_t n4, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const f32x4 v = reinterpret_cast<const f32x4 *>(x)[i];
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
        reinterpret_cast<bf16x4 *>(y)[i] = o;
    }
    if (blockIdx.x == 0)
        for (size_t e = 4 * n4 + threadIdx.x; e < n; e += 256) y[e] = (__bf16)x[e];
}
__global__ __launch_bounds__(256) void cast_bf16_f32_kernel(const __bf16 *__restrict__ x, float *__restrict__ y, float scale, size_t n4, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const bf16x4 v = reinterpret_cast<const bf16x4 *>(x)[i];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (float)v[e] * scale;
        reinterpret_cast<f32x4 *>(y)[i] = o;
    }
    if (blockIdx.x == 0)
        for (size_t e = 4 * n4 + threadIdx.x; e < n; e += 256) y[e] = (float)x[e] * scale;
}
inline unsigned cast_grid(size_t n4) {
    size_t g = (n4 + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}
}  // namespace

extern "C" {

int halo_cast_f32_bf16(const float *x, void *y, size_t n, halo_stream_t stream) {
    HALO_CHECK_ARG(x && y && ((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 8 == 0));
    if (n == 0) return HALO_OK;
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(cast_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, x, (__bf16 *)y, n / 4, n);
    return halo_launch_status();
}

int halo_cast_bf16_f32(const void *x, float *y, float scale, size_t n, halo_stream_t stream) {
    HALO_CHECK_ARG(x && y && ((uintptr_t)x % 8 == 0) && ((uintptr_t)y % 16 == 0));
    if (n == 0) return HALO_OK;
    hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(cast_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, (const __bf16 *)x, y, scale, n / 4, n);
    return halo_launch_status();
}

int halo_scale_add_guarded(float *y, const float *x, float alpha, float beta, size_t n, const float *guard, halo_stream_t stream) {
    HALO_CHECK_ARG(y && x && ((uintptr_t)y % 16 == 0) && ((uintptr_t)x % 16 == 0));
    if (n == 0) return HALO_OK;
    const size_t n4 = n / 4;
    hipLaunchKernelGGL(scale_add_kernel, dim3((unsigned)((n4 + 1 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, y, x, alpha, beta, n4, n,
                       guard);
    return halo_launch_status();
}

int halo_scale_add(float *y, const float *x, float alpha, float beta, size_t n, halo_stream_t stream) {
    return halo_scale_add_guarded(y, x, alpha, beta, n, nullptr, stream);
}

int halo_sumsq(const float *x, size_t n, float *partials, halo_stream_t stream) {
    HALO_CHECK_ARG(x && partials && ((uintptr_t)x % 16 == 0));
    hipLaunchKernelGGL(sumsq_kernel, dim3(HALO_SUMSQ_PARTS), dim3(256), 0, (hipStream_t)stream, x, n, partials);
    return halo_launch_status();
}

namespace {
int make_ranges8(int n, const size_t *begin, const size_t *end, Ranges8 &r, size_t &total4) {
    if (n < 1 || n > 8 || !begin || !end) return HALO_EINVAL;
    r.n = n; total4 = 0;
    for (int k = 0; k < n; ++k) {
        if (end[k] < begin[k] || begin[k] % 4 || end[k] % 4) return HALO_EINVAL;
        r.begin[k] = begin[k]; r.len4[k] = (end[k] - begin[k]) / 4;
        total4 += r.len4[k];
    }
    return HALO_OK;
}
}  // namespace

int halo_sumsq_ranges(const float *x, int n, const size_t *begin, const size_t *end, float *partials, halo_stream_t stream) {
    HALO_CHECK_ARG(x && partials && ((uintptr_t)x % 16 == 0));
    Ranges8 r; size_t total4;
    { const int rc = make_ranges8(n, begin, end, r, total4); if (rc != HALO_OK) return rc; }
    hipLaunchKernelGGL(sumsq_ranges_kernel, dim3(HALO_SUMSQ_PARTS), dim3(256), 0, (hipStream_t)stream, x, r, partials);
    return halo_launch_status();
}

int halo_pack_ranges_bf16(const float *src, int n, const size_t *begin, const size_t *end, void *dst, halo_stream_t stream) {
    HALO_CHECK_ARG(src && dst && ((uintptr_t)src % 16 == 0) && ((uintptr_t)dst % 8 == 0));
    Ranges8 r; size_t total4;
    { const int rc = make_ranges8(n, begin, end, r, total4); if (rc != HALO_OK) return rc; }
    if (total4 == 0) return HALO_OK;
    const unsigned blocks = (unsigned)((total4 + 255) / 256 < 2048 ? (total4 + 255) / 256 : 2048);
    hipLaunchKernelGGL(pack_ranges_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, r, (__bf16 *)dst);
    return halo_launch_status();
}

int halo_expand_ranges_bf16(const void *stage, int n, const size_t *begin, const size_t *chunk, int world, int skip_rank, float *dst,
                            halo_stream_t stream) {
    HALO_CHECK_ARG(stage && dst && world >= 1 && ((uintptr_t)dst % 16 == 0) && ((uintptr_t)stage % 8 == 0));
    Ranges8 r; size_t total4;
    size_t end[8];
    if (n < 1 || n > 8 || !begin || !chunk) return HALO_EINVAL;
    for (int k = 0; k < n; ++k) end[k] = begin[k] + chunk[k];
    { const int rc = make_ranges8(n, begin, end, r, total4); if (rc != HALO_OK) return rc; }
    if (total4 == 0) return HALO_OK;
    const size_t all4 = total4 * (size_t)world;
    const unsigned blocks = (unsigned)((all4 + 255) / 256 < 4096 ? (all4 + 255) / 256 : 4096);
    hipLaunchKernelGGL(expand_ranges_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const __bf16 *)stage, r, world, skip_rank, dst);
    return halo_launch_status();
}

int halo_clip_coef_step(const float *partials, int count, float max_norm, float *coef, float *norm_out, uint32_t *applied_steps,
                        halo_stream_t stream) {
    HALO_CHECK_ARG(partials && coef && count > 0);
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partials, count, max_norm, coef,
                       norm_out, applied_steps, (const uint32_t *)halo_ctx_cur().status);
    return halo_launch_status();
}

int halo_clip_coef(const float *partials, int count, float max_norm, float *coef, float *norm_out,
                   halo_stream_t stream) {
    return halo_clip_coef_step(partials, count, max_norm, coef, norm_out, nullptr, stream);
}

int halo_adamw(float *p, const float *g, float *m, float *v, size_t n, float lr, float beta1, float beta2, float eps,
               float weight_decay, int step, const float *grad_scale, halo_stream_t stream) {
    HALO_CHECK_ARG(p && g && m && v && step >= 1);
    HALO_CHECK_ARG(((uintptr_t)p % 16 == 0) && ((uintptr_t)g % 16 == 0) && ((uintptr_t)m % 16 == 0) &&
                   ((uintptr_t)v % 16 == 0));
    if (n == 0) return HALO_OK;
    AdamArgs a;
    a.p = p; a.g = g; a.m = m; a.v = v; a.n = n;
    // scalar prep in double like torch's python-side bias corrections (optim/adamw.py, _single_tensor)
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    a.decay_mul = (float)(1.0 - (double)lr * (double)weight_decay);
    a.beta1_w = (float)(1.0 - (double)beta1);
    a.beta2 = beta2;
    a.beta2_w = (float)(1.0 - (double)beta2);
    a.step_size = (float)((double)lr / bc1);
    a.bc2_sqrt = (float)sqrt(bc2);
    a.eps = eps;
    a.grad_scale = grad_scale;
    size_t blocks = (n / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    return halo_launch_status();
}

size_t halo_adamw_multi_tensor_bytes(void) { return sizeof(AdamTensor); }
unsigned halo_adamw_multi_chunk(void) { return ADAM_CHUNK; }

int halo_adamw_multi_max_tensors(void) { return ADAM_MULTI_MAX; }

int halo_adamw_multi(const void *tensor_table, const void *chunk_table, int n_chunks, const float *const *grads, int n_tensors, float lr,
                     float beta1, float beta2, float eps, int step, const float *grad_scale, halo_stream_t stream) {
    HALO_CHECK_ARG(tensor_table && chunk_table && n_chunks > 0 && step >= 1 && grads && n_tensors > 0 && n_tensors <= ADAM_MULTI_MAX);
    AdamGrads gr;
    for (int i = 0; i < ADAM_MULTI_MAX; ++i) gr.g[i] = i < n_tensors ? grads[i] : nullptr;
    for (int i = 0; i < n_tensors; ++i) HALO_CHECK_ARG(grads[i] != nullptr);
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(adamw_multi_kernel, dim3((unsigned)n_chunks), dim3(256), 0, (hipStream_t)stream, (const AdamTensor *)tensor_table,
                       (const uint2 *)chunk_table, gr, lr, (float)(1.0 - (double)beta1), beta2, (float)(1.0 - (double)beta2),
                       (float)((double)lr / bc1), (float)sqrt(bc2), eps, grad_scale);
    return halo_launch_status();
}

static int adamw_ranges_impl(float *p, const float *g, float *m, float *v, int n_ranges, const size_t *begin, const size_t *end,
                      const float *weight_decay, const float *const *grad_scale, float lr, float beta1, float beta2, float eps, int step,
                      const uint32_t *step_dev, const float *lr_dev, uint32_t *counter, halo_stream_t stream) {
    HALO_CHECK_ARG(p && g && m && v && begin && end && weight_decay && grad_scale && n_ranges > 0 && n_ranges <= ADAM_MAX_RANGES &&
                   (step >= 1 || step_dev) && (!lr_dev || step_dev));
    HALO_CHECK_ARG(((uintptr_t)p % 16 == 0) && ((uintptr_t)g % 16 == 0) && ((uintptr_t)m % 16 == 0) && ((uintptr_t)v % 16 == 0));
    AdamRangesArgs a;
    a.p = p; a.g = g; a.m = m; a.v = v; a.n_ranges = n_ranges; a.counter = counter;
    a.step_dev = step_dev; a.lr_dev = lr_dev; a.lr = lr; a.beta1 = beta1;
    if (step < 1) step = 1;
    size_t biggest = 0;
    for (int r = 0; r < n_ranges; ++r) {
        HALO_CHECK_ARG(begin[r] % 4 == 0 && end[r] % 4 == 0 && begin[r] <= end[r]);
        a.begin4[r] = begin[r] / 4; a.end4[r] = end[r] / 4;
        a.decay_mul[r] = (float)(1.0 - (double)lr * (double)weight_decay[r]);
        a.weight_decay[r] = weight_decay[r];
        a.grad_scale[r] = grad_scale[r];
        if (a.end4[r] - a.begin4[r] > biggest) biggest = a.end4[r] - a.begin4[r];
    }
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    a.beta1_w = (float)(1.0 - (double)beta1);
    a.beta2 = beta2;
    a.beta2_w = (float)(1.0 - (double)beta2);
    a.step_size = (float)((double)lr / bc1);
    a.bc2_sqrt = (float)sqrt(bc2);
    a.eps = eps;
    size_t blocks = (biggest + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    static int nt = -1;
    if (nt < 0) { const char *e = getenv("HALO_ADAMW_NT"); nt = (e && e[0] == '0') ? 0 : 1; }
    if (nt) hipLaunchKernelGGL(adamw_ranges_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(adamw_ranges_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    return halo_launch_status();
}

int halo_adamw_ranges(float *p, const float *g, float *m, float *v, int n_ranges, const size_t *begin, const size_t *end,
                      const float *weight_decay, const float *const *grad_scale, float lr, float beta1, float beta2, float eps, int step,
                      uint32_t *counter, halo_stream_t stream) {
    return adamw_ranges_impl(p, g, m, v, n_ranges, begin, end, weight_decay, grad_scale, lr, beta1, beta2, eps, step, nullptr, nullptr,
                             counter, stream);
}

int halo_adamw_ranges_dev(float *p, const float *g, float *m, float *v, int n_ranges, const size_t *begin, const size_t *end,
                          const float *weight_decay, const float *const *grad_scale, float lr, const float *lr_dev, float beta1, float beta2,
                          float eps, const uint32_t *step_dev, uint32_t *counter, halo_stream_t stream) {
    HALO_CHECK_ARG(step_dev);
    return adamw_ranges_impl(p, g, m, v, n_ranges, begin, end, weight_decay, grad_scale, lr, beta1, beta2, eps, 0, step_dev, lr_dev,
                             counter, stream);
}

}  // extern "C"
