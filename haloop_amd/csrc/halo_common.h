// Shared device/host helpers for libhalo (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "halo.h"

#define HALO_WAVE 64

#define HALO_CHECK_ARG(cond) \
    do {                     \
        if (!(cond)) return HALO_EINVAL; \
    } while (0)

static inline int halo_launch_status() {
    return hipGetLastError() == hipSuccess ? HALO_OK : HALO_ELAUNCH;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- Philox4x32-10 (Random123 constants).  Must match oracle/philox.py bit for bit. -------------
struct Philox4 {
    uint32_t v[4];
};

__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                 uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32 x 32 -> 64 multiply per product (v_mad_u64_u32) instead of a v_mul_hi_u32 / v_mul_lo_u32 pair: both quarter rate
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0;
        const uint32_t n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    Philox4 o;
    o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
    return o;
}

struct DropoutCfg {
    uint32_t k0, k1;      // seed
    uint32_t stream_id;
    uint32_t offset;
    const uint32_t *offset_dev;   // optional device counter added to offset (graph replays advance it)
    uint32_t threshold;   // keep iff r >= threshold
    float scale;          // 1/(1-p); p<=0 -> disabled (threshold 0, scale 1)
};

static inline DropoutCfg make_dropout(float p, uint64_t seed, uint32_t stream_id, uint32_t offset,
                                      const uint32_t *offset_dev) {
    DropoutCfg d;
    d.offset_dev = p > 0.f ? offset_dev : nullptr;
    d.k0 = (uint32_t)(seed & 0xffffffffu);
    d.k1 = (uint32_t)(seed >> 32);
    d.stream_id = stream_id;
    d.offset = offset;
    if (p > 0.f) {
        double t = (double)p * 4294967296.0;
        d.threshold = t >= 4294967295.0 ? 0xffffffffu : (uint32_t)t;
        d.scale = 1.0f / (1.0f - p);
    } else {
        d.threshold = 0u;
        d.scale = 1.0f;
    }
    return d;
}

// mask multiplier for flat element index e
__device__ __forceinline__ float dropout_mult(const DropoutCfg &d, uint64_t e) {
    const uint64_t q = e >> 2;
    const uint32_t off = d.offset + (d.offset_dev ? *d.offset_dev : 0u);
    const Philox4 r = philox4x32_10((uint32_t)q, (uint32_t)(q >> 32), d.stream_id, off, d.k0, d.k1);
    const uint32_t lane = (uint32_t)(e & 3);
    const uint32_t v = lane == 0 ? r.v[0] : lane == 1 ? r.v[1] : lane == 2 ? r.v[2] : r.v[3];
    return v >= d.threshold ? d.scale : 0.f;
}

// four multipliers for elements 4q..4q+3 (e must be a multiple of 4)
__device__ __forceinline__ f32x4 dropout_mult4(const DropoutCfg &d, uint64_t e) {
    const uint64_t q = e >> 2;
    const uint32_t off = d.offset + (d.offset_dev ? *d.offset_dev : 0u);
    const Philox4 r = philox4x32_10((uint32_t)q, (uint32_t)(q >> 32), d.stream_id, off, d.k0, d.k1);
    f32x4 m;
    m[0] = r.v[0] >= d.threshold ? d.scale : 0.f;
    m[1] = r.v[1] >= d.threshold ? d.scale : 0.f;
    m[2] = r.v[2] >= d.threshold ? d.scale : 0.f;
    m[3] = r.v[3] >= d.threshold ? d.scale : 0.f;
    return m;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ float log_add_exp(float a, float b) {
    // torch.logaddexp semantics: equal infinities return themselves
    if (isinf(a) && a == b) return a;
    const float m = fmaxf(a, b);
    return m + log1pf(expf(-fabsf(a - b)));
}


// tanh for the GELU approximation on the hardware exp2 and reciprocal: with m = e^{-2|u|} in (0, 1], tanh|u| = (1 - m) / (1 + m) and
// 1 - tanh^2 = 4 m / (1 + m)^2 -- no overflow, and the tails (where the derivative multiplies 1 - tanh^2 by ~|x|^3) keep their
// relative accuracy.  Absolute error of tanh ~1e-7: the GELU it feeds is good to fp32 rounding of its value; the library tanhf costs
// ~5x the VALU work, which the GEMM epilogue pays per element ([8192 x 3072 x 768] bf16 with the GELU epilogue: 84 -> 67 us).
__device__ __forceinline__ float gelu_tanh(float u, float *sech2 = nullptr) {
    const float m = __builtin_amdgcn_exp2f(-2.8853900817779268f * fabsf(u)), r = __builtin_amdgcn_rcpf(1.0f + m);
    if (sech2) *sech2 = 4.0f * m * r * r;
    return copysignf((1.0f - m) * r, u);
}

// d gelu(x) / dx: kind 0 tanh-GELU (ha/attention.py:12-17), kind 1 exact (erf) GELU
__device__ __forceinline__ float gelu_grad(float x, int kind) {
    if (kind) return 0.5f * (1.0f + erff(x * 0.7071067811865476f)) + x * expf(-0.5f * x * x) * 0.3989422804014327f;
    float s2;
    const float k = 0.7978845608028654f, u = k * (x + 0.044715f * x * x * x), t = gelu_tanh(u, &s2);
    return 0.5f * (1.0f + t) + 0.5f * x * s2 * k * (1.0f + 3.0f * 0.044715f * x * x);
}

// All-reduce over the 16 lanes of a DPP row (lanes 16g .. 16g+15) on the VALU: row_ror:8,4,2,1 folds the row in four
// v_*_dpp instructions, no trip through the LDS crossbar that a ds_bpermute-based __shfl_xor takes.  Every lane of the
// row ends with the same bits (after the shift-s step the values are s-periodic, so each step adds one commutative pair).
template <int N>
__device__ __forceinline__ float row_ror(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x120 | N, 0xf, 0xf, false));
}
__device__ __forceinline__ float row16_sum(float v) {
    v += row_ror<8>(v); v += row_ror<4>(v); v += row_ror<2>(v); v += row_ror<1>(v);
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, row_ror<8>(v)); v = fmaxf(v, row_ror<4>(v)); v = fmaxf(v, row_ror<2>(v)); v = fmaxf(v, row_ror<1>(v));
    return v;
}

// all-reduce over the four 16-lane rows of a wave (lanes l, l^16, l^32, l^48).  v_permlane16_swap exchanges the odd
// rows of its first operand with the even rows of its second, v_permlane32_swap the upper half of the first with the
// lower half of the second; fed the same value twice they leave (x_r, x_r^1) resp. (x_lo, x_hi) in every lane.
// Issued as inline asm: through __builtin_amdgcn_permlane*_swap this compiler (ROCm 7.2) folds the second result into
// the first (the ISA showed max(s0, s0)).  The s_nop covers the VALU-write -> permlane-swap read hazard the compiler
// would otherwise schedule around.
__device__ __forceinline__ void swap_rows16(float &a, float &b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void swap_rows32(float &a, float &b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ float rows4_max(float x) {
    float a = x, b = x;
    swap_rows16(a, b);
    a = b = fmaxf(a, b);
    swap_rows32(a, b);
    return fmaxf(a, b);
}
__device__ __forceinline__ float rows4_sum(float x) {
    float a = x, b = x;
    swap_rows16(a, b);
    a = b = a + b;
    swap_rows32(a, b);
    return a + b;
}
// whole-wave sums / maxima without a trip through the LDS crossbar: 16 lanes on DPP, then the four rows by permlane swaps
__device__ __forceinline__ float wave_sum(float v) { return rows4_sum(row16_sum(v)); }
__device__ __forceinline__ float wave_max(float v) { return rows4_max(row16_max(v)); }

// GEMM epilogue flags (include/halo.h): bit 0 relu, bit 1 tanh-GELU (ha/attention.py:12-17), bit 2 C += result,
// bit 3 exact (erf) GELU (nn.GELU() / F.gelu: ha/transformer.py:456, ha/conv.py:46)
__device__ __forceinline__ float gemm_activation(float v, int flags) {
    if (flags & 1) v = fmaxf(v, 0.f);
    if (flags & 2) v = 0.5f * v * (1.0f + gelu_tanh(0.7978845608028654f * (v + 0.044715f * v * v * v)));
    if (flags & 8) v = 0.5f * v * (1.0f + erff(v * 0.7071067811865476f));
    return v;
}
