// The tiled operand image of the split-bf16 GEMM (gemm_bf16x3.hip) and the block routine that writes it from a row-major fp32 source:
// shared with lstm.hip, whose weight-packing launch writes the layer-0 input projection's two images from its own tail blocks.
#pragma once
#include "halo_common.h"

namespace halo_img {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int TR = 128;          // rows per tile
constexpr int TK = 32;           // k per tile
constexpr int PART_BYTES = TR * TK * 2;      // 8192
constexpr int BLOCK_BYTES = 2 * PART_BYTES;  // hi + lo

__device__ __forceinline__ int swz_byte(int r, int c) { return r * 64 + ((c ^ ((r >> 2) & 3)) << 4); }

__device__ __forceinline__ void split8(const float (&x)[8], bf16x8 &hi, bf16x8 &lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 h = (__bf16)x[j];
        hi[j] = h;
        lo[j] = (__bf16)(x[j] - (float)h);
    }
}

// src row-major [R][K] (leading dimension ld): one workgroup writes one (rt, kt) block
__device__ __forceinline__ void prep_rowmajor_block(const float *__restrict__ src, int R, int K, int ld, char *__restrict__ img,
                                                    int KT, int with_lo, int kt, int rt) {
    char *blk = img + ((long)rt * KT + kt) * BLOCK_BYTES;
    const bool vec = (ld % 4 == 0) && ((uintptr_t)src % 16 == 0);
#pragma unroll
    for (int u = threadIdx.x; u < TR * 4; u += 256) {
        const int row = u >> 2, c = u & 3;
        const int gr = rt * TR + row, gk = kt * TK + c * 8;
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = 0.f;
        if (gr < R) {
            const float *p = src + (long)gr * ld + gk;
            if (vec && gk + 7 < K) {
                const f32x4 a = *reinterpret_cast<const f32x4 *>(p), b = *reinterpret_cast<const f32x4 *>(p + 4);
                x[0] = a[0]; x[1] = a[1]; x[2] = a[2]; x[3] = a[3]; x[4] = b[0]; x[5] = b[1]; x[6] = b[2]; x[7] = b[3];
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) if (gk + j < K) x[j] = p[j];
            }
        }
        bf16x8 hi, lo;
        split8(x, hi, lo);
        const int off = swz_byte(row, c);
        *reinterpret_cast<bf16x8 *>(blk + off) = hi;
        if (with_lo) *reinterpret_cast<bf16x8 *>(blk + PART_BYTES + off) = lo;   // HALO_MATH_BF16 never reads the lo part
    }
}

// src stored transposed: memory [K][R] (leading dimension ld, R contiguous); logical X[r][k] = src[k*ld + r]
__device__ __forceinline__ void prep_transposed_block(const float *__restrict__ src, int R, int K, int ld, char *__restrict__ img,
                                                      int KT, int with_lo, int kt, int rt, float (*tile)[TR + 1]) {
    char *blk = img + ((long)rt * KT + kt) * BLOCK_BYTES;
    const bool vec = (ld % 4 == 0) && ((uintptr_t)src % 16 == 0);
    for (int u = threadIdx.x; u < TK * (TR / 4); u += 256) {
        const int kk = u / (TR / 4), r4 = (u % (TR / 4)) * 4;
        const int gk = kt * TK + kk, gr = rt * TR + r4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (gk < K) {
            const float *p = src + (long)gk * ld + gr;
            if (vec && gr + 3 < R) v = *reinterpret_cast<const f32x4 *>(p);
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (gr + e < R) v[e] = p[e];
            }
        }
        tile[kk][r4] = v[0]; tile[kk][r4 + 1] = v[1]; tile[kk][r4 + 2] = v[2]; tile[kk][r4 + 3] = v[3];
    }
    __syncthreads();
    for (int u = threadIdx.x; u < TR * 4; u += 256) {
        const int row = u >> 2, c = u & 3;
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = tile[c * 8 + j][row];
        bf16x8 hi, lo;
        split8(x, hi, lo);
        const int off = swz_byte(row, c);
        *reinterpret_cast<bf16x8 *>(blk + off) = hi;
        if (with_lo) *reinterpret_cast<bf16x8 *>(blk + PART_BYTES + off) = lo;   // HALO_MATH_BF16 never reads the lo part
    }
}

}  // namespace halo_img
