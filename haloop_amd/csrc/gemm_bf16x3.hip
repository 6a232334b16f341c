// Split-bf16 ("bf16x3") GEMM for gfx950: fp32 operands are split once into hi = bf16(x) and
// lo = bf16(x - hi) and the product is accumulated in fp32 on the bf16 MFMA as
//     A*B ~= Ahi*Bhi + Ahi*Blo + Alo*Bhi          (dropped: Alo*Blo ~ 2^-16 |ab|)
// i.e. three v_mfma_f32_32x32x16_bf16 per K=16 instead of eight v_mfma_f32_32x32x2_f32: 5.3x the
// matrix rate of the exact-f32 path at ~2^-16 relative error per product (random sign).
//
// Operands are first rewritten ("prep") into TILED IMAGES: [row-tile(128)][k-tile(32)][hi|lo][8 KiB],
// each 8 KiB part being exactly the LDS image the MFMA fragment reads want (64-byte rows, the four
// 16-byte chunks of a row XOR-swizzled by (row>>2)&3 so that ds_read_b128 is bank-conflict free).
// The GEMM kernel therefore stages tiles with global_load_lds (16 B/lane, linear 1 KiB per wave
// instruction, no address math, no VALU conversion in the hot loop), zero-padded at the edges so
// the main loop has no bounds checks.  Prep also absorbs every transpose: all GEMMs become "NT".
#include "halo_common.h"
#include "halo_internal.h"

namespace {

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
__global__ __launch_bounds__(256) void prep_rowmajor_kernel(const float *__restrict__ src, int R, int K, int ld,
                                                            char *__restrict__ img, int KT) {
    const int kt = blockIdx.x, rt = blockIdx.y;
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
        *reinterpret_cast<bf16x8 *>(blk + PART_BYTES + off) = lo;
    }
}

// src stored transposed: memory [K][R] (leading dimension ld, R contiguous); logical X[r][k] = src[k*ld + r]
__global__ __launch_bounds__(256) void prep_transposed_kernel(const float *__restrict__ src, int R, int K, int ld,
                                                              char *__restrict__ img, int KT) {
    __shared__ float tile[TK][TR + 1];
    const int kt = blockIdx.x, rt = blockIdx.y;
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
        *reinterpret_cast<bf16x8 *>(blk + PART_BYTES + off) = lo;
    }
}

struct TiledGemmArgs {
    const char *A;     // image of A [M][K]
    const char *B;     // image of B [N][K]
    float *C;
    const float *bias1;
    const float *bias2;
    int M, N, KT;
    int ldc;
    int relu;
    int tiles_n;
    DropoutCfg drop;
    int use_drop;
    int ntiles, ksplit, ktper;   // split-K over k-tiles; raw sums of slice s go to slab[s][M][N]
    float *slab;
};

__device__ __forceinline__ void stage_block(const char *gblk, char *lds_dst, int wave, int lane) {
    // 16 KiB block = 16 wave-instructions of 1 KiB; each of the 4 waves issues 4
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int piece = i * 4 + wave;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gblk + piece * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void *)(lds_dst + piece * 1024), 16, 0, 0);
    }
}

__global__ __launch_bounds__(256) void gemm_bf16x3_kernel(const TiledGemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];       // 2 stages x (A block | B block) = 64 KiB
    const int tile = blockIdx.x % p.ntiles, kslice = blockIdx.x / p.ntiles;
    const int tile_m = tile / p.tiles_n, tile_n = tile % p.tiles_n;
    const int kt0 = kslice * p.ktper, kt1 = min(p.KT, kt0 + p.ktper);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, lh = lane >> 5;

    const char *Ablk = p.A + (long)tile_m * p.KT * BLOCK_BYTES;
    const char *Bblk = p.B + (long)tile_n * p.KT * BLOCK_BYTES;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment byte offsets inside a part: row = w*64 + i*32 + lr, chunk = ks*2 + lh
    int aoff[2][2], boff[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            aoff[i][ks] = swz_byte(wm * 64 + i * 32 + lr, ks * 2 + lh);
            boff[i][ks] = swz_byte(wn * 64 + i * 32 + lr, ks * 2 + lh);
        }

    stage_block(Ablk + (long)kt0 * BLOCK_BYTES, lds, wave, lane);
    stage_block(Bblk + (long)kt0 * BLOCK_BYTES, lds + BLOCK_BYTES, wave, lane);
    for (int t = kt0; t < kt1; ++t) {
        char *cur = lds + ((t - kt0) & 1) * 2 * BLOCK_BYTES;
        char *nxt = lds + ((t - kt0 + 1) & 1) * 2 * BLOCK_BYTES;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();            // tile t has landed for every wave; everyone is done with the other stage
        if (t + 1 < kt1) {
            stage_block(Ablk + (long)(t + 1) * BLOCK_BYTES, nxt, wave, lane);
            stage_block(Bblk + (long)(t + 1) * BLOCK_BYTES, nxt + BLOCK_BYTES, wave, lane);
        }
        const char *ah = cur, *al = cur + PART_BYTES, *bh = cur + BLOCK_BYTES, *bl = cur + BLOCK_BYTES + PART_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fah[2], fal[2], fbh[2], fbl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fah[i] = *reinterpret_cast<const bf16x8 *>(ah + aoff[i][ks]);
                fal[i] = *reinterpret_cast<const bf16x8 *>(al + aoff[i][ks]);
                fbh[i] = *reinterpret_cast<const bf16x8 *>(bh + boff[i][ks]);
                fbl[i] = *reinterpret_cast<const bf16x8 *>(bl + boff[i][ks]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fal[i], fbh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah[i], fbl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah[i], fbh[j], acc[i][j], 0, 0, 0);
                }
        }
    }

    // epilogue (C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5))
    const int m0 = tile_m * TR, n0 = tile_n * TR;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + lr;
            if (col >= p.N) continue;
            if (p.ksplit > 1) {
                float *slab = p.slab + (long)kslice * p.M * p.N;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (row < p.M) slab[(long)row * p.N + col] = acc[i][j][r];
                }
                continue;
            }
            float bias = 0.f;
            if (p.bias1) bias += p.bias1[col];
            if (p.bias2) bias += p.bias2[col];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row >= p.M) continue;
                float v = acc[i][j][r] + bias;
                if (p.relu) v = fmaxf(v, 0.f);
                const long e = (long)row * p.ldc + col;
                if (p.use_drop) v *= dropout_mult(p.drop, (uint64_t)e);
                p.C[e] = v;
            }
        }
    }
}

}  // namespace

size_t halo_tiled_image_bytes(int R, int K) {
    return (size_t)((R + TR - 1) / TR) * ((K + TK - 1) / TK) * BLOCK_BYTES;
}

int halo_prep_tiles(const float *src, int R, int K, int ld, int src_transposed, void *image, hipStream_t st) {
    const int KT = (K + TK - 1) / TK, RT = (R + TR - 1) / TR;
    if (src_transposed)
        hipLaunchKernelGGL(prep_transposed_kernel, dim3(KT, RT), dim3(256), 0, st, src, R, K, ld, (char *)image, KT);
    else
        hipLaunchKernelGGL(prep_rowmajor_kernel, dim3(KT, RT), dim3(256), 0, st, src, R, K, ld, (char *)image, KT);
    return halo_launch_status();
}

int halo_gemm_bf16x3_tiled(const void *Aimg, const void *Bimg, int M, int N, int K, float *C, int ldc,
                           const float *bias1, const float *bias2, int relu, const DropoutCfg *drop, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        // 64 KiB of dynamic LDS: opt in once (idempotent)
        if (hipFuncSetAttribute((const void *)gemm_bf16x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                4 * BLOCK_BYTES) != hipSuccess)
            return HALO_ELAUNCH;
        attr_set = true;
    }
    TiledGemmArgs p;
    p.A = (const char *)Aimg; p.B = (const char *)Bimg; p.C = C; p.bias1 = bias1; p.bias2 = bias2;
    p.M = M; p.N = N; p.KT = (K + TK - 1) / TK; p.ldc = ldc; p.relu = relu;
    p.tiles_n = (N + TR - 1) / TR;
    p.use_drop = drop && drop->threshold != 0u;
    if (drop) p.drop = *drop; else p.drop = make_dropout(0.f, 0, 0, 0, nullptr);
    p.ntiles = ((M + TR - 1) / TR) * p.tiles_n;
    p.ksplit = halo_pick_ksplit(p.ntiles, p.KT, (long)M * N);
    p.ktper = (p.KT + p.ksplit - 1) / p.ksplit;
    p.ksplit = (p.KT + p.ktper - 1) / p.ktper;
    void *scratch; size_t bytes;
    halo_get_scratch(&scratch, &bytes);
    p.slab = (float *)scratch;
    hipLaunchKernelGGL(gemm_bf16x3_kernel, dim3((unsigned)(p.ntiles * p.ksplit)), dim3(256), 4 * BLOCK_BYTES, st, p);
    int rc = halo_launch_status();
    if (rc != HALO_OK || p.ksplit == 1) return rc;
    return halo_splitk_reduce(p.slab, p.ksplit, M, N, C, ldc, bias1, bias2, relu, p.drop, p.use_drop, st);
}
