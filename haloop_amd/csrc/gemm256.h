// Single-pass bf16 product on 256 x 256 workgroup tiles over the tiled operand images (tiled_image.h: 128-row x 32-k blocks, hi part
// first), several products per launch.  C = A [M][K] x B [N][K]^T as plain fp32 sums (no bias / activation / dropout): the LSTM's
// weight-gradient products and the carried K-slices of its input gradient (lstm.hip) -- the two weight-gradient products of both layers
// are 208 tiles of 256 x 256 and run as ONE launch of one workgroup per CU where the 128 x 128-tile kernel ran two launches of 512 / 288.
//
// Why another kernel (cdna_hip_programming.md section 5, "the step-3 structure's ceiling"): gemm_bf16x3.hip's 128 x 128 tile gives a wave a
// 64 x 64 block -- every MFMA byte comes out of LDS once per 64 columns, the LDS pipe (128 B/clk/CU) is as busy as the matrix cores, and
// a k-tile of 32 puts a barrier behind every 8 MFMAs.  Here a wave owns 128 x 64 (acc: 128 VGPRs), eight waves a 256 x 256 tile:
//   * LDS ring of four k-blocks (32 k each): [A rows 0-127 | A rows 128-255 | B rows 0-127 | B rows 128-255] x 8 KiB, filled by LDS-DMA
//     (global_load_lds, 16 B per lane) straight from the image blocks -- the image IS the swizzled LDS layout;
//   * a k-block is two PHASES of 16 k: {6 ds_read_b128 || 2 LDS-DMA issues -> barrier -> 8 MFMA 32x32x16 -> barrier};
//   * the two wave groups (rows 0-127 / 128-255 of the tile; one wave of each per SIMD) run ONE BARRIER APART: while one group's MFMAs
//     run, the other reads its fragments and issues the prefetch, so the matrix pipe of every SIMD always has a wave on it;
//   * the prefetch stays in flight across barriers: counted s_waitcnt vmcnt(6) once per k-block, raw s_barrier, never vmcnt(0) in the loop.
// Schedule (phase f = 2 j + p of k-block j; group 1 runs every section one barrier later than group 0):
//   phase (j, 0): read k16 #0 of slot j % 4;  issue the B pieces of k-block j + 2 -> slot (j + 2) % 4
//   phase (j, 1): read k16 #1;                issue the A pieces of k-block j + 3 -> slot (j + 3) % 4;  s_waitcnt vmcnt(6)
// RAW: the wait in phase (j, 1) retires this wave's loads up to B(j + 1) (behind it: A(j + 2), B(j + 2), A(j + 3) = 6 loads); it stands in
// front of the phase's FIRST barrier, so the slower group has waited too before the faster group's first read of k-block j + 1, which
// follows that phase's second barrier.  WAR: slot (j + 3) % 4 last held k-block j - 1, whose last reads (phase (j - 1, 1)) both groups
// retired with the lgkmcnt(0) in front of their MFMAs, two barriers before the first issue into it.
// Past the last k-block the issues re-read it into slots nobody reads again, so every phase issues exactly two loads and the count holds.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "halo_internal.h"

namespace halo_g256 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int PART = 8192;               // one 128-row x 32-k part
constexpr int BLOCK = 16384;             // image block: hi part | lo part
constexpr int SLOT = 4 * PART;           // [A0 | A1 | B0 | B1]
constexpr int NSLOT = 4;
constexpr int LDS_BYTES = NSLOT * SLOT;  // 128 KiB
constexpr int MAXP = 3;

struct Prob {
    const char *A, *B;                   // tiled images of A [M][K], B [N][K]
    int M, N, KT;                        // KT: 32-deep k-blocks
    int kslices;                         // K split into this many slices (>= 1); slice s goes to C + s * slab_stride
    long slab_stride;
    float *C, *C2;                       // columns [0, n_split) -> C (ldc), [n_split, N) -> C2 (ldc2) from its column 0; n_split % 256 == 0 or == N
    int ldc, ldc2, n_split;
    float *sumsq;                        // optional: workgroup t of this problem writes the sum of the squares of what it stored to sumsq[t]
    // filled by finish():
    int rbA, rbB, tiles_m, tiles_n, ktper, first;
};
struct Args {
    Prob p[MAXP];
    int nprob;
};

static inline void finish(Prob &g, int first) {
    g.rbA = (g.M + 127) / 128; g.rbB = (g.N + 127) / 128;
    g.tiles_m = (g.M + 255) / 256; g.tiles_n = (g.N + 255) / 256;
    if (g.kslices < 1) g.kslices = 1;
    g.ktper = (g.KT + g.kslices - 1) / g.kslices;
    g.kslices = (g.KT + g.ktper - 1) / g.ktper;
    g.first = first;
}

__device__ __forceinline__ int swz(int r, int c) { return r * 64 + ((c ^ ((r >> 2) & 3)) << 4); }

__device__ __forceinline__ void dma16(const char *src, char *lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src, (__attribute__((address_space(3))) void *)lds_dst, 16, 0, 0);
}

// position of tile t (of n) in the XCD-contiguous order: consecutive workgroups go to consecutive XCDs, so the tiles t, t + 8, t + 16, ...
// of a problem share an L2; they get the contiguous run number t % 8 (of n / 8 or n / 8 + 1 tiles), in order.  Bijective for any n.
__device__ __forceinline__ int xcd_order(int t, int n) {
    const int rank = t & 7, k = t >> 3, q = n >> 3, r = n & 7;
    return (rank < r ? rank * (q + 1) : r * (q + 1) + (rank - r) * q) + k;
}

// LAB: 0 the product; measurement variants of tools/gemm256_lab.hip (wrong results): 1 no prefetch issues in the loop, 2 no counted wait,
// 4 no fragment reads in the loop, 8 no epilogue stores
template <int LAB = 0>
__global__ __launch_bounds__(512) void gemm256_kernel(const Args a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int wr = wave >> 2, wc = wave & 3;
    const int lr = lane & 31, lh = lane >> 5;
    // which problem, slice and tile
    int q = 0;
    if (a.nprob > 1 && (int)blockIdx.x >= a.p[1].first) q = 1;
    if (a.nprob > 2 && (int)blockIdx.x >= a.p[2].first) q = 2;
    const Prob &g = a.p[q];
    const int local = (int)blockIdx.x - g.first, ntile = g.tiles_m * g.tiles_n;
    const int kslice = local / ntile;
    // tile order: workgroups are dealt round-robin over the 8 XCDs (block index % 8 labels the L2 they share), so each XCD gets a CONTIGUOUS
    // run of tiles, walked in groups of four tile rows column by column -- the ~16-32 tiles resident on one XCD then cover a compact block
    // (4 x 4: 8 operand panels through that L2) instead of one tile column (17 panels).  Bijective for any count; speed only.
    const int tile = xcd_order(local % ntile, ntile);
    const int grp = tile / (4 * g.tiles_n), gm0 = grp * 4, gh = min(4, g.tiles_m - gm0), ing = tile % (4 * g.tiles_n);
    const int tile_m = gm0 + ing % gh, tile_n = ing / gh;
    const int KT = g.KT;
    const int kt0 = kslice * g.ktper, nkb = min(KT, kt0 + g.ktper) - kt0;
    // the four image rows of blocks this tile reads (a row block past the operand re-reads its last one: those products are not stored)
    const char *srcA0 = g.A + ((long)min(2 * tile_m, g.rbA - 1) * KT + kt0) * BLOCK, *srcA1 = g.A + ((long)min(2 * tile_m + 1, g.rbA - 1) * KT + kt0) * BLOCK;
    const char *srcB0 = g.B + ((long)min(2 * tile_n, g.rbB - 1) * KT + kt0) * BLOCK, *srcB1 = g.B + ((long)min(2 * tile_n + 1, g.rbB - 1) * KT + kt0) * BLOCK;
    // LDS-DMA: a pair of parts (16 KiB) = 16 wave-instructions of 1 KiB; wave w issues pieces w (first part) and w + 8 (second part)
    const int dma_off = wave * 1024 + lane * 16;
    auto issue_A = [&](int kb, int slot) {
        const long o = (long)min(kb, nkb - 1) * BLOCK + dma_off;
        dma16(srcA0 + o, lds + slot * SLOT + wave * 1024);
        dma16(srcA1 + o, lds + slot * SLOT + PART + wave * 1024);
    };
    auto issue_B = [&](int kb, int slot) {
        const long o = (long)min(kb, nkb - 1) * BLOCK + dma_off;
        dma16(srcB0 + o, lds + slot * SLOT + 2 * PART + wave * 1024);
        dma16(srcB1 + o, lds + slot * SLOT + 3 * PART + wave * 1024);
    };
    // fragment offsets inside a slot: A part wr, rows 32 i + lr; B part 2 + (wc >> 1), rows 64 (wc & 1) + 32 jn + lr; chunk 2 p + lh
    int aoff[2][4], boff[2][2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int i = 0; i < 4; ++i) aoff[p][i] = wr * PART + swz(32 * i + lr, 2 * p + lh);
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) boff[p][jn] = (2 + (wc >> 1)) * PART + swz(64 * (wc & 1) + 32 * jn + lr, 2 * p + lh);
    }
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jn][r] = 0.f;

    // prologue: A(0) B(0) A(1) B(1) A(2); k-block 0 has landed when all but the newest three issues (6 loads) have
    issue_A(0, 0); issue_B(0, 0); issue_A(1, 1); issue_B(1, 1); issue_A(2, 2);
    asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
    if (wr == 1) __builtin_amdgcn_s_barrier();             // the second group runs one barrier behind the first
    for (int j = 0; j < nkb; ++j) {
        const char *cur = lds + (j & 3) * SLOT;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            bf16x8 fa[4], fb[2];
            if (!(LAB & 4) || j == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const bf16x8 *>(cur + aoff[p][i]);
#pragma unroll
                for (int jn = 0; jn < 2; ++jn) fb[jn] = *reinterpret_cast<const bf16x8 *>(cur + boff[p][jn]);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("" : "=v"(fa[i]));
#pragma unroll
                for (int jn = 0; jn < 2; ++jn) asm volatile("" : "=v"(fb[jn]));
            }
            if (!(LAB & 1)) {
                if (p == 0) issue_B(j + 2, (j + 2) & 3);
                else issue_A(j + 3, (j + 3) & 3);
            }
            if (p == 1 && !(LAB & 3)) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jn = 0; jn < 2; ++jn) acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[jn], acc[i][jn], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
        }
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();             // (every wave has now passed the same number of barriers)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the clamped tail loads: drained before the workgroup retires

    // epilogue: element (i, jn, r) of this lane is row 256 tile_m + 128 wr + 32 i + (r & 3) + 8 (r >> 2) + 4 lh, column 256 tile_n + 64 wc + 32 jn + lr.
    // Buffer stores: the lane part of the address (4 lh rows down, the lane's column) is computed once per column block, the element's row
    // travels as the scalar offset; a lane whose column (or, in a tile that hangs over the last row, whose row) lies outside the matrix
    // addresses far beyond the descriptor's range and its store is dropped -- no branch per element.
    const int m0 = tile_m * 256 + wr * 128, n0 = tile_n * 256 + wc * 64;
    const bool second = tile_n * 256 >= g.n_split;          // workgroup-uniform: n_split is a multiple of 256 (or N)
    float *cq = (second ? g.C2 : g.C) + (long)kslice * g.slab_stride;
    const int ldq = second ? g.ldc2 : g.ldc, cshift = second ? g.n_split : 0;
    const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(cq, 0, 0x40000000, 0x00020000);
    constexpr int FAR = 0x7ffffff0;
    float ssj[2] = {0.f, 0.f};
    bool colok[2];
    int voff[2];
#pragma unroll
    for (int jn = 0; jn < 2; ++jn) {
        const int col = n0 + 32 * jn + lr;
        colok[jn] = col < g.N;
        voff[jn] = colok[jn] ? (4 * lh * ldq + col - cshift) * 4 : FAR;
    }
    if (m0 + 128 <= g.M) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int soff = (m0 + 32 * i + (r & 3) + 8 * (r >> 2)) * ldq * 4;
#pragma unroll
                for (int jn = 0; jn < 2; ++jn) {
                    const float v = acc[i][jn][r];
                    if (!(LAB & 8)) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), crs, voff[jn], soff, 0);
                    ssj[jn] += v * v;
                }
            }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + 32 * i + (r & 3) + 8 * (r >> 2);
                const int soff = row * ldq * 4;
                const bool rowok = row + 4 * lh < g.M;
#pragma unroll
                for (int jn = 0; jn < 2; ++jn) {
                    const float v = acc[i][jn][r];
                    if (!(LAB & 8)) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), crs, rowok ? voff[jn] : FAR, soff, 0);
                    ssj[jn] += rowok ? v * v : 0.f;
                }
            }
    }
    float ss = (colok[0] ? ssj[0] : 0.f) + (colok[1] ? ssj[1] : 0.f);
    if (g.sumsq) {                                          // (workgroup-uniform; the ring is free: every wave is past its last read)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) ss += __shfl_xor(ss, d, 64);
        __syncthreads();
        float *red = reinterpret_cast<float *>(lds);
        if (lane == 0) red[wave] = ss;
        __syncthreads();
        if (threadIdx.x == 0) g.sumsq[local] = ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]));
    }
}

template <int LAB = 0>
static inline hipError_t launch(const Args &a, int nwg, hipStream_t st) {
    if (!halo_func_attr_done(1 + (LAB != 0))) {         // per device (halo_internal.h)
        const hipError_t e = hipFuncSetAttribute((const void *)gemm256_kernel<LAB>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return e;
        halo_func_attr_set(1 + (LAB != 0));
    }
    hipLaunchKernelGGL(gemm256_kernel<LAB>, dim3((unsigned)nwg), dim3(512), LDS_BYTES, st, a);
    return hipGetLastError();
}

}  // namespace halo_g256
