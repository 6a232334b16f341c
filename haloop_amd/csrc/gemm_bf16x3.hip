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
#include <stdlib.h>
#include "halo_common.h"
#include "halo_internal.h"
#include "tiled_image.h"

namespace {

using namespace halo_img;

__global__ __launch_bounds__(256) void prep_rowmajor_kernel(const float *__restrict__ src, int R, int K, int ld,
                                                            char *__restrict__ img, int KT, int with_lo) {
    prep_rowmajor_block(src, R, K, ld, img, KT, with_lo, blockIdx.x, blockIdx.y);
}

__global__ __launch_bounds__(256) void prep_transposed_kernel(const float *__restrict__ src, int R, int K, int ld,
                                                              char *__restrict__ img, int KT, int with_lo) {
    __shared__ float tile[TK][TR + 1];
    prep_transposed_block(src, R, K, ld, img, KT, with_lo, blockIdx.x, blockIdx.y, tile);
}

// One read of a [R][C] fp32 source (optionally through an elementwise operator) -> BOTH operand images a Linear's backward wants:
// the row-major image (rows R, k = C: the A operand of dx = dy W) and the transposed one (rows C, k = R: the A / B operand of
// dW = dy^T x).  A workgroup owns a 128 x 32 source tile: it is one whole block of the row-major image and a 32-row quarter of
// four blocks of the transposed image (tile rows 32q..32q+31 are k-tile 4*rt + q), turned through LDS.
enum PairOp { PAIR_COPY = 0, PAIR_GELU_TANH = 1, PAIR_GELU_ERF = 2, PAIR_GELU_TANH_BWD = 3, PAIR_GELU_ERF_BWD = 4, PAIR_CE_BWD = 5 };

struct PairArgs {
    const float *src, *src2;       // src2: the pre-activation of the GELU backward (src = dy)
    long ld, ld2;
    int R, C;
    char *img_rm, *img_tr;         // either may be NULL
    int KT_rm, KT_tr, with_lo;
    // PAIR_CE_BWD: src = logits [R][C]; value = (softmax - onehot(target)) * grad, 0 on ignored rows
    const int64_t *target;
    const float *lse, *grad;
    long grad_stride, ignore_index;
};

template <int OP>
__device__ __forceinline__ void image_pair_block(const PairArgs &p, int kt, int rt, float (*tile)[TK + 1]) {
    const bool vec = (p.ld % 4 == 0) && ((uintptr_t)p.src % 16 == 0) &&
                     (!(OP == PAIR_GELU_TANH_BWD || OP == PAIR_GELU_ERF_BWD) || ((p.ld2 % 4 == 0) && ((uintptr_t)p.src2 % 16 == 0)));
    const bool write_rm = p.img_rm && kt < p.KT_rm;
    char *blk = p.img_rm + ((long)rt * p.KT_rm + kt) * BLOCK_BYTES;
#pragma unroll
    for (int u = threadIdx.x; u < TR * 4; u += 256) {
        const int row = u >> 2, c = u & 3;
        const int gr = rt * TR + row, gk = kt * TK + c * 8;
        float x[8], y[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { x[j] = 0.f; y[j] = 0.f; }
        if (gr < p.R && gk < p.C) {
            const float *s = p.src + (long)gr * p.ld + gk;
            const float *s2 = (OP == PAIR_GELU_TANH_BWD || OP == PAIR_GELU_ERF_BWD) ? p.src2 + (long)gr * p.ld2 + gk : nullptr;
            if (vec && gk + 7 < p.C) {
                const f32x4 a = *reinterpret_cast<const f32x4 *>(s), b = *reinterpret_cast<const f32x4 *>(s + 4);
                x[0] = a[0]; x[1] = a[1]; x[2] = a[2]; x[3] = a[3]; x[4] = b[0]; x[5] = b[1]; x[6] = b[2]; x[7] = b[3];
                if (s2) {
                    const f32x4 e = *reinterpret_cast<const f32x4 *>(s2), f = *reinterpret_cast<const f32x4 *>(s2 + 4);
                    y[0] = e[0]; y[1] = e[1]; y[2] = e[2]; y[3] = e[3]; y[4] = f[0]; y[5] = f[1]; y[6] = f[2]; y[7] = f[3];
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (gk + j < p.C) { x[j] = s[j]; if (s2) y[j] = s2[j]; }
            }
            if (OP == PAIR_GELU_TANH || OP == PAIR_GELU_ERF) {
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = gemm_activation(x[j], OP == PAIR_GELU_ERF ? 8 : 2);       // gelu(0) = 0: the padding stays 0
            } else if (OP == PAIR_GELU_TANH_BWD || OP == PAIR_GELU_ERF_BWD) {
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = x[j] * gelu_grad(y[j], OP == PAIR_GELU_ERF_BWD);
            } else if (OP == PAIR_CE_BWD) {
                const long tgt = p.target[gr];
                const float l = p.lse[gr], g = p.grad[(long)gr * p.grad_stride];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    x[j] = (tgt == p.ignore_index || gk + j >= p.C) ? 0.f : (expf(x[j] - l) - (gk + j == tgt ? 1.0f : 0.f)) * g;
            }
        }
        if (write_rm) {
            bf16x8 hi, lo;
            split8(x, hi, lo);
            const int off = swz_byte(row, c);
            *reinterpret_cast<bf16x8 *>(blk + off) = hi;
            if (p.with_lo) *reinterpret_cast<bf16x8 *>(blk + PART_BYTES + off) = lo;
        }
        if (p.img_tr) {
#pragma unroll
            for (int j = 0; j < 8; ++j) tile[row][c * 8 + j] = x[j];
        }
    }
    if (!p.img_tr) return;
    __syncthreads();
    // transposed image: its rows are the source columns kt*32 + cc, its k runs over the source rows
#pragma unroll
    for (int u = threadIdx.x; u < 4 * TK * 4; u += 256) {
        const int c = u & 3, cc = (u >> 2) & 31, q = u >> 7;
        const int tkt = rt * 4 + q;
        if (tkt >= p.KT_tr) continue;
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = tile[q * 32 + c * 8 + j][cc];
        bf16x8 hi, lo;
        split8(x, hi, lo);
        char *tb = p.img_tr + ((long)(kt >> 2) * p.KT_tr + tkt) * BLOCK_BYTES;
        const int off = swz_byte((kt & 3) * 32 + cc, c);
        *reinterpret_cast<bf16x8 *>(tb + off) = hi;
        if (p.with_lo) *reinterpret_cast<bf16x8 *>(tb + PART_BYTES + off) = lo;
    }
}

template <int OP>
__global__ __launch_bounds__(256) void image_pair_kernel(const PairArgs p) {
    __shared__ float tile[TR][TK + 1];
    image_pair_block<OP>(p, blockIdx.x, blockIdx.y, tile);
}

// Several operand images in ONE launch (the LSTM builds 2-4 per layer and direction, each a ~5 us kernel behind its own ~1.5 us
// launch boundary): block b belongs to the job whose [first, first + gx*gy) range holds it.
constexpr int PREP_MAX_JOBS = 6;
struct PrepJobs {
    PairArgs job[PREP_MAX_JOBS];      // kind 0: row-major source -> img_rm; 1: transposed source -> img_rm; 2: pair (copy) -> img_rm + img_tr;
                                      // 3: fp32 column sums of src [R][C] -> (float *)img_rm and (float *)img_tr
    int kind[PREP_MAX_JOBS], first[PREP_MAX_JOBS], gx[PREP_MAX_JOBS];
    int n;
};
__global__ __launch_bounds__(256) void prep_jobs_kernel(const PrepJobs a) {
    __shared__ float tile[TR * (TK + 1)];           // >= TK * (TR + 1) as well
    int j = 0;
#pragma unroll
    for (int i = 1; i < PREP_MAX_JOBS; ++i)
        if (i < a.n && (int)blockIdx.x >= a.first[i]) j = i;
    const PairArgs &p = a.job[j];
    const int local = blockIdx.x - a.first[j];
    const int kt = local % a.gx[j], rt = local / a.gx[j];
    if (a.kind[j] == 0) prep_rowmajor_block(p.src, p.R, p.C, (int)p.ld, p.img_rm, p.KT_rm, p.with_lo, kt, rt);
    else if (a.kind[j] == 1) prep_transposed_block(p.src, p.R, p.C, (int)p.ld, p.img_rm, p.KT_rm, p.with_lo, kt, rt, reinterpret_cast<float (*)[TR + 1]>(tile));
    else if (a.kind[j] == 2) image_pair_block<PAIR_COPY>(p, kt, rt, reinterpret_cast<float (*)[TK + 1]>(tile));
    else {      // kind 3: out[c] (and out2[c]) = sum over the R rows of src [R][C], rows in order: the batch-tile partials of a bias gradient
        const int c = local * 256 + threadIdx.x;
        if (c < p.C) {
            float s = 0.f;
            for (int r = 0; r < p.R; ++r) s += p.src[(long)r * p.ld + c];
            reinterpret_cast<float *>(p.img_rm)[c] = s;
            if (p.img_tr) reinterpret_cast<float *>(p.img_tr)[c] = s;
        }
    }
}

struct TiledGemmArgs {
    const char *A;     // image of A [M][K]
    const char *B;     // image of B [N][K]
    float *C;
    const float *R;    // HALO_GEMM_ACCUM: the addend [M][ldr] (C itself for C += result)
    int ldr;
    // IO & 1: A is not an image but row-major bf16 [M][lda] (hi, and lo with three passes), K % 32 == 0
    const __bf16 *Arm_hi, *Arm_lo;
    long lda;
    // IO & 8 ("TN"): BOTH operands are row-major bf16 with the contraction index as their ROW: A = Arm_hi [K][M] (lda), B = Brm [K][N]
    // (ldb); C [M][N] = A^T B.  The weight gradient of a Linear straight from the row-major activations and output gradients: no
    // transposed operand image is built (K % 32 == 0, M % 8 == 0, N % 8 == 0; single pass only)
    const __bf16 *Brm;
    long ldb;
    // IO & 2: the result also (C != NULL) or only (C == NULL) goes out as row-major bf16 [M][ldo]: hi = bf16(v), lo = bf16(v - hi)
    __bf16 *Ohi, *Olo;
    long ldo;
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
    // fused cross-entropy epilogue (lm_head): per row and 64-column strip the (max, sum exp) of the logits go to
    // ce_part[row][strip][2], the target's logit to ce_tlogit[row]; C may then be NULL (the logits are never written)
    const int64_t *ce_target;
    float *ce_part, *ce_tlogit;
    int ce_strips;               // strips per row = 2 * tiles_n
    // IO & 4: the result's columns [n_split, N) go to a second matrix C2 (leading dimension ldc2) from its column 0 on; n_split is a
    // multiple of the 128-column tile.  Two products that share their A operand run as ONE launch over the stacked B operands.
    float *C2;
    int n_split, ldc2;
    // ... and such a launch can CARRY a second, independent product in the workgroups behind its own (block index >= rider_first > 0):
    // K-slices of rA [rM][K'] x rB [rN][K']^T, slice s to rslab + s rM rN as plain sums (halo_gemm_bf16x3_tiled_slices' product).  A
    // 512-tile product leaves every CU's third workgroup slot free; 100-200 more workgroups of a latency-bound product ride there.
    const char *rA, *rB;
    float *rslab;
    int rM, rN, rKT, r_ntiles, r_tiles_n, r_ktper, rider_first;
    // IO & 4: sumsq_part != NULL -> workgroup b of the main product also writes the sum of the squares of the elements it stored to
    // sumsq_part[b] (fixed order inside the workgroup): the squared gradient norm's partials from the registers that hold the gradient,
    // instead of a pass that re-reads it
    float *sumsq_part;
};

template <int PASSES, int ITERS = (PASSES == 3 ? 4 : 2)>
__device__ __forceinline__ void stage_block(const char *gblk, char *lds_dst, int wave, int lane) {
    // 16 KiB block = 16 wave-instructions of 1 KiB; each of the 4 waves issues 4 (PASSES == 1: only the hi part, 2 each; ITERS = 1: the
    // 64 rows of a half block)
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
        const int piece = i * 4 + wave;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gblk + piece * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void *)(lds_dst + piece * 1024), 16, 0, 0);
    }
}

// The same LDS slot filled from a ROW-MAJOR bf16 matrix instead of a tiled image.  LDS-DMA puts lane l's 16 bytes at l * 16 of the
// piece, i.e. at (row = 16 piece + l / 4, position l % 4) of the slot's 64-byte rows; the swizzled image holds logical chunk
// pos ^ ((row >> 2) & 3) there -- so the lane simply FETCHES that chunk: the swizzle is applied on the source side and no operand-image
// pass over the activations is needed.  Rows past M read row M - 1 (their products are never stored).
template <int PASSES, int ITERS = (PASSES == 3 ? 4 : 2)>
__device__ __forceinline__ void stage_rowmajor(const __bf16 *hi, const __bf16 *lo, long lda, int row0, int k0, int M, char *lds_dst, int wave,
                                               int lane) {
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
        const int piece = i * 4 + wave;                        // 0..7: hi part, 8..15: lo part (as stage_block)
        const int row = (piece & 7) * 16 + (lane >> 2), chunk = (lane & 3) ^ ((row >> 2) & 3);
        const __bf16 *src = (piece < 8 ? hi : lo) + (long)min(row0 + row, M - 1) * lda + k0 + chunk * 8;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(lds_dst + piece * 1024), 16, 0, 0);
    }
}

// TN operands (IO & 8): a k-tile of a row-major bf16 matrix X [K][cols] -> 32 k-rows of 256 B (this tile's 128 columns) in the slot part.
// Row t keeps its 16-byte chunk c (columns 8c .. 8c+7) at position (c + 4 (t & 3)) & 15: the four rows a transposed read
// (ds_read_b64_tr_b16) gathers for one fragment then lie on disjoint banks.  As in stage_rowmajor the permutation is applied on the
// SOURCE side (LDS-DMA writes lane l at l * 16 of its 1 KiB piece = row 4 piece + l / 16, position l % 16).  Columns past the matrix
// read its last eight (their products are never stored).
__device__ __forceinline__ void stage_kmajor(const __bf16 *x, long ld, int col0, int k0, int ncols, char *lds_dst, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int piece = i * 4 + wave;                        // 0..7
        const int t = piece * 4 + (lane >> 4), c = ((lane & 15) - 4 * (t & 3)) & 15;
        const __bf16 *src = x + (long)(k0 + t) * ld + min(col0 + c * 8, ncols - 8);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(lds_dst + piece * 1024), 16, 0, 0);
    }
}
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) bf16x4_t lds_bf16x4_t;
// the 32x32x16 operand fragment (lane: row = lane & 31, k = 8 (lane >> 5) .. + 7) of such a part: two transposed reads of 4 k-rows x 16
// columns per 16-lane group (lane 4q + p of a group addresses row q, columns 4p .. 4p+3 and receives column lane & 15 of the four rows)
// Issued as inline assembly: behind the builtin the compiler drains every pending LDS-DMA (s_waitcnt vmcnt(0)) before the read, because
// it cannot tell the ring slot being read from the one being filled, and the loop loses its pipelining (measured: 1.4x the time).  The
// caller waits for the reads itself (tr_wait) before the first MFMA that takes them.
__device__ __forceinline__ void tr_read2(bf16x4_t &x0, bf16x4_t &x1, unsigned lds_addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:1024" : "=&v"(x0), "=&v"(x1) : "v"(lds_addr) : "memory");
}
__device__ __forceinline__ bf16x8 join8(bf16x4_t x0, bf16x4_t x1) { return bf16x8{x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]}; }
// every LDS read of the wave has returned; the fragments pass through so that their MFMAs are ordered behind the wait
__device__ __forceinline__ void tr_wait(bf16x8 &a0, bf16x8 &a1, bf16x8 &a2, bf16x8 &a3, bf16x8 &b0, bf16x8 &b1, bf16x8 &b2, bf16x8 &b3) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : : "memory");
}

constexpr int STAGE_BYTES = 2 * BLOCK_BYTES;    // one ring slot: A block | B block = 32 KiB
constexpr int LOADS_PER_STAGE = 8;              // global_load_lds per thread and stage (hi + lo; half of it when only hi is staged)

template <int N>
__device__ __forceinline__ void wait_vm_and_barrier() {
    // counted wait (the newest N LDS-DMA loads of this wave may stay in flight), then a raw barrier:
    // __syncthreads() would drain vmcnt(0) because LDS-DMA counts as a pending LDS write
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

// XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 labels the
// blocks that share an L2), so give each XCD a CONTIGUOUS run of output tiles (a few tile rows x all
// tile columns): its L2 then holds the A panels it needs instead of all of A.  Bijective for any
// count (cdna_hip_programming.md T1); placement only changes speed, never results.
__device__ __forceinline__ int xcd_remap(int bid, int n) {
    const int q = n / 8, r = n % 8, xcd = bid % 8, k = bid / 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// PASSES == 3: split-bf16 (hi*hi + hi*lo + lo*hi);  PASSES == 1: plain bf16 operands (hi parts only: HALO_MATH_BF16)
// CE: the cross-entropy epilogue is compiled into its own instantiations, so the plain products keep the register count and
// the epilogue code they had without it
// EPI: 0 the full epilogue; 1 bias only; 3 bias + residual add (neither: no activation, no dropout, no split-K); 2 a split-K slice (raw sums to its slab).
// The plain products run on 1 / 2: the code they do not need costs them 3-6 % when it is compiled in.
// IO: bit 0 = A staged from row-major bf16 (stage_rowmajor), bit 1 = bf16 row-major output; both only in their own instantiations
// TI: 32-row blocks per wave = 2 (a 128 x 128 workgroup tile) or 1 (64 x 128: the single-pass, whole-K products whose 128-row tiles would
// number 257 .. 511 -- one or two workgroups on a CU that holds three -- run on twice as many half-height tiles, three per CU)
// (the body as a device function of the arguments and the workgroup's index: gemm_bf16x3_kernel runs it on blockIdx.x, the grouped TN
//  launch -- several products in one grid -- on the index inside the product the workgroup belongs to)
template <int NSTAGE, int PASSES, bool CE, int EPI, int IO, int TI>
__device__ __forceinline__ void gemm_tile(const TiledGemmArgs &p, const int block) {
    static_assert(TI == 2 || (TI == 1 && PASSES == 1 && !CE && !(IO & 8)), "64-row tiles: single pass, plain epilogues");
    constexpr int TRM = 64 * TI;
    constexpr int LOADS = PASSES == 3 ? LOADS_PER_STAGE : (TI == 2 ? LOADS_PER_STAGE / 2 : 3);
    // ring slot: [A block | B block]; with one pass only the hi parts are staged, so a slot is half the size and the
    // same LDS holds a ring twice as deep (the single-pass loop is bound by LDS-DMA latency, not by the MFMAs)
    constexpr int OPERB = PASSES == 3 ? BLOCK_BYTES : PART_BYTES, OPER = OPERB * TI / 2, SLOT = OPER + OPERB;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    // a carried product's workgroup (IO & 4, TI == 2 launches only) works on the rider's operands and tile grid
    const bool rider = (IO & 4) && TI == 2 && p.rider_first > 0 && block >= p.rider_first;
    const int bid = rider ? block - p.rider_first : block;
    const int ntiles = rider ? p.r_ntiles : p.ntiles, tiles_n = rider ? p.r_tiles_n : p.tiles_n, KT = rider ? p.rKT : p.KT;
    const int ktper = rider ? p.r_ktper : p.ktper;
    const char *Aimg = rider ? p.rA : p.A, *Bimg = rider ? p.rB : p.B;
    const int kslice = bid / ntiles;
    // within an XCD's contiguous run, walk the tiles in groups of 8 tile rows, column by column: the
    // ~64 workgroups resident on one XCD then cover an 8x8 block (8 A panels + 8 B panels, in k
    // lockstep) instead of 2 x 32, so each staged k-tile is fetched from beyond L2 once, not 4 times
    const int tile = xcd_remap(bid % ntiles, ntiles);
    const int tiles_m = ntiles / tiles_n;
    const int GM = 8;
    const int group = tile / (GM * tiles_n), first_m = group * GM;
    const int gm = min(GM, tiles_m - first_m), in_group = tile % (GM * tiles_n);
    const int tile_m = first_m + in_group % gm, tile_n = in_group / gm;
    const int kt0 = kslice * ktper, kt1 = min(KT, kt0 + ktper);
    const int nkt = kt1 - kt0;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, lh = lane >> 5;

    const char *Ablk = TI == 2 ? Aimg + ((long)tile_m * KT + kt0) * BLOCK_BYTES
                               : Aimg + ((long)(tile_m >> 1) * KT + kt0) * BLOCK_BYTES + (tile_m & 1) * (PART_BYTES / 2);
    const char *Bblk = Bimg + ((long)tile_n * KT + kt0) * BLOCK_BYTES;

    f32x16 acc[TI][2];
#pragma unroll
    for (int i = 0; i < TI; ++i)
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
            aoff[i][ks] = swz_byte(wm * 32 * TI + (i < TI ? i : 0) * 32 + lr, ks * 2 + lh);
            boff[i][ks] = swz_byte(wn * 64 + i * 32 + lr, ks * 2 + lh);
        }

    // TN: byte offset of this lane's first transposed read inside a part, k-step 0 (k-step 1: + 16 rows = 4096)
    int aoffT[2], boffT[2];
    if (IO & 8) {
        static_assert(!(IO & 8) || PASSES == 1, "the TN operands are single-pass bf16");
        const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ca = wm * 64 + i * 32 + 16 * (g & 1) + 4 * pp, cb = wn * 64 + i * 32 + 16 * (g & 1) + 4 * pp;
            aoffT[i] = (8 * (g >> 1) + q) * 256 + ((((ca >> 3) + 4 * q) & 15) << 4) + (ca & 7) * 2;
            boffT[i] = (8 * (g >> 1) + q) * 256 + ((((cb >> 3) + 4 * q) & 15) << 4) + (cb & 7) * 2;
        }
    }
    // one k-tile of both operands into a ring slot
    auto stage = [&](int t, char *slot) {
        if (IO & 8) {
            stage_kmajor(p.Arm_hi, p.lda, tile_m * TR, (kt0 + t) * TK, p.M, slot, wave, lane);
            stage_kmajor(p.Brm, p.ldb, tile_n * TR, (kt0 + t) * TK, p.N, slot + OPER, wave, lane);
        } else if (TI == 1) {
            if (IO & 1) stage_rowmajor<PASSES, 1>(p.Arm_hi, p.Arm_lo, p.lda, tile_m * TRM, (kt0 + t) * TK, p.M, slot, wave, lane);
            else stage_block<PASSES, 1>(Ablk + (long)t * BLOCK_BYTES, slot, wave, lane);
            stage_block<PASSES>(Bblk + (long)t * BLOCK_BYTES, slot + OPER, wave, lane);
        } else {
            if (IO & 1) stage_rowmajor<PASSES>(p.Arm_hi, p.Arm_lo, p.lda, tile_m * TR, (kt0 + t) * TK, p.M, slot, wave, lane);
            else stage_block<PASSES>(Ablk + (long)t * BLOCK_BYTES, slot, wave, lane);
            stage_block<PASSES>(Bblk + (long)t * BLOCK_BYTES, slot + OPER, wave, lane);
        }
    };

    // prologue: NSTAGE-1 k-tiles in flight.  Tiles past the end are clamped to the last one (a
    // harmless re-read into a ring slot nobody reads again) so every iteration issues exactly
    // LOADS_PER_STAGE loads and the vmcnt arithmetic stays exact.
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s) {
        stage(min(s, nkt - 1), lds + s * SLOT);
    }
    if (NSTAGE == 1) stage(0, lds);      // single slot: filled here, refilled each iteration once every wave holds its fragments in registers
    for (int t = 0; t < nkt; ++t) {
        // tile t is complete once all but the newest (NSTAGE-2) stages have landed
        wait_vm_and_barrier<(NSTAGE >= 2 ? NSTAGE - 2 : 0) * LOADS>();
        if (NSTAGE >= 2) {   // refill the slot consumed in iteration t-1 (every wave is past it: they all passed the barrier)
            stage(min(t + NSTAGE - 1, nkt - 1), lds + ((t + NSTAGE - 1) % NSTAGE) * SLOT);
        }
        const char *cur = lds + (t % NSTAGE) * SLOT;
        const char *ah = cur, *al = cur + PART_BYTES, *bh = cur + OPER, *bl = cur + OPER + PART_BYTES;
        // both k-steps' fragments are requested up front: the second set lands under the first set's MFMAs
        bf16x8 fah[2][2], fal[2][2], fbh[2][2], fbl[2][2];
        if (IO & 8) {
            const unsigned la = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char *)ah;
            bf16x4_t xa[2][2][2], xb[2][2][2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    tr_read2(xa[ks][i][0], xa[ks][i][1], la + aoffT[i] + ks * 4096);
                    tr_read2(xb[ks][i][0], xb[ks][i][1], la + OPER + boffT[i] + ks * 4096);
                }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    fah[ks][i] = join8(xa[ks][i][0], xa[ks][i][1]);
                    fbh[ks][i] = join8(xb[ks][i][0], xb[ks][i][1]);
                }
            tr_wait(fah[0][0], fah[0][1], fah[1][0], fah[1][1], fbh[0][0], fbh[0][1], fbh[1][0], fbh[1][1]);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (IO & 8) continue;
                if (i < TI) fah[ks][i] = *reinterpret_cast<const bf16x8 *>(ah + aoff[i][ks]);
                fbh[ks][i] = *reinterpret_cast<const bf16x8 *>(bh + boff[i][ks]);
                if (PASSES == 3) {
                    fal[ks][i] = *reinterpret_cast<const bf16x8 *>(al + aoff[i][ks]);
                    fbl[ks][i] = *reinterpret_cast<const bf16x8 *>(bl + boff[i][ks]);
                }
            }
        if (NSTAGE == 1) {
            // every wave has its fragments: the slot is free, the next tile streams in under this tile's MFMAs
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            stage(min(t + 1, nkt - 1), lds);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (PASSES == 3) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fal[ks][i], fbh[ks][j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah[ks][i], fbl[ks][j], acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah[ks][i], fbh[ks][j], acc[i][j], 0, 0, 0);
                }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the clamped tail loads before the workgroup retires

    // epilogue (C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5))
    const int m0 = tile_m * TRM, n0 = tile_n * TR;
    if (rider) {            // the carried product's slice: raw sums to its slab
        float *slab = p.rslab + (long)kslice * p.rM * p.rN;
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = n0 + wn * 64 + j * 32 + lr;
                if (col >= p.rN) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * 32 * TI + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (row < p.rM) slab[(long)row * p.rN + col] = acc[i][j][r];
                }
            }
        return;
    }
    if (CE) {               // softmax statistics of this wave's 64 x 64 block, row by row (a row = 32 lanes x 2)
        const int c0 = n0 + wn * 64 + lr, c1 = c0 + 32;
        const bool ok0 = c0 < p.N, ok1 = c1 < p.N;
        float b0 = 0.f, b1 = 0.f;
        if (p.bias1) { if (ok0) b0 = p.bias1[c0]; if (ok1) b1 = p.bias1[c1]; }
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 32 * TI + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const float v0 = ok0 ? acc[i][0][r] + b0 : -INFINITY, v1 = ok1 ? acc[i][1][r] + b1 : -INFINITY;
                float mx = row16_max(fmaxf(v0, v1)), other = mx;
                swap_rows16(mx, other);                      // the two 16-lane rows of each 32-lane half
                mx = fmaxf(mx, other);
                const float msafe = mx == -INFINITY ? 0.f : mx;
                float sm = row16_sum(__expf(v0 - msafe) + __expf(v1 - msafe)), so = sm;
                swap_rows16(sm, so);
                sm += so;
                if (row >= p.M) continue;
                if (lr == 0) {
                    float *pp = p.ce_part + ((long)row * p.ce_strips + tile_n * 2 + wn) * 2;
                    pp[0] = mx; pp[1] = sm;
                }
                const long tgt = p.ce_target[row];
                if (ok0 && c0 == tgt) p.ce_tlogit[row] = v0;
                if (ok1 && c1 == tgt) p.ce_tlogit[row] = v1;
                if (p.C) {
                    if (ok0) p.C[(long)row * p.ldc + c0] = v0;
                    if (ok1) p.C[(long)row * p.ldc + c1] = v1;
                }
            }
        return;
    }
    if (EPI == 3) {
        // bias + addend R + result (the residual add; R may be C itself).  With loads and stores through possibly one buffer, loads
        // written in the store loop each wait for the store before them ([8192 x 768 x 768] bf16: 21 us plain, 39 us with the add); here the 16 addends of a 32 x 32
        // block are requested together (30 us).  Element (i, j, r) of this lane: row = m0 + 64 wm + 32 i + (r & 3) + 8 (r >> 2) + 4 lh,
        // col = n0 + 64 wn + 32 j + lr -- everything but 4 lh and lr is wave-uniform
        const int urow0 = m0 + wm * 32 * TI, ucol0 = n0 + wn * 64;
        float bias[2] = {0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = ucol0 + j * 32 + lr;
            if (col < p.N) {
                if (p.bias1) bias[j] += p.bias1[col];
                if (p.bias2) bias[j] += p.bias2[col];
            }
        }
        const unsigned lane_off = (unsigned)(4 * lh) * (unsigned)p.ldc + (unsigned)lr;
        const unsigned lane_off_r = (unsigned)(4 * lh) * (unsigned)p.ldr + (unsigned)lr;
#pragma unroll
        for (int q = 0; q < 2 * TI; ++q) {
            const int i = q >> 1, j = q & 1;
            const bool col_ok = ucol0 + j * 32 + lr < p.N;
            float *cb = p.C + (long)(urow0 + i * 32) * p.ldc + ucol0 + j * 32;       // wave-uniform
            const float *rb = p.R + (long)(urow0 + i * 32) * p.ldr + ucol0 + j * 32;
            float rcur[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ur = (r & 3) + 8 * (r >> 2);
                rcur[r] = (col_ok && urow0 + i * 32 + ur + 4 * lh < p.M) ? (rb + (long)ur * p.ldr)[lane_off_r] : 0.f;
            }
            if (!col_ok) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ur = (r & 3) + 8 * (r >> 2);
                if (urow0 + i * 32 + ur + 4 * lh < p.M) {
                    const float v = acc[i][j][r] + bias[j] + rcur[r];
                    if (IO & 2) {
                        const long o = (long)(urow0 + i * 32 + ur + 4 * lh) * p.ldo + ucol0 + j * 32 + lr;
                        const __bf16 h = (__bf16)v;
                        p.Ohi[o] = h;
                        if (PASSES == 3) p.Olo[o] = (__bf16)(v - (float)h);
                    }
                    if (!(IO & 2) || p.C) (cb + (long)ur * p.ldc)[lane_off] = v;
                }
            }
        }
        return;
    }
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < TI; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + lr;
            if (col >= p.N) continue;
            if (EPI == 2 || (EPI == 0 && p.ksplit > 1)) {
                float *slab = p.slab + (long)kslice * p.M * p.N;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * 32 * TI + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (row < p.M) slab[(long)row * p.N + col] = acc[i][j][r];
                }
                continue;
            }
            float bias = 0.f;
            if (p.bias1) bias += p.bias1[col];
            if (p.bias2) bias += p.bias2[col];
            if (IO & 4) {       // two output matrices, cut at a tile boundary (n0 is workgroup-uniform); no bias, activation or addend
                float *cq = n0 >= p.n_split ? p.C2 : p.C;
                const int ldq = n0 >= p.n_split ? p.ldc2 : p.ldc, colq = n0 >= p.n_split ? col - p.n_split : col;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * 32 * TI + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (row < p.M) { cq[(long)row * ldq + colq] = acc[i][j][r]; ss += acc[i][j][r] * acc[i][j][r]; }
                }
                continue;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 32 * TI + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row >= p.M) continue;
                float v = acc[i][j][r] + bias;
                if (EPI == 0) v = gemm_activation(v, p.relu);
                const long e = (long)row * p.ldc + col;
                if (EPI == 0 && p.use_drop) v *= dropout_mult(p.drop, (uint64_t)e);
                if (p.relu & 4) v += p.R[(long)row * p.ldr + col];
                if (IO & 2) {
                    const long o = (long)row * p.ldo + col;
                    const __bf16 h = (__bf16)v;
                    p.Ohi[o] = h;
                    if (PASSES == 3) p.Olo[o] = (__bf16)(v - (float)h);
                }
                if (!(IO & 2) || p.C) p.C[e] = v;
            }
        }
    }
    if ((IO & 4) && p.sumsq_part) {         // (workgroup-uniform; the ring's LDS is free: every wave is past its last k-tile after the barrier)
        ss = wave_sum(ss);
        __syncthreads();
        float *red = reinterpret_cast<float *>(lds);
        if (lane == 0) red[wave] = ss;
        __syncthreads();
        if (threadIdx.x == 0) p.sumsq_part[block] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}

template <int NSTAGE, int PASSES, bool CE = false, int EPI = 0, int IO = 0, int TI = 2>
__global__ __launch_bounds__(256) void gemm_bf16x3_kernel(const TiledGemmArgs p) {
    gemm_tile<NSTAGE, PASSES, CE, EPI, IO, TI>(p, (int)blockIdx.x);
}

// Several TN products (C_i [M_i][N_i] = A_i^T B_i over the SAME contraction length) in ONE grid, each on whole-K 128 x 128 tiles: the four
// weight gradients of a GPT block (ha/attention.py:96-144 under autograd) contract over the B T = 8192 token rows, and alone each has so
// few output tiles (36 .. 144) that it ran as three K-slices + a reduce launch; together they are 432 tiles -- no slabs, no reduces.
constexpr int TN_GROUP_MAX = 4;
struct TnGroup {
    const __bf16 *a[TN_GROUP_MAX], *b[TN_GROUP_MAX];
    float *c[TN_GROUP_MAX];
    long lda[TN_GROUP_MAX], ldb[TN_GROUP_MAX];
    int ldc[TN_GROUP_MAX], M[TN_GROUP_MAX], N[TN_GROUP_MAX], first[TN_GROUP_MAX];
    int n, KT, accumulate;
};
template <int EPI>
__global__ __launch_bounds__(256) void gemm_tn_group_kernel(const TnGroup g) {
    int q = 0;
#pragma unroll
    for (int i = 1; i < TN_GROUP_MAX; ++i)
        if (i < g.n && (int)blockIdx.x >= g.first[i]) q = i;
    TiledGemmArgs p = {};
    p.Arm_hi = g.a[q]; p.lda = g.lda[q]; p.Brm = g.b[q]; p.ldb = g.ldb[q];
    p.C = g.c[q]; p.R = g.c[q]; p.ldc = g.ldc[q]; p.ldr = g.ldc[q];
    p.M = g.M[q]; p.N = g.N[q]; p.KT = g.KT; p.ktper = g.KT; p.ksplit = 1;
    p.tiles_n = (p.N + TR - 1) / TR;
    p.ntiles = ((p.M + TR - 1) / TR) * p.tiles_n;
    gemm_tile<3, 1, false, EPI, 8, 2>(p, (int)blockIdx.x - g.first[q]);
}


// F.layer_norm fused with the operand-image pass of the GEMM that consumes it: one wave per row normalises the row (fp32,
// biased variance) and writes it straight into the tiled hi|lo image (and, optionally, as fp32 rows for the backward), so the
// normalised activations are not written and re-read once more just to be split.  C % 32 == 0 (whole k-tiles).
// MAXG > 0: the row (C <= 512 * MAXG) is held in registers -- one pass over x instead of three.  MAXG == 0: any C, three passes.
template <int MAXG>
__global__ __launch_bounds__(256) void layernorm_image_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                              const float *__restrict__ b, float *__restrict__ y, char *__restrict__ img,
                                                              int rows, int C, float eps, int with_lo, __bf16 *__restrict__ rm = nullptr) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float *xr = x + (long)row * C;
    const int groups = C / 8, KT = C / TK;
    const int rt = row / TR, rin = row % TR;
    auto emit = [&](int g, const f32x4 &a, const f32x4 &c, float mean, float rstd) {
        float wv[8], bv[8];
        if (MAXG > 0) {                                           // weight / bias 16-byte aligned (checked by the launcher)
            const f32x4 w0 = *reinterpret_cast<const f32x4 *>(w + 8 * g), w1 = *reinterpret_cast<const f32x4 *>(w + 8 * g + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { wv[e] = w0[e]; wv[4 + e] = w1[e]; }
            if (b) {
                const f32x4 b0 = *reinterpret_cast<const f32x4 *>(b + 8 * g), b1 = *reinterpret_cast<const f32x4 *>(b + 8 * g + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { bv[e] = b0[e]; bv[4 + e] = b1[e]; }
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) { wv[e] = w[8 * g + e]; bv[e] = b ? b[8 * g + e] : 0.f; }
        }
        float o[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o[e] = (a[e] - mean) * rstd * wv[e];
            o[4 + e] = (c[e] - mean) * rstd * wv[4 + e];
        }
        if (b) {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] += bv[e];
        }
        if (y) {
            *reinterpret_cast<f32x4 *>(y + (long)row * C + 8 * g) = f32x4{o[0], o[1], o[2], o[3]};
            *reinterpret_cast<f32x4 *>(y + (long)row * C + 8 * g + 4) = f32x4{o[4], o[5], o[6], o[7]};
        }
        bf16x8 hi, lo;
        split8(o, hi, lo);
        if (rm) *reinterpret_cast<bf16x8 *>(rm + (long)row * C + 8 * g) = hi;      // row-major bf16 (halo_layernorm_bf16)
        if (!img) return;
        char *blk = img + ((long)rt * KT + (8 * g) / TK) * BLOCK_BYTES;
        const int off = swz_byte(rin, ((8 * g) % TK) / 8);
        *reinterpret_cast<bf16x8 *>(blk + off) = hi;
        if (with_lo) *reinterpret_cast<bf16x8 *>(blk + PART_BYTES + off) = lo;
    };
    if (MAXG > 0) {
        f32x4 ra[MAXG > 0 ? MAXG : 1], rc[MAXG > 0 ? MAXG : 1];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < MAXG; ++k) {
            const int g = lane + 64 * k;
            const bool in = g < groups;
            ra[k] = in ? *reinterpret_cast<const f32x4 *>(xr + 8 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
            rc[k] = in ? *reinterpret_cast<const f32x4 *>(xr + 8 * g + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
            s += (ra[k][0] + ra[k][1]) + (ra[k][2] + ra[k][3]) + (rc[k][0] + rc[k][1]) + (rc[k][2] + rc[k][3]);
        }
        const float mean = wave_sum(s) / (float)C;
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < MAXG; ++k)
            if (lane + 64 * k < groups) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d0 = ra[k][e] - mean, d1 = rc[k][e] - mean; v += d0 * d0 + d1 * d1; }
            }
        const float rstd = rsqrtf(wave_sum(v) / (float)C + eps);
#pragma unroll
        for (int k = 0; k < MAXG; ++k)
            if (lane + 64 * k < groups) emit(lane + 64 * k, ra[k], rc[k], mean, rstd);
        return;
    }
    float s = 0.f;
    for (int g = lane; g < groups; g += 64) {
        const f32x4 a = *reinterpret_cast<const f32x4 *>(xr + 8 * g), c = *reinterpret_cast<const f32x4 *>(xr + 8 * g + 4);
        s += (a[0] + a[1]) + (a[2] + a[3]) + (c[0] + c[1]) + (c[2] + c[3]);
    }
    const float mean = wave_sum(s) / (float)C;
    float v = 0.f;
    for (int g = lane; g < groups; g += 64) {
        const f32x4 a = *reinterpret_cast<const f32x4 *>(xr + 8 * g), c = *reinterpret_cast<const f32x4 *>(xr + 8 * g + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d0 = a[e] - mean, d1 = c[e] - mean; v += d0 * d0 + d1 * d1; }
    }
    const float rstd = rsqrtf(wave_sum(v) / (float)C + eps);
    for (int g = lane; g < groups; g += 64) {
        const f32x4 a = *reinterpret_cast<const f32x4 *>(xr + 8 * g), c = *reinterpret_cast<const f32x4 *>(xr + 8 * g + 4);
        emit(g, a, c, mean, rstd);
    }
}

}  // namespace

size_t halo_tiled_image_bytes(int R, int K) {
    return (size_t)((R + TR - 1) / TR) * ((K + TK - 1) / TK) * BLOCK_BYTES;
}

int halo_prep_tiles(const float *src, int R, int K, int ld, int src_transposed, void *image, hipStream_t st) {
    const int with_lo = halo_math_mode() != HALO_MATH_BF16;   // images made in bf16 mode must be remade after a mode switch
    const int KT = (K + TK - 1) / TK, RT = (R + TR - 1) / TR;
    if (src_transposed)
        hipLaunchKernelGGL(prep_transposed_kernel, dim3(KT, RT), dim3(256), 0, st, src, R, K, ld, (char *)image, KT, with_lo);
    else
        hipLaunchKernelGGL(prep_rowmajor_kernel, dim3(KT, RT), dim3(256), 0, st, src, R, K, ld, (char *)image, KT, with_lo);
    return halo_launch_status();
}

int halo_prep_jobs(const HaloPrepJob *jobs, int n, hipStream_t st) {
    if (n <= 0) return HALO_OK;
    if (n > PREP_MAX_JOBS) return HALO_EINVAL;
    PrepJobs a = {};
    a.n = n;
    const int with_lo = halo_math_mode() != HALO_MATH_BF16;
    int total = 0;
    for (int i = 0; i < n; ++i) {
        const HaloPrepJob &q = jobs[i];
        PairArgs &p = a.job[i];
        p.src = q.src; p.ld = q.ld; p.R = q.R; p.C = q.K; p.with_lo = with_lo;
        p.img_rm = (char *)q.image; p.img_tr = (char *)q.image_tr;
        a.kind[i] = q.kind; a.first[i] = total;
        const int RT = (q.R + TR - 1) / TR;
        if (q.kind == 3) {                 // column sums of a few rows: one thread per column
            a.gx[i] = (q.K + 255) / 256;
            total += a.gx[i];
            continue;
        }
        if (q.kind == 2) {
            p.KT_rm = (q.K + TK - 1) / TK;
            p.KT_tr = (q.R + TK - 1) / TK;
            a.gx[i] = p.img_tr ? 4 * ((q.K + TR - 1) / TR) : p.KT_rm;
        } else {
            p.KT_rm = (q.K + TK - 1) / TK;
            a.gx[i] = p.KT_rm;
        }
        total += a.gx[i] * RT;
    }
    hipLaunchKernelGGL(prep_jobs_kernel, dim3((unsigned)total), dim3(256), 0, st, a);
    return halo_launch_status();
}

// loss[row] = logsumexp(logits[row, :]) - logits[row, target[row]] from the strip statistics of the fused epilogue (0 and lse 0 on
// ignored rows, like cross_entropy_kernel); one wave per row
__global__ __launch_bounds__(256) void ce_merge_kernel(const float *__restrict__ part, const float *__restrict__ tlogit,
                                                       const int64_t *__restrict__ target, float *__restrict__ loss,
                                                       float *__restrict__ lse_out, int rows, int strips, long ignore_index) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    if (target[row] == ignore_index) {
        if (lane == 0) { loss[row] = 0.f; if (lse_out) lse_out[row] = 0.f; }
        return;
    }
    const float *pr = part + (long)row * strips * 2;
    float m = -INFINITY;
    for (int t = lane; t < strips; t += 64) m = fmaxf(m, pr[2 * t]);
    m = wave_max(m);
    float s = 0.f;
    for (int t = lane; t < strips; t += 64) s += pr[2 * t + 1] * __expf(pr[2 * t] - m);      // exp(-inf) = 0 for all-padding strips
    s = wave_sum(s);
    if (lane == 0) {
        const float l = m + logf(s);
        loss[row] = l - tlogit[row];
        if (lse_out) lse_out[row] = l;
    }
}

static int launch_image_pair(PairArgs &p, int op, hipStream_t st) {
    p.with_lo = halo_math_mode() != HALO_MATH_BF16;
    p.KT_rm = (p.C + TK - 1) / TK;
    p.KT_tr = (p.R + TK - 1) / TK;
    // the transposed image is zero padded to whole 128-row tiles: cover the source columns in groups of four k-tiles
    const int kts = p.img_tr ? 4 * ((p.C + TR - 1) / TR) : p.KT_rm;
    const dim3 grid(kts, (p.R + TR - 1) / TR);
    switch (op) {
        case PAIR_COPY: hipLaunchKernelGGL(image_pair_kernel<PAIR_COPY>, grid, dim3(256), 0, st, p); break;
        case PAIR_GELU_TANH: hipLaunchKernelGGL(image_pair_kernel<PAIR_GELU_TANH>, grid, dim3(256), 0, st, p); break;
        case PAIR_GELU_ERF: hipLaunchKernelGGL(image_pair_kernel<PAIR_GELU_ERF>, grid, dim3(256), 0, st, p); break;
        case PAIR_GELU_TANH_BWD: hipLaunchKernelGGL(image_pair_kernel<PAIR_GELU_TANH_BWD>, grid, dim3(256), 0, st, p); break;
        case PAIR_GELU_ERF_BWD: hipLaunchKernelGGL(image_pair_kernel<PAIR_GELU_ERF_BWD>, grid, dim3(256), 0, st, p); break;
        case PAIR_CE_BWD: hipLaunchKernelGGL(image_pair_kernel<PAIR_CE_BWD>, grid, dim3(256), 0, st, p); break;
        default: return HALO_EINVAL;
    }
    return halo_launch_status();
}

// 64 x 128 workgroup tiles (TI = 1) for single-pass, whole-K products whose 128-row tiles would number 257 .. 511: a CU holds three
// workgroups, so such a launch leaves half the CUs with two and half with one; twice as many half-height tiles are three per CU.
// HALO_GEMM_HALF_TILES=0 keeps the 128-row tiles.
static bool half_tiles_wanted(int M, int ntiles128, int ksplit) {
    const char *e = getenv("HALO_GEMM_HALF_TILES");
    if (e && atoi(e) == 0) return false;
    return ksplit == 1 && ntiles128 > 256 && ntiles128 < 512 && M >= 128;
}

struct CeEpilogue {
    const int64_t *target;
    float *part, *tlogit;
};
static int gemm_bf16x3_tiled_impl(const void *Aimg, const void *Bimg, int M, int N, int K, float *C, int ldc,
                                  const float *bias1, const float *bias2, int relu, const DropoutCfg *drop, const CeEpilogue *ce,
                                  hipStream_t st, const float *resid, int ldr);

// both operand images of src [R][C] from one read: image_rm = halo_prep_tiles(src, R, C, ld, 0), image_tr = halo_prep_tiles(src, C, R, ld, 1)
int halo_prep_pair(const float *src, int R, int C, int ld, void *image_rm, void *image_tr, hipStream_t st) {
    PairArgs p = {};
    p.src = src; p.ld = ld; p.R = R; p.C = C;
    p.img_rm = (char *)image_rm; p.img_tr = (char *)image_tr;
    return launch_image_pair(p, PAIR_COPY, st);
}

int halo_gemm_bf16x3_tiled(const void *Aimg, const void *Bimg, int M, int N, int K, float *C, int ldc,
                           const float *bias1, const float *bias2, int relu, const DropoutCfg *drop, hipStream_t st) {
    return gemm_bf16x3_tiled_impl(Aimg, Bimg, M, N, K, C, ldc, bias1, bias2, relu, drop, nullptr, st, C, ldc);
}

static int gemm_bf16x3_tiled_impl(const void *Aimg, const void *Bimg, int M, int N, int K, float *C, int ldc,
                                  const float *bias1, const float *bias2, int relu, const DropoutCfg *drop, const CeEpilogue *ce,
                                  hipStream_t st, const float *resid, int ldr) {
    static int nstage = -1, nstage1 = 0;     // nstage: -1 not read yet, 0 chosen per call by tile count, else forced by HALO_GEMM_STAGES
    if (nstage < 0) {
        // three-pass ring depth: 1 slot = 32 KiB (three workgroups per CU), 2 = 64 KiB (two), 4 = 128 KiB (one); opt in to the LDS size once
        const char *e = getenv("HALO_GEMM_STAGES");
        const int want = e ? atoi(e) : 2;
        if (hipFuncSetAttribute((const void *)gemm_bf16x3_kernel<2, 3>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                2 * STAGE_BYTES) != hipSuccess ||
            hipFuncSetAttribute((const void *)gemm_bf16x3_kernel<4, 3>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                4 * STAGE_BYTES) != hipSuccess ||
            hipFuncSetAttribute((const void *)gemm_bf16x3_kernel<4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                2 * STAGE_BYTES) != hipSuccess ||
            hipFuncSetAttribute((const void *)gemm_bf16x3_kernel<8, 1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                4 * STAGE_BYTES) != hipSuccess ||
            hipFuncSetAttribute((const void *)gemm_bf16x3_kernel<2, 3, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                2 * STAGE_BYTES) != hipSuccess ||
            hipFuncSetAttribute((const void *)gemm_bf16x3_kernel<2, 3, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                2 * STAGE_BYTES) != hipSuccess ||
            hipFuncSetAttribute((const void *)gemm_bf16x3_kernel<2, 3, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                2 * STAGE_BYTES) != hipSuccess ||
            hipFuncSetAttribute((const void *)gemm_bf16x3_kernel<2, 3, false, 3>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                2 * STAGE_BYTES) != hipSuccess)
            return HALO_ELAUNCH;
        nstage = e ? (want == 4 ? 4 : (want == 1 ? 1 : 2)) : 0;      // 0: by tile count (below)
        // single pass, half-size slots: 3 = 48 KiB, THREE workgroups per CU (default: [8192 x 3072 x 768] 90 -> 69 us and the lm_head
        // product 1365 -> 1020 us against 4 slots on the same box); 4 = 64 KiB (two per CU), 8 = 128 KiB (one), 2 = 32 KiB
        const char *e1 = getenv("HALO_GEMM_STAGES_BF16");
        nstage1 = e1 ? atoi(e1) : 3;
        if (nstage1 != 8 && nstage1 != 4 && nstage1 != 2) nstage1 = 3;
    }
    TiledGemmArgs p = {};
    p.A = (const char *)Aimg; p.B = (const char *)Bimg; p.C = C; p.R = resid; p.ldr = ldr; p.bias1 = bias1; p.bias2 = bias2;
    p.M = M; p.N = N; p.KT = (K + TK - 1) / TK; p.ldc = ldc; p.relu = relu;
    p.tiles_n = (N + TR - 1) / TR;
    p.use_drop = drop && drop->threshold != 0u;
    if (drop) p.drop = *drop; else p.drop = make_dropout(0.f, 0, 0, 0, nullptr);
    p.ntiles = ((M + TR - 1) / TR) * p.tiles_n;
    p.ksplit = ce ? 1 : halo_pick_ksplit(p.ntiles, p.KT, (long)M * N);      // the fused statistics need whole dot products
    p.ce_target = ce ? ce->target : nullptr; p.ce_part = ce ? ce->part : nullptr; p.ce_tlogit = ce ? ce->tlogit : nullptr;
    p.ce_strips = 2 * p.tiles_n;
    p.ktper = (p.KT + p.ksplit - 1) / p.ksplit;
    p.ksplit = (p.KT + p.ktper - 1) / p.ktper;
    void *scratch; size_t bytes;
    halo_get_scratch(&scratch, &bytes);
    p.slab = (float *)scratch;
    const dim3 grid((unsigned)(p.ntiles * p.ksplit));
    const bool one_pass = halo_math_mode() == HALO_MATH_BF16;
    // epilogue specialisations (same-box A/B: the LSTM-CTC step 1.274 -> 1.247 ms, products 1-7 % faster)
    const bool slice = p.ksplit > 1;                                                     // raw sums: the reduce kernel applies the epilogue
    const bool lean_any = p.ksplit == 1 && !p.use_drop && (relu & ~HALO_GEMM_ACCUM) == 0;
    const bool lean = lean_any && !(relu & HALO_GEMM_ACCUM), lean_add = lean_any && (relu & HALO_GEMM_ACCUM);   // bias only / bias + C += result
    if (one_pass && nstage1 == 3 && !ce && (lean || lean_add) && half_tiles_wanted(M, p.ntiles, p.ksplit)) {
        p.ntiles = ((M + 63) / 64) * p.tiles_n;
        const dim3 gh((unsigned)p.ntiles);
        if (lean) hipLaunchKernelGGL((gemm_bf16x3_kernel<3, 1, false, 1, 0, 1>), gh, dim3(256), 3 * (PART_BYTES / 2 + PART_BYTES), st, p);
        else hipLaunchKernelGGL((gemm_bf16x3_kernel<3, 1, false, 3, 0, 1>), gh, dim3(256), 3 * (PART_BYTES / 2 + PART_BYTES), st, p);
        return halo_launch_status();
    }
    if (ce) {                       // the epilogue instantiations (kept apart: compiled into the common kernel the extra registers and
                                    // epilogue code cost every product 6-17 % and the LSTM-CTC step 2.6 %, same-box A/B)
        if (one_pass) hipLaunchKernelGGL((gemm_bf16x3_kernel<3, 1, true>), grid, dim3(256), 3 * STAGE_BYTES / 2, st, p);
        else if ((long)p.ntiles * p.ksplit >= 768) hipLaunchKernelGGL((gemm_bf16x3_kernel<1, 3, true>), grid, dim3(256), STAGE_BYTES, st, p);
        else hipLaunchKernelGGL((gemm_bf16x3_kernel<2, 3, true>), grid, dim3(256), 2 * STAGE_BYTES, st, p);
    } else
    if (one_pass && nstage1 == 3 && lean) hipLaunchKernelGGL((gemm_bf16x3_kernel<3, 1, false, 1>), grid, dim3(256), 3 * STAGE_BYTES / 2, st, p);
    else if (one_pass && nstage1 == 3 && lean_add) hipLaunchKernelGGL((gemm_bf16x3_kernel<3, 1, false, 3>), grid, dim3(256), 3 * STAGE_BYTES / 2, st, p);
    else if (one_pass && nstage1 == 3 && slice) hipLaunchKernelGGL((gemm_bf16x3_kernel<3, 1, false, 2>), grid, dim3(256), 3 * STAGE_BYTES / 2, st, p);
    else if (one_pass && nstage1 == 3) hipLaunchKernelGGL((gemm_bf16x3_kernel<3, 1>), grid, dim3(256), 3 * STAGE_BYTES / 2, st, p);
    else if (one_pass && nstage1 == 2) hipLaunchKernelGGL((gemm_bf16x3_kernel<2, 1>), grid, dim3(256), STAGE_BYTES, st, p);
    else if (one_pass && nstage1 == 8) hipLaunchKernelGGL((gemm_bf16x3_kernel<8, 1>), grid, dim3(256), 4 * STAGE_BYTES, st, p);
    else if (one_pass) hipLaunchKernelGGL((gemm_bf16x3_kernel<4, 1>), grid, dim3(256), 2 * STAGE_BYTES, st, p);
    // three passes: ONE 32 KiB slot (refilled under the MFMAs once every wave holds its fragments) lets three workgroups share a CU
    // and wins where there are that many (>= 768 tiles: [8192 x 3072 x 768] 153 -> 136 us, lm_head 2172 -> 1928 us); with fewer the
    // two-slot ring at two per CU is faster ([4096 x 1024 x 1344], the LSTM weight gradient: 54 vs 63 us)
    else if (nstage == 1 || (nstage == 0 && (long)p.ntiles * p.ksplit >= 768)) {
        if (lean) hipLaunchKernelGGL((gemm_bf16x3_kernel<1, 3, false, 1>), grid, dim3(256), STAGE_BYTES, st, p);
        else if (lean_add) hipLaunchKernelGGL((gemm_bf16x3_kernel<1, 3, false, 3>), grid, dim3(256), STAGE_BYTES, st, p);
        else hipLaunchKernelGGL((gemm_bf16x3_kernel<1, 3>), grid, dim3(256), STAGE_BYTES, st, p);
    } else if (nstage == 4) hipLaunchKernelGGL((gemm_bf16x3_kernel<4, 3>), grid, dim3(256), 4 * STAGE_BYTES, st, p);
    else if (lean) hipLaunchKernelGGL((gemm_bf16x3_kernel<2, 3, false, 1>), grid, dim3(256), 2 * STAGE_BYTES, st, p);
    else if (lean_add) hipLaunchKernelGGL((gemm_bf16x3_kernel<2, 3, false, 3>), grid, dim3(256), 2 * STAGE_BYTES, st, p);
    else if (slice) hipLaunchKernelGGL((gemm_bf16x3_kernel<2, 3, false, 2>), grid, dim3(256), 2 * STAGE_BYTES, st, p);
    else hipLaunchKernelGGL((gemm_bf16x3_kernel<2, 3>), grid, dim3(256), 2 * STAGE_BYTES, st, p);
    int rc = halo_launch_status();
    if (rc != HALO_OK || p.ksplit == 1) return rc;
    // the reduce adds into C: an addend held elsewhere is copied there first
    if ((relu & HALO_GEMM_ACCUM) && resid != C &&
        hipMemcpy2DAsync(C, (size_t)ldc * sizeof(float), resid, (size_t)ldr * sizeof(float), (size_t)N * sizeof(float), (size_t)M,
                         hipMemcpyDeviceToDevice, st) != hipSuccess)
        return HALO_ELAUNCH;
    return halo_splitk_reduce(p.slab, p.ksplit, M, N, C, ldc, bias1, bias2, relu, p.drop, p.use_drop, st);
}


// K-slices of A [M][K] x B [N][K]^T left UNREDUCED: slice s (k-tiles [s ktper, (s + 1) ktper)) goes to slab + s M N as plain [M][N] sums, for a
// consumer that adds the slices while it reads them (the front-end convolution's backward reads the LSTM's input gradient exactly once:
// the reduce launch between them, 8 us for 2.7 MB, is pure latency).  *slices = how many were written (<= want).
int halo_gemm_bf16x3_tiled_slices(const void *Aimg, const void *Bimg, int M, int N, int K, float *slab, int want, int *slices, hipStream_t st) {
    if (!Aimg || !Bimg || !slab || !slices || M <= 0 || N <= 0 || K <= 0 || want < 1) return HALO_EINVAL;
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void *)gemm_bf16x3_kernel<2, 3, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES) != hipSuccess)
            return HALO_ELAUNCH;
        attr = true;
    }
    TiledGemmArgs p = {};
    p.A = (const char *)Aimg; p.B = (const char *)Bimg;
    p.M = M; p.N = N; p.KT = (K + TK - 1) / TK; p.ldc = N;
    p.tiles_n = (N + TR - 1) / TR;
    p.drop = make_dropout(0.f, 0, 0, 0, nullptr);
    p.ntiles = ((M + TR - 1) / TR) * p.tiles_n;
    p.ktper = (p.KT + want - 1) / want;
    p.ksplit = (p.KT + p.ktper - 1) / p.ktper;
    p.slab = slab;
    *slices = p.ksplit;
    const dim3 grid((unsigned)(p.ntiles * p.ksplit));
    if (halo_math_mode() == HALO_MATH_BF16) hipLaunchKernelGGL((gemm_bf16x3_kernel<3, 1, false, 2>), grid, dim3(256), 3 * STAGE_BYTES / 2, st, p);
    else hipLaunchKernelGGL((gemm_bf16x3_kernel<2, 3, false, 2>), grid, dim3(256), 2 * STAGE_BYTES, st, p);
    return halo_launch_status();
}

// C [M][n_split] | C2 [M][N - n_split] = A [M][K] x (B [n_split][K] stacked on B2 [N - n_split][K])^T in ONE launch: the image of the
// stacked operand is the two images one after the other when n_split is a multiple of the 128-row tile.  The LSTM's two weight-gradient
// products of a layer share dG^T this way (lstm.hip).  Plain sums: no bias, activation, dropout or split-K.
int halo_gemm_bf16x3_tiled_nsplit(const void *Aimg, const void *Bimg, int M, int N, int K, float *C, int ldc, int n_split, float *C2, int ldc2,
                                  hipStream_t st) {
    return halo_gemm_bf16x3_tiled_nsplit_carry(Aimg, Bimg, M, N, K, C, ldc, n_split, C2, ldc2, nullptr, nullptr, 0, 0, 0, nullptr, 0, nullptr, nullptr, nullptr, st);
}

// ... carrying halo_gemm_bf16x3_tiled_slices(rA, rB, rM, rN, rK, rslab, want, slices) in the same launch when the main product runs on
// whole 128-row tiles in single-pass mode and leaves room beside them (*slices > 0 on return: carried; 0: the caller launches it itself)
int halo_gemm_bf16x3_tiled_nsplit_carry(const void *Aimg, const void *Bimg, int M, int N, int K, float *C, int ldc, int n_split, float *C2, int ldc2,
                                        const void *rA, const void *rB, int rM, int rN, int rK, float *rslab, int want, int *slices,
                                        float *sumsq_part, int *sumsq_parts, hipStream_t st) {
    if (slices) *slices = 0;
    if (n_split % TR != 0 || n_split <= 0 || n_split >= N) return HALO_EINVAL;
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void *)gemm_bf16x3_kernel<2, 3, false, 1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES) != hipSuccess ||
            hipFuncSetAttribute((const void *)gemm_bf16x3_kernel<1, 3, false, 1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, STAGE_BYTES) != hipSuccess ||
            hipFuncSetAttribute((const void *)gemm_bf16x3_kernel<3, 1, false, 1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * STAGE_BYTES / 2) != hipSuccess)
            return HALO_ELAUNCH;
        attr = true;
    }
    TiledGemmArgs p = {};
    p.A = (const char *)Aimg; p.B = (const char *)Bimg; p.C = C; p.C2 = C2; p.n_split = n_split; p.ldc2 = ldc2;
    p.M = M; p.N = N; p.KT = (K + TK - 1) / TK; p.ldc = ldc;
    p.tiles_n = (N + TR - 1) / TR;
    p.drop = make_dropout(0.f, 0, 0, 0, nullptr);
    p.ntiles = ((M + TR - 1) / TR) * p.tiles_n;
    p.ksplit = 1; p.ktper = p.KT;
    dim3 grid((unsigned)p.ntiles);
    if (rA && rB && rslab && slices && want >= 1 && rM > 0 && rN > 0 && rK > 0 && halo_math_mode() == HALO_MATH_BF16 &&
        !half_tiles_wanted(M, p.ntiles, 1)) {
        p.rA = (const char *)rA; p.rB = (const char *)rB; p.rslab = rslab; p.rM = rM; p.rN = rN;
        p.rKT = (rK + TK - 1) / TK;
        p.r_tiles_n = (rN + TR - 1) / TR;
        p.r_ntiles = ((rM + TR - 1) / TR) * p.r_tiles_n;
        p.r_ktper = (p.rKT + want - 1) / want;
        const int rs = (p.rKT + p.r_ktper - 1) / p.r_ktper;
        p.rider_first = p.ntiles;
        grid = dim3((unsigned)(p.ntiles + p.r_ntiles * rs));
        *slices = rs;
    }
    p.sumsq_part = sumsq_parts ? sumsq_part : nullptr;
    if (halo_math_mode() == HALO_MATH_BF16 && half_tiles_wanted(M, p.ntiles, 1)) {
        p.ntiles = ((M + 63) / 64) * p.tiles_n;
        if (sumsq_parts) *sumsq_parts = p.ntiles;
        hipLaunchKernelGGL((gemm_bf16x3_kernel<3, 1, false, 1, 4, 1>), dim3((unsigned)p.ntiles), dim3(256), 3 * (PART_BYTES / 2 + PART_BYTES), st, p);
        return halo_launch_status();
    }
    if (sumsq_parts) *sumsq_parts = p.ntiles;       // (the carried product's workgroups write none)
    if (halo_math_mode() == HALO_MATH_BF16) hipLaunchKernelGGL((gemm_bf16x3_kernel<3, 1, false, 1, 4>), grid, dim3(256), 3 * STAGE_BYTES / 2, st, p);
    else if (p.ntiles >= 768) hipLaunchKernelGGL((gemm_bf16x3_kernel<1, 3, false, 1, 4>), grid, dim3(256), STAGE_BYTES, st, p);
    else hipLaunchKernelGGL((gemm_bf16x3_kernel<2, 3, false, 1, 4>), grid, dim3(256), 2 * STAGE_BYTES, st, p);
    return halo_launch_status();
}

// The tiled product with row-major bf16 on either side (IO instantiations of the kernel): A from a row-major bf16 matrix (a_hi [, a_lo])
// instead of an image, and / or the result as row-major bf16 (o_hi [, o_lo]) beside or instead of fp32 C.  No split-K, no dropout.
template <int EPI, int IO>
static int launch_io(const TiledGemmArgs &p0, bool one_pass, hipStream_t st) {
    TiledGemmArgs p = p0;
    if constexpr (IO == 1 && (EPI == 1 || EPI == 3)) {
        if (one_pass && half_tiles_wanted(p.M, p.ntiles, 1)) {
            p.ntiles = ((p.M + 63) / 64) * p.tiles_n;
            hipLaunchKernelGGL((gemm_bf16x3_kernel<3, 1, false, EPI, IO, 1>), dim3((unsigned)p.ntiles), dim3(256), 3 * (PART_BYTES / 2 + PART_BYTES), st, p);
            return halo_launch_status();
        }
    }
    const dim3 grid((unsigned)p.ntiles);
    if (one_pass) hipLaunchKernelGGL((gemm_bf16x3_kernel<3, 1, false, EPI, IO>), grid, dim3(256), 3 * STAGE_BYTES / 2, st, p);
    else if (p.ntiles >= 768) hipLaunchKernelGGL((gemm_bf16x3_kernel<1, 3, false, EPI, IO>), grid, dim3(256), STAGE_BYTES, st, p);
    else {
        static bool attr = false;
        if (!attr) {
            if (hipFuncSetAttribute((const void *)gemm_bf16x3_kernel<2, 3, false, EPI, IO>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    2 * STAGE_BYTES) != hipSuccess)
                return HALO_ELAUNCH;
            attr = true;
        }
        hipLaunchKernelGGL((gemm_bf16x3_kernel<2, 3, false, EPI, IO>), grid, dim3(256), 2 * STAGE_BYTES, st, p);
    }
    return halo_launch_status();
}

static int gemm_bf16x3_io(const void *Aimg, const void *a_hi, const void *a_lo, long lda, const void *Bimg, int M, int N, int K, float *C,
                          int ldc, void *o_hi, void *o_lo, long ldo, const float *resid, int ldr, const float *bias1, const float *bias2,
                          int flags, hipStream_t st) {
    const bool one_pass = halo_math_mode() == HALO_MATH_BF16;
    const int io = (a_hi ? 1 : 0) | (o_hi ? 2 : 0);
    TiledGemmArgs p = {};
    p.A = (const char *)Aimg; p.B = (const char *)Bimg; p.C = C; p.R = resid ? resid : C; p.ldr = resid ? ldr : ldc;
    p.Arm_hi = (const __bf16 *)a_hi; p.Arm_lo = (const __bf16 *)a_lo; p.lda = lda;
    p.Ohi = (__bf16 *)o_hi; p.Olo = (__bf16 *)o_lo; p.ldo = ldo;
    p.bias1 = bias1; p.bias2 = bias2;
    p.M = M; p.N = N; p.KT = (K + TK - 1) / TK; p.ldc = ldc; p.relu = flags;
    p.tiles_n = (N + TR - 1) / TR;
    p.drop = make_dropout(0.f, 0, 0, 0, nullptr);
    p.ntiles = ((M + TR - 1) / TR) * p.tiles_n;
    p.ksplit = 1; p.ktper = p.KT;
    const bool act = (flags & ~HALO_GEMM_ACCUM) != 0, add = (flags & HALO_GEMM_ACCUM) != 0;
    if (io == 2 && act && !add) return launch_io<0, 2>(p, one_pass, st);
    if (io == 2 && !act && !add) return launch_io<1, 2>(p, one_pass, st);
    if (io == 1 && !act && add) return launch_io<3, 1>(p, one_pass, st);
    if (io == 1 && !act && !add) return launch_io<1, 1>(p, one_pass, st);
    return HALO_ENOTSUP;
}

// C [M][N] (+)= A^T B with A [K][M], B [K][N] row-major bf16 (IO & 8).  Split-K by the same rule as the image products.
static int gemm_tn_bf16(const void *a, long lda, const void *b, long ldb, int M, int N, int K, float *C, int ldc, int flags, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void *)gemm_bf16x3_kernel<3, 1, false, 1, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * STAGE_BYTES / 2) != hipSuccess ||
            hipFuncSetAttribute((const void *)gemm_bf16x3_kernel<3, 1, false, 2, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * STAGE_BYTES / 2) != hipSuccess ||
            hipFuncSetAttribute((const void *)gemm_bf16x3_kernel<3, 1, false, 3, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * STAGE_BYTES / 2) != hipSuccess)
            return HALO_ELAUNCH;
        attr = true;
    }
    TiledGemmArgs p = {};
    p.Arm_hi = (const __bf16 *)a; p.lda = lda; p.Brm = (const __bf16 *)b; p.ldb = ldb;
    p.C = C; p.R = C; p.ldr = ldc; p.ldc = ldc;
    p.M = M; p.N = N; p.KT = K / TK; p.relu = flags;
    p.tiles_n = (N + TR - 1) / TR;
    p.drop = make_dropout(0.f, 0, 0, 0, nullptr);
    p.ntiles = ((M + TR - 1) / TR) * p.tiles_n;
    p.ksplit = halo_pick_ksplit(p.ntiles, p.KT, (long)M * N);
    p.ktper = (p.KT + p.ksplit - 1) / p.ksplit;
    p.ksplit = (p.KT + p.ktper - 1) / p.ktper;
    void *scratch; size_t bytes;
    halo_get_scratch(&scratch, &bytes);
    p.slab = (float *)scratch;
    const dim3 grid((unsigned)(p.ntiles * p.ksplit));
    if (p.ksplit > 1) hipLaunchKernelGGL((gemm_bf16x3_kernel<3, 1, false, 2, 8>), grid, dim3(256), 3 * STAGE_BYTES / 2, st, p);
    else if (flags & HALO_GEMM_ACCUM) hipLaunchKernelGGL((gemm_bf16x3_kernel<3, 1, false, 3, 8>), grid, dim3(256), 3 * STAGE_BYTES / 2, st, p);
    else hipLaunchKernelGGL((gemm_bf16x3_kernel<3, 1, false, 1, 8>), grid, dim3(256), 3 * STAGE_BYTES / 2, st, p);
    const int rc = halo_launch_status();
    if (rc != HALO_OK || p.ksplit == 1) return rc;
    return halo_splitk_reduce(p.slab, p.ksplit, M, N, C, ldc, nullptr, nullptr, flags, p.drop, 0, st);
}

static int gemm_tn_bf16_group(int n, const void *const *a, const long *lda, const void *const *b, const long *ldb, const int *M, const int *N, int K,
                              float *const *C, const int *ldc, int flags, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void *)gemm_tn_group_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * STAGE_BYTES / 2) != hipSuccess ||
            hipFuncSetAttribute((const void *)gemm_tn_group_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * STAGE_BYTES / 2) != hipSuccess)
            return HALO_ELAUNCH;
        attr = true;
    }
    TnGroup g = {};
    g.n = n; g.KT = K / TK; g.accumulate = (flags & HALO_GEMM_ACCUM) ? 1 : 0;
    int total = 0;
    for (int i = 0; i < n; ++i) {
        g.a[i] = (const __bf16 *)a[i]; g.lda[i] = lda[i]; g.b[i] = (const __bf16 *)b[i]; g.ldb[i] = ldb[i];
        g.c[i] = C[i]; g.ldc[i] = ldc[i]; g.M[i] = M[i]; g.N[i] = N[i]; g.first[i] = total;
        total += ((M[i] + TR - 1) / TR) * ((N[i] + TR - 1) / TR);
    }
    if (g.accumulate) hipLaunchKernelGGL(gemm_tn_group_kernel<3>, dim3((unsigned)total), dim3(256), 3 * STAGE_BYTES / 2, st, g);
    else hipLaunchKernelGGL(gemm_tn_group_kernel<1>, dim3((unsigned)total), dim3(256), 3 * STAGE_BYTES / 2, st, g);
    return halo_launch_status();
}

// Diagnostic (halo_debug_mfma_clock): the clock the chip holds under a bare bf16 MFMA loop on random operands -- fragments in
// registers, no memory traffic, four independent accumulators per wave, one wave per SIMD -- as delta s_memtime / delta s_memrealtime
// x 100 MHz (MI355X_MICROARCH.md, 'DVFS give-back' item 6), and with it the matrix pipes' sustained rate.  SHAPE 0: 32x32x16, 1: 16x16x32.
template <int SHAPE>
__global__ __launch_bounds__(256) void mfma_clock_kernel(unsigned long long *out, int iters, unsigned seed, float *sink) {
    unsigned h = seed ^ (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
    auto rnd = [&]() { h ^= h << 13; h ^= h >> 17; h ^= h << 5; return (__bf16)(((int)(h & 0xffff) - 32768) * (1.0f / 32768.0f)); };
    bf16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) { a[i][e] = rnd(); b[i][e] = rnd(); }
    f32x16 acc32[4];
    f32x4 acc16[8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc32[i][e] = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc16[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (SHAPE == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc32[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j], b[(j + 1) & 3], acc32[j], 0, 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc16[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j & 3], b[(j + 1) & 3], acc16[j], 0, 0, 0);
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc32[i][0] + acc32[i][7];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc16[i][0];
    if (s == 123456.789f) *sink = s;                 // keeps the accumulators alive; never true in practice
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = c1 - c0; out[2 * blockIdx.x + 1] = r1 - r0; }
}

// ---- public entry points (include/halo.h) -----------------------------------------------------
extern "C" {

int halo_debug_mfma_clock(unsigned long long *ticks, float *sink, int blocks, int iters, int shape, unsigned seed, halo_stream_t stream) {
    HALO_CHECK_ARG(ticks && sink && blocks > 0 && iters > 0 && (shape == 0 || shape == 1));
    if (shape == 0) hipLaunchKernelGGL(mfma_clock_kernel<0>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, ticks, iters, seed, sink);
    else hipLaunchKernelGGL(mfma_clock_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, ticks, iters, seed, sink);
    return halo_launch_status();
}

int halo_gemm_tn_bf16(const void *a, long lda, const void *b, long ldb, int M, int N, int K, float *C, int ldc, int flags,
                      halo_stream_t stream) {
    HALO_CHECK_ARG(a && b && C && M > 0 && N > 0 && K > 0);
    HALO_CHECK_ARG(K % TK == 0 && M % 8 == 0 && N % 8 == 0 && lda >= M && ldb >= N && ldc >= N && lda % 8 == 0 && ldb % 8 == 0);
    HALO_CHECK_ARG(((uintptr_t)a | (uintptr_t)b) % 16 == 0 && (flags & ~HALO_GEMM_ACCUM) == 0);
    return gemm_tn_bf16(a, lda, b, ldb, M, N, K, C, ldc, flags, (hipStream_t)stream);
}

int halo_gemm_tn_bf16_group(int n, const void *const *a, const long *lda, const void *const *b, const long *ldb, const int *M, const int *N, int K,
                            float *const *C, const int *ldc, int flags, halo_stream_t stream) {
    HALO_CHECK_ARG(n >= 1 && n <= TN_GROUP_MAX && a && lda && b && ldb && M && N && C && ldc && K > 0 && K % TK == 0 && (flags & ~HALO_GEMM_ACCUM) == 0);
    for (int i = 0; i < n; ++i) {
        HALO_CHECK_ARG(a[i] && b[i] && C[i] && M[i] > 0 && N[i] > 0 && M[i] % 8 == 0 && N[i] % 8 == 0);
        HALO_CHECK_ARG(lda[i] >= M[i] && ldb[i] >= N[i] && ldc[i] >= N[i] && lda[i] % 8 == 0 && ldb[i] % 8 == 0);
        HALO_CHECK_ARG(((uintptr_t)a[i] | (uintptr_t)b[i]) % 16 == 0);
    }
    return gemm_tn_bf16_group(n, a, lda, b, ldb, M, N, K, C, ldc, flags, (hipStream_t)stream);
}

size_t halo_split_image_bytes(int rows, int k) {
    if (rows <= 0 || k <= 0) return 0;
    return halo_tiled_image_bytes(rows, k);
}

int halo_split_image(const float *src, int rows, int k, int ld, int src_transposed, void *image, halo_stream_t stream) {
    HALO_CHECK_ARG(src && image && rows > 0 && k > 0);
    HALO_CHECK_ARG(ld >= (src_transposed ? rows : k));
    HALO_CHECK_ARG((uintptr_t)image % 16 == 0);
    return halo_prep_tiles(src, rows, k, ld, src_transposed, image, (hipStream_t)stream);
}

int halo_image_pair(const float *src, const float *src2, int rows, int cols, long ld, long ld2, int op, void *image_rows,
                    void *image_cols, halo_stream_t stream) {
    HALO_CHECK_ARG(src && rows > 0 && cols > 0 && ld >= cols && (image_rows || image_cols));
    HALO_CHECK_ARG(op >= HALO_PAIR_COPY && op <= HALO_PAIR_GELU_ERF_BWD);
    const bool two = op == HALO_PAIR_GELU_TANH_BWD || op == HALO_PAIR_GELU_ERF_BWD;
    HALO_CHECK_ARG(!two || (src2 && ld2 >= cols));
    HALO_CHECK_ARG(((uintptr_t)image_rows | (uintptr_t)image_cols) % 16 == 0);
    PairArgs p = {};
    p.src = src; p.src2 = two ? src2 : nullptr; p.ld = ld; p.ld2 = ld2; p.R = rows; p.C = cols;
    p.img_rm = (char *)image_rows; p.img_tr = (char *)image_cols;
    return launch_image_pair(p, op, (hipStream_t)stream);
}

int halo_cross_entropy_bwd_images(const float *logits, const int64_t *targets, const float *lse, const float *grad, long grad_stride,
                                  int rows, int V, long ld, long ignore_index, void *image_rows, void *image_cols,
                                  halo_stream_t stream) {
    HALO_CHECK_ARG(logits && targets && lse && grad && rows > 0 && V > 0 && ld >= V && (grad_stride == 0 || grad_stride == 1));
    HALO_CHECK_ARG((image_rows || image_cols) && ((uintptr_t)image_rows | (uintptr_t)image_cols) % 16 == 0);
    PairArgs p = {};
    p.src = logits; p.ld = ld; p.R = rows; p.C = V;
    p.img_rm = (char *)image_rows; p.img_tr = (char *)image_cols;
    p.target = targets; p.lse = lse; p.grad = grad; p.grad_stride = grad_stride; p.ignore_index = ignore_index;
    return launch_image_pair(p, PAIR_CE_BWD, (hipStream_t)stream);
}

size_t halo_gemm_split_ce_workspace_bytes(int M, int N) {
    if (M <= 0 || N <= 0) return 0;
    return ((size_t)M * 2 * ((N + TR - 1) / TR) * 2 + (size_t)M) * sizeof(float);
}

int halo_gemm_split_ce(const void *a_image, const void *b_image, int M, int N, int K, float *logits, int ldc, const float *bias,
                       const int64_t *targets, long ignore_index, void *workspace, float *loss, float *lse, halo_stream_t stream) {
    HALO_CHECK_ARG(a_image && b_image && targets && workspace && loss && M > 0 && N > 0 && K > 0);
    HALO_CHECK_ARG(!logits || ldc >= N);
    HALO_CHECK_ARG(halo_math_mode() != HALO_MATH_F32);
    hipStream_t st = (hipStream_t)stream;
    const int strips = 2 * ((N + TR - 1) / TR);
    CeEpilogue ce;
    ce.target = targets; ce.part = (float *)workspace; ce.tlogit = ce.part + (size_t)M * strips * 2;
    int rc = gemm_bf16x3_tiled_impl(a_image, b_image, M, N, K, logits, ldc, bias, nullptr, 0, nullptr, &ce, st, logits, ldc);
    if (rc != HALO_OK) return rc;
    hipLaunchKernelGGL(ce_merge_kernel, dim3((M + 3) / 4), dim3(256), 0, st, ce.part, ce.tlogit, targets, loss, lse, M, strips,
                       ignore_index);
    return halo_launch_status();
}

static int layernorm_image_launch(const float *x, const float *weight, const float *bias, float *y, void *image, __bf16 *rm, int rows, int C,
                                  float eps, hipStream_t st) {
    const int with_lo = halo_math_mode() != HALO_MATH_BF16;
    const bool vec = (((uintptr_t)weight | (uintptr_t)bias) % 16) == 0;       // the row-in-registers variants load w / b as float4
    const dim3 grid((rows + 3) / 4);
    if (vec && C <= 1024) hipLaunchKernelGGL(layernorm_image_kernel<2>, grid, dim3(256), 0, st, x, weight, bias, y, (char *)image, rows, C, eps, with_lo, rm);
    else if (vec && C <= 2048) hipLaunchKernelGGL(layernorm_image_kernel<4>, grid, dim3(256), 0, st, x, weight, bias, y, (char *)image, rows, C, eps, with_lo, rm);
    else hipLaunchKernelGGL(layernorm_image_kernel<0>, grid, dim3(256), 0, st, x, weight, bias, y, (char *)image, rows, C, eps, with_lo, rm);
    return halo_launch_status();
}

int halo_layernorm_image(const float *x, const float *weight, const float *bias, float *y, void *image, int rows, int C, float eps,
                         halo_stream_t stream) {
    HALO_CHECK_ARG(x && weight && image && rows > 0 && C > 0 && C % TK == 0);
    HALO_CHECK_ARG(((uintptr_t)x | (uintptr_t)image | (uintptr_t)y) % 16 == 0);
    return layernorm_image_launch(x, weight, bias, y, image, nullptr, rows, C, eps, (hipStream_t)stream);
}

int halo_layernorm_bf16(const float *x, const float *weight, const float *bias, float *y, void *y_bf16, void *image, int rows, int C,
                        float eps, halo_stream_t stream) {
    HALO_CHECK_ARG(x && weight && y_bf16 && rows > 0 && C > 0 && C % 8 == 0 && (!image || C % TK == 0));
    HALO_CHECK_ARG(((uintptr_t)x | (uintptr_t)y_bf16 | (uintptr_t)y | (uintptr_t)image) % 16 == 0);
    return layernorm_image_launch(x, weight, bias, y, image, (__bf16 *)y_bf16, rows, C, eps, (hipStream_t)stream);
}

int halo_gemm_split(const void *a_image, const void *b_image, int M, int N, int K, float *C, int ldc, const float *bias1,
                    const float *bias2, int flags, float p_drop, uint64_t seed, uint32_t stream_id, uint32_t offset,
                    const uint32_t *offset_dev, halo_stream_t stream) {
    HALO_CHECK_ARG(a_image && b_image && C && M > 0 && N > 0 && K > 0 && ldc >= N);
    const DropoutCfg d = make_dropout(p_drop, seed, stream_id, offset, offset_dev);
    return halo_gemm_bf16x3_tiled(a_image, b_image, M, N, K, C, ldc, bias1, bias2, flags & 15, &d,
                                  (hipStream_t)stream);
}

int halo_gemm_split_residual(const void *a_image, const void *b_image, int M, int N, int K, float *C, int ldc, const float *residual,
                             int ldr, const float *bias1, const float *bias2, int flags, float p_drop, uint64_t seed,
                             uint32_t stream_id, uint32_t offset, const uint32_t *offset_dev, halo_stream_t stream) {
    HALO_CHECK_ARG(a_image && b_image && C && residual && M > 0 && N > 0 && K > 0 && ldc >= N && ldr >= N);
    const DropoutCfg d = make_dropout(p_drop, seed, stream_id, offset, offset_dev);
    return gemm_bf16x3_tiled_impl(a_image, b_image, M, N, K, C, ldc, bias1, bias2, (flags & 15) | HALO_GEMM_ACCUM, &d, nullptr,
                                  (hipStream_t)stream, residual, ldr);
}

int halo_gemm_split_io(const void *a_image, const void *a_hi, const void *a_lo, long lda, const void *b_image, int M, int N, int K, float *C,
                       int ldc, void *out_hi, void *out_lo, long ldo, const float *residual, int ldr, const float *bias1,
                       const float *bias2, int flags, halo_stream_t stream) {
    HALO_CHECK_ARG(b_image && M > 0 && N > 0 && K > 0 && (a_image != nullptr) != (a_hi != nullptr) && (C || out_hi));
    HALO_CHECK_ARG(!C || ldc >= N);
    HALO_CHECK_ARG(halo_math_mode() != HALO_MATH_F32);
    const bool x3 = halo_math_mode() != HALO_MATH_BF16;
    if (a_hi) HALO_CHECK_ARG(K % 32 == 0 && lda >= K && lda % 8 == 0 && (uintptr_t)a_hi % 16 == 0 && (!x3 || (a_lo && (uintptr_t)a_lo % 16 == 0)));
    if (out_hi) HALO_CHECK_ARG(ldo >= N && (!x3 || out_lo));
    HALO_CHECK_ARG(!(flags & HALO_GEMM_ACCUM) || ((residual || C) && (!residual || ldr >= N)));
    return gemm_bf16x3_io(a_image, a_hi, a_lo, lda, b_image, M, N, K, C, ldc, out_hi, out_lo, ldo, residual, ldr, bias1, bias2, flags & 15,
                          (hipStream_t)stream);
}

int halo_image_pairs(int n, const float *const *src, const int *rows, const int *cols, const long *ld, void *const *image_rows,
                     void *const *image_cols, halo_stream_t stream) {
    HALO_CHECK_ARG(n >= 0 && (n == 0 || (src && rows && cols && ld && image_rows && image_cols)));
    HaloPrepJob jobs[PREP_MAX_JOBS];
    for (int done = 0; done < n;) {                    // PREP_MAX_JOBS matrices per launch
        int m = 0;
        for (; m < PREP_MAX_JOBS && done + m < n; ++m) {
            const int i = done + m;
            HALO_CHECK_ARG(src[i] && rows[i] > 0 && cols[i] > 0 && ld[i] >= cols[i] && image_rows[i] && image_cols[i]);
            HALO_CHECK_ARG(((uintptr_t)image_rows[i] | (uintptr_t)image_cols[i]) % 16 == 0);
            HALO_CHECK_ARG(ld[i] <= 0x7fffffffL);
            jobs[m] = HaloPrepJob{2, src[i], rows[i], cols[i], (int)ld[i], image_rows[i], image_cols[i]};
        }
        const int rc = halo_prep_jobs(jobs, m, (hipStream_t)stream);
        if (rc != HALO_OK) return rc;
        done += m;
    }
    return HALO_OK;
}

}  // extern "C"
