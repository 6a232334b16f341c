// Single-pass bf16 product C [M][N] = A [M][K] x B [N][K]^T for the GPT path's activation-by-weight products (forward and input-gradient
// products of every Linear, the lm_head): A = activations, ROW-MAJOR bf16 as the producing launch left them (or a tiled image), B = a
// weight's tiled image (tiled_image.h: 128-row x 32-k blocks, hi part).  One workgroup = 256 rows x BN = 32 TN columns, TN in {3, 6, 9}:
//
//   * why these tiles: M = B T = 8192 rows give 32 tile rows; BN = 96 / 288 cut N = 768 / 2304 into exactly 8 tile columns -> 256 workgroups,
//     ONE per CU, one round, no tail (gemm_bf16x3.hip's 128 x 128 tiles number 384 / 1152 there: 1.5 rounds on three slots per CU); BN = 192
//     cuts N = 3072 into 16 (two exact rounds) and the lm_head's 50304 into 262 whole tile columns.
//   * every wave owns 32 ROWS x all BN columns (8 waves = 256 rows): per 16-deep k-step one A fragment of its own rows and the TN B fragments
//     all eight waves share -- (TN + 1) ds_read_b128 for TN MFMA 32x32x16.  The accumulators hold C TRANSPOSED (the B fragment is the MFMA's
//     first operand): a lane then carries ONE row m = lane & 31 and, per 32-column block, columns (r & 3) + 8 (r >> 2) + 4 (lane >> 5) --
//     four consecutive columns per register group, so fp32 results leave as 16-byte stores without any shuffle, bf16 results as 16-byte
//     stores after one v_permlane32_swap per dword (cdna_hip_programming.md T21), and a row's softmax statistics (the lm_head's
//     cross-entropy epilogue) are in-lane reductions plus ONE exchange with lane ^ 32.
//   * LDS ring of NS k-blocks (32 k each): [A 256 rows x 64 B | B BN rows x 64 B], filled by LDS-DMA (global_load_lds, 16 B per lane); a
//     row-major A is swizzled on the SOURCE address (rule 21: linear destination, permuted source, the same permutation on the read),
//     images are copied verbatim.  A k-block's 16 + BN / 16 one-KiB pieces are dealt over the eight waves, P = ceil(pieces / 8) each (a
//     wave whose last index runs past the end re-issues the last piece: identical bytes to the identical place).
//   * schedule = gemm256.h's: phases of {fragment reads || LDS-DMA issues -> barrier -> MFMAs -> barrier}, waves 4-7 ONE BARRIER behind
//     waves 0-3 (one wave of each group per SIMD: one computes while the other reads and issues), the prefetch in flight across barriers
//     behind a counted vmcnt once per k-block, raw s_barrier, never vmcnt(0) in the loop.  A phase is one 16-deep k-step (TN = 6, 9) or
//     the whole 32-deep k-block (TN = 3: three MFMAs would not cover a phase's reads and issues).
//   RAW: k-block j + D is issued during k-block j (D = NS - 2); the wait at the end of k-block j -- vmcnt((D - 1) P), in front of the
//   phase's FIRST barrier -- retires this wave's loads up to k-block j + 1; both groups have passed it before the faster group's first
//   read of k-block j + 1, which follows that phase's second barrier.  WAR: slot (j + D) % NS last held k-block j - 2, whose last reads
//   were retired (lgkmcnt(0) in front of the MFMAs) two phases or more before the first issue into it, by either group.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "halo_common.h"
#include "halo_internal.h"

namespace halo_gr {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

constexpr int PART = 8192;               // one 128-row x 32-k part of an image block
constexpr int BLOCK = 16384;             // image block: hi part | lo part
constexpr int A_BYTES = 16384;           // 256 rows x 64 B

enum { EPI_BF16 = 0, EPI_F32 = 1, EPI_RESID = 2, EPI_CE = 3 };

struct Args {
    const char *a_img;                   // tiled image of A [M][K], or
    const __bf16 *a_rm; long lda;        // row-major bf16 A (K % 32 == 0, lda % 8 == 0)
    const char *b_img;                   // tiled image of B [N][K]
    int M, N, KT;                        // KT: 32-deep k-blocks
    int tiles_m, tiles_n;
    int kslices, ktper;                  // EPI_F32 only: the k-blocks cut into kslices runs of ktper; workgroup b takes slice b / tiles and writes its
    long slab_stride;                    // plain sums to C + slice * slab_stride (a reduce launch adds the slabs)
    float *C; long ldc;                  // EPI_F32 / EPI_RESID: fp32 result
    const float *R; long ldr;            // EPI_RESID: C = R + A B^T (R may be C)
    __bf16 *O; long ldo;                 // EPI_BF16: bf16 result; EPI_CE: optional bf16 logits
    int act;                             // EPI_BF16: 0, or the GELU flag of gemm_activation (2 tanh form, 8 erf form): O = gelu(A B^T) ...
    __bf16 *O2;                          // ... and, when given, O2 [M][ldo] = A B^T itself (the pre-activation a backward pass keeps)
    // EPI_CE: per row and tile column the (max, sum exp) of the logits -> ce_part[(row * tiles_n + tile_n) * 2], the target's logit -> ce_tlogit[row]
    const int64_t *ce_target; float *ce_part, *ce_tlogit;
};

template <int TN> struct Cfg {
    static constexpr int BN = 32 * TN;
    static constexpr int KPH = TN >= 6 ? 1 : 2;            // 16-deep k-steps per phase
    static constexpr int NPH = 2 / KPH;                    // phases per k-block
    static constexpr int BPIECES = BN / 16;
    static constexpr int NPIECE = 16 + BPIECES;
    static constexpr int P = (NPIECE + 7) / 8;             // LDS-DMA issues per wave and k-block
    static constexpr int SLOT = A_BYTES + BN * 64;
    static constexpr int NS = TN == 9 ? 4 : (TN == 6 ? 5 : 6);
    static constexpr int D = NS - 2;                       // prefetch distance in k-blocks
    static constexpr int LDS_BYTES = NS * SLOT;
    static_assert(TN == 3 || TN == 6 || TN == 9, "tile columns: 96, 192 or 288");
    static_assert((D - 1) * P <= 63, "vmcnt is six bits");
};

__device__ __forceinline__ void dma16(const char *src, char *lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src, (__attribute__((address_space(3))) void *)lds_dst, 16, 0, 0);
}

// position of workgroup b (of n) in the XCD-contiguous order (as gemm256.h): consecutive workgroups go to consecutive XCDs, so b, b + 8, ...
// share an L2 and get a contiguous run of tiles.  Bijective for any n.
__device__ __forceinline__ int xcd_order(int t, int n) {
    const int rank = t & 7, k = t >> 3, q = n >> 3, r = n & 7;
    return (rank < r ? rank * (q + 1) : r * (q + 1) + (rank - r) * q) + k;
}

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// LAB: 0 the product; measurement variants of tools/gemm_rows_lab.hip (wrong results): 1 no LDS-DMA issues in the loop (and no counted wait),
// 2 no fragment reads in the loop, 4 no epilogue stores, 8 no MFMAs
template <int TN, int EPI, bool AIMG, int LAB = 0>
__global__ __launch_bounds__(512) void gemm_rows_kernel(const Args a) {
    using K = Cfg<TN>;
    constexpr int BN = K::BN, P = K::P, NS = K::NS, D = K::D, SLOT = K::SLOT, KPH = K::KPH, NPH = K::NPH;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int lr = lane & 31, lh = lane >> 5;
    // tile order inside an XCD's contiguous run: groups of four tile rows, walked column by column -- the ~32 tiles resident on one XCD
    // cover 4 A panels x 8 B panels (both stay in its L2) instead of one tile row x 32 columns
    const int ntile = a.tiles_m * a.tiles_n;
    const int kslice = EPI == EPI_F32 ? (int)blockIdx.x / ntile : 0;
    const int tile = xcd_order((int)blockIdx.x - kslice * ntile, ntile);
    const int grp = tile / (4 * a.tiles_n), gm0 = grp * 4, gh = min(4, a.tiles_m - gm0), ing = tile % (4 * a.tiles_n);
    const int tile_m = gm0 + ing % gh, tile_n = ing / gh;
    // this workgroup's k-blocks: [kb0, kb0 + KT) of the operands' KTall
    const int KTall = a.KT, kb0 = EPI == EPI_F32 ? kslice * a.ktper : 0;
    const int KT = EPI == EPI_F32 ? min(a.ktper, KTall - kb0) : KTall;

    // ---- LDS-DMA sources: P pieces per wave and k-block; piece pc < 16: rows 16 pc .. of the A tile, else rows 16 (pc - 16) .. of the B tile
    const char *src[P];
    int kstride[P], dst[P];
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const int pc = min(wave + 8 * i, K::NPIECE - 1);
        if (pc < 16) {
            dst[i] = pc * 1024;
            if (AIMG) {
                const int g0 = tile_m * 256 + 16 * pc, rbmax = (a.M + 127) / 128 - 1;
                src[i] = a.a_img + ((long)min(g0 >> 7, rbmax) * KTall + kb0) * BLOCK + (g0 & 127) * 64 + lane * 16;
                kstride[i] = BLOCK;
            } else {
                // lane l of the piece lands at (row 16 pc + l / 4, position l % 4) of the slot's 64-byte rows; the swizzled image keeps
                // logical chunk pos ^ ((row >> 2) & 3) there, so the lane FETCHES that chunk
                const int row = min(tile_m * 256 + 16 * pc + (lane >> 2), a.M - 1), chunk = (lane & 3) ^ ((lane >> 4) & 3);
                src[i] = reinterpret_cast<const char *>(a.a_rm + (long)row * a.lda + kb0 * 32 + chunk * 8);
                kstride[i] = 64;
            }
        } else {
            const int q = pc - 16, n0 = tile_n * BN + 16 * q, rbmax = (a.N + 127) / 128 - 1;
            dst[i] = A_BYTES + q * 1024;
            src[i] = a.b_img + ((long)min(n0 >> 7, rbmax) * KTall + kb0) * BLOCK + (n0 & 127) * 64 + lane * 16;
            kstride[i] = BLOCK;
        }
    }
    auto issue = [&](int i, int kb, int slot) { dma16(src[i] + (long)min(kb, KT - 1) * kstride[i], lds + slot * SLOT + dst[i]); };

    // fragment offsets inside a slot (k-step ks of the block: chunk 2 ks + lh, XOR-swizzled by (row >> 2) & 3): A row 32 wave + lr; B row 32 t + lr
    int aoff[2], boff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int c = ((2 * ks + lh) ^ ((lr >> 2) & 3)) << 4;
        aoff[ks] = (32 * wave + lr) * 64 + c;
        boff[ks] = A_BYTES + lr * 64 + c;
    }
    f32x16 acc[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // ---- prologue: k-blocks 0 .. D-1 in flight; k-block 0 has landed when all but the newest (D - 1) P loads have
#pragma unroll
    for (int kb = 0; kb < D; ++kb)
#pragma unroll
        for (int i = 0; i < P; ++i) issue(i, kb, kb);
    wait_vm<(D - 1) * P>();
    __builtin_amdgcn_s_barrier();
    if (wave >= 4) __builtin_amdgcn_s_barrier();             // the second group runs one barrier behind the first
    int slot = 0, pslot = D;                                 // slot of k-block j, of k-block j + D
    for (int j = 0; j < KT; ++j) {
        const char *cur = lds + slot * SLOT;
#pragma unroll
        for (int ph = 0; ph < NPH; ++ph) {
            bf16x8 fa[KPH], fb[KPH][TN];
#pragma unroll
            for (int u = 0; u < KPH; ++u) {
                const int ks = ph * KPH + u;
                if (!(LAB & 2) || j == 0) {
                    fa[u] = *reinterpret_cast<const bf16x8 *>(cur + aoff[ks]);
#pragma unroll
                    for (int t = 0; t < TN; ++t) fb[u][t] = *reinterpret_cast<const bf16x8 *>(cur + boff[ks] + t * 2048);
                } else {
                    asm volatile("" : "=v"(fa[u]));
#pragma unroll
                    for (int t = 0; t < TN; ++t) asm volatile("" : "=v"(fb[u][t]));
                }
            }
            // this phase's share of the P pieces of k-block j + D
#pragma unroll
            for (int i = (ph * P + NPH - 1) / NPH; i < ((ph + 1) * P + NPH - 1) / NPH; ++i)
                if (!(LAB & 1)) issue(i, j + D, pslot);
            if (ph == NPH - 1 && !(LAB & 1)) wait_vm<(D - 1) * P>();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int u = 0; u < KPH; ++u)
#pragma unroll
                for (int t = 0; t < TN; ++t) {
                    if (!(LAB & 8)) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[u][t], fa[u], acc[t], 0, 0, 0);
                    else asm volatile("" : "+v"(acc[t]) : "v"(fb[u][t]), "v"(fa[u]));
                }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
        }
        slot = slot + 1 == NS ? 0 : slot + 1;
        pslot = pslot + 1 == NS ? 0 : pslot + 1;
    }
    if (wave < 4) __builtin_amdgcn_s_barrier();              // (every wave has now passed the same number of barriers)
    wait_vm<0>();                                            // the clamped tail loads: drained before the workgroup retires

    // ---- epilogue: this lane's row m; element (t, r): column 32 t + (r & 3) + 8 (r >> 2) + 4 lh of the tile
    if (LAB & 4) {                                           // (keeps the accumulators alive; never true)
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < TN; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[t][r];
        if (s == 123456.789f && a.ce_part) a.ce_part[0] = s;
        return;
    }
    const int m = tile_m * 256 + 32 * wave + lr;
    const bool rowok = m < a.M;
    const int ncol0 = tile_n * BN;
    if constexpr (EPI == EPI_F32 || EPI == EPI_RESID) {
        float *crow = a.C + (long)kslice * a.slab_stride + (long)m * a.ldc + ncol0 + 4 * lh;
        const float *rrow = EPI == EPI_RESID ? a.R + (long)m * a.ldr + ncol0 + 4 * lh : nullptr;
#pragma unroll
        for (int t = 0; t < TN; ++t) {
            f32x4v res[4];
            if (EPI == EPI_RESID) {                          // the four addends of a 32-column block are requested together
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const bool ok = rowok && ncol0 + 32 * t + 8 * g + 4 * lh < a.N;
                    res[g] = ok ? *reinterpret_cast<const f32x4v *>(rrow + 32 * t + 8 * g) : f32x4v{0.f, 0.f, 0.f, 0.f};
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4v v = {acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]};
                if (EPI == EPI_RESID) v += res[g];
                if (rowok && ncol0 + 32 * t + 8 * g + 4 * lh < a.N) *reinterpret_cast<f32x4v *>(crow + 32 * t + 8 * g) = v;
            }
        }
    }
    if constexpr (EPI == EPI_CE) {
        // softmax statistics of this row over the tile's columns: in-lane over the registers, then one exchange with lane ^ 32
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < TN; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const bool ok = ncol0 + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * lh < a.N;
                mx = fmaxf(mx, ok ? acc[t][r] : -INFINITY);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float msafe = mx == -INFINITY ? 0.f : mx;
        float sm = 0.f;
#pragma unroll
        for (int t = 0; t < TN; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const bool ok = ncol0 + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * lh < a.N;
                sm += ok ? __expf(acc[t][r] - msafe) : 0.f;
            }
        sm += __shfl_xor(sm, 32, 64);
        if (rowok) {
            if (lh == 0) {
                float *pp = a.ce_part + ((long)m * a.tiles_n + tile_n) * 2;
                pp[0] = mx; pp[1] = sm;
            }
            const long tc = a.ce_target[m] - ncol0;          // the target's column inside this tile, if it is here and in this half-wave
            if (tc >= 0 && tc < BN && ((tc >> 2) & 1) == lh) {
                const int want = ((int)tc >> 5) * 16 + (((int)tc >> 3) & 3) * 4 + ((int)tc & 3);
                float tl = 0.f;
#pragma unroll
                for (int t = 0; t < TN; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) tl = (t * 16 + r == want) ? acc[t][r] : tl;
                a.ce_tlogit[m] = tl;
            }
        }
    }
    if constexpr (EPI == EPI_BF16 || EPI == EPI_CE) {
        if (EPI == EPI_CE && !a.O) return;                   // (uniform: scoring keeps no logits)
        // bf16 rows: pack column pairs, then per pair of register groups (g, g + 1) one v_permlane32_swap per dword: lanes 0-31 end up with
        // columns 8 g .. 8 g + 7 of their row, lanes 32-63 with 8 (g + 1) .. 8 (g + 1) + 7 -> one 16-byte store each (T21)
        const long ooff = (long)m * a.ldo + ncol0 + 8 * lh;
        // (every index static: a lambda instantiated per form, not a loop over the forms)
        auto store_rows = [&](__bf16 *base, auto activate_c) {
            constexpr bool activate = decltype(activate_c)::value;
            __bf16 *orow = base + ooff;
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                unsigned d[4][2];
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
                        float v0 = acc[t][4 * g + 2 * h], v1 = acc[t][4 * g + 2 * h + 1];
                        if (activate) {
                            v0 = gemm_activation((float)(__bf16)v0, a.act);
                            v1 = gemm_activation((float)(__bf16)v1, a.act);
                        }
                        const bf16x2 pk = {(__bf16)v0, (__bf16)v1};
                        d[g][h] = __builtin_bit_cast(unsigned, pk);
                    }
#pragma unroll
                for (int g = 0; g < 4; g += 2) {
                    u32x4v o;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const auto sw = __builtin_amdgcn_permlane32_swap(d[g][h], d[g + 1][h], false, false);
                        o[h] = sw[0]; o[2 + h] = sw[1];
                    }
                    if (rowok && ncol0 + 32 * t + 8 * (g + lh) < a.N) *reinterpret_cast<u32x4v *>(orow + 32 * t + 8 * g) = o;
                }
            }
        };
        // EPI_BF16 with an activation: the rounded pre-activation first when the caller keeps it, then the activation of the ROUNDED value --
        // what a separate pass over the stored bf16 rows computes
        if (EPI == EPI_BF16 && a.act) {
            if (a.O2) store_rows(a.O2, std::false_type{});
            store_rows(a.O, std::true_type{});
        } else {
            store_rows(a.O, std::false_type{});
        }
    }
}

template <int TN, int EPI, bool AIMG, int LAB = 0>
static inline hipError_t launch(const Args &a, hipStream_t st) {
    constexpr int slot = 8 + (TN / 3 - 1) * 8 + EPI * 2 + (AIMG ? 1 : 0);
    if (LAB || !halo_func_attr_done(slot)) {         // per device (halo_internal.h)
        const hipError_t e = hipFuncSetAttribute((const void *)gemm_rows_kernel<TN, EPI, AIMG, LAB>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg<TN>::LDS_BYTES);
        if (e != hipSuccess) return e;
        if (!LAB) halo_func_attr_set(slot);
    }
    const int ks = EPI == EPI_F32 && a.kslices > 1 ? a.kslices : 1;
    hipLaunchKernelGGL((gemm_rows_kernel<TN, EPI, AIMG, LAB>), dim3((unsigned)(a.tiles_m * a.tiles_n * ks)), dim3(512), Cfg<TN>::LDS_BYTES, st, a);
    return hipGetLastError();
}

}  // namespace halo_gr
