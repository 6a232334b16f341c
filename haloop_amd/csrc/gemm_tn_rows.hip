// Weight-gradient products C [M][N] = A^T B, A [K][M] and B [K][N] ROW-MAJOR bf16 with the contraction along the ROWS (dW = dy^T x of a
// Linear: K = the B T token rows), on 256 x 128 or 256 x 256 tiles of whole K -- gemm_rows.h's machine (LDS-DMA ring, eight waves,
// transposed accumulators, counted vmcnt, two wave groups one barrier apart) with k-major operands:
//
//   * why: gemm_bf16x3.hip's TN product cuts the four weight gradients of a GPT block into 432 tiles of 128 x 128 -- 1.69 per CU, 64 FLOP
//     per operand byte, and a CU takes in ~56-60 GB/s from L2 whatever the kernel (profiles/r05_experiments.md): 151 us per layer, 33 % MFMA
//     busy.  256 x 128 tiles are 216 -- one per CU -- at 85 FLOP per byte; 256 x 256 (the lm_head's 50304 x 768 gradient) 128 FLOP per byte.
//   * staging: a k-block is 32 ROWS of the operands: A 32 x 256 columns (512 B per row), B 32 x 32 TN columns.  One LDS-DMA instruction
//     (64 lanes x 16 B, lane-linear destination) fetches two whole A rows / 1 KiB of B rows: full 128-byte lines, nothing to transpose on
//     the way in.
//   * fragments: the MFMA wants, per lane, eight consecutive k of ONE column -- a column of the k-major LDS image: ds_read_b64_tr_b16
//     (16 lanes read 4 rows x 16 columns and receive them transposed), two per operand fragment.  All rows of an operand start at the same
//     LDS bank (512 / 256-byte pitch), so the 16-byte chunks of row r sit XOR-swizzled by 4 (r & 3) (permuted on the SOURCE address of the
//     DMA, the same permutation on the read): the 4 rows x 2 lane groups of a transposing read land on 8 distinct 32-byte bank spans.
//   * a wave multiplies two 32-row blocks by half the tile's columns: 2 + TN / 2 fragments per k-step for TN MFMAs.
//   * one tile = one workgroup; tiles that do not fill a round of the CUs run as K-slices into scratch slabs + a sum launch in slice order
//     (plan_slices); up to four products of one K per launch; fp32 results.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <utility>
#include "halo_common.h"
#include "halo_internal.h"

#ifndef TNR_LAB
#define TNR_LAB 0          // measurement builds (wrong results): 1 no LDS-DMA issues in the loop, 2 no fragment reads after the first k-block, 8 no MFMAs
#endif

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

constexpr int TNR_MAX = 4;
constexpr int A_BYTES = 16384;               // 32 k-rows x 256 columns x 2 B

struct TnrArgs {
    const __bf16 *a[TNR_MAX], *b[TNR_MAX];
    float *c[TNR_MAX];
    long lda[TNR_MAX], ldb[TNR_MAX], ldc[TNR_MAX];
    int M[TNR_MAX], N[TNR_MAX], tiles_n[TNR_MAX], first[TNR_MAX];
    int n, KT, nitems;
    // the LAST n_split tiles of the group run as S = slices K-slices each (the tail of a launch whose tiles do not fill their last round of the
    // CUs: 591 tiles on 256 CUs are two rounds and 79 tiles -- as 237 thirds they finish in a third of a round): workgroups [0, n_full) take
    // whole tiles, workgroup n_full + i slice i % S of tile n_full + i / S and writes its plain sums to slab[(i / S) * S + i % S] (256 x BN
    // floats each, in the scratch lent by halo_set_scratch); a second launch adds a tile's slabs in slice order
    int n_full, slices, ktper;
    float *slab;
};

template <int TN> struct Cfg {
    static constexpr int BN = 32 * TN;
    static constexpr int KPH = 2;                          // 16-deep k-steps per phase: one phase = the whole k-block (8 or 16 MFMAs per wave)
    static constexpr int NPH = 2 / KPH;
    static constexpr int BROW = 64 * TN;                   // bytes of a B row in the slot
    static constexpr int BPIECES = 32 * BROW / 1024;       // 2 TN
    static constexpr int NPIECE = 16 + BPIECES;
    static constexpr int P = (NPIECE + 7) / 8;
    static constexpr int SLOT = A_BYTES + 32 * BROW;
    static constexpr int NS = TN == 8 ? 4 : 6;
    static constexpr int D = NS - 2;
    static constexpr int LDS_BYTES = NS * SLOT;
    static_assert(TN == 4 || TN == 8, "tile columns: 128 or 256");
    static_assert(NPIECE % 8 == 0, "the pieces deal evenly over the eight waves");
    static_assert(LDS_BYTES <= 160 * 1024 && (D - 1) * P <= 63, "LDS, vmcnt");
};

__device__ __forceinline__ void dma16(const char *src, char *lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src, (__attribute__((address_space(3))) void *)lds_dst, 16, 0, 0);
}
__device__ __forceinline__ int xcd_order(int t, int n) {   // (as gemm_rows.h: consecutive workgroups -> consecutive XCDs; contiguous runs per XCD)
    const int rank = t & 7, k = t >> 3, q = n >> 3, r = n & 7;
    return (rank < r ? rank * (q + 1) : r * (q + 1) + (rank - r) * q) + k;
}
template <typename F, int... I> __device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F> __device__ __forceinline__ void static_for(F &&f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }   // f(integral_constant<int, i>), i = 0 .. N-1
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int TN>
__global__ __launch_bounds__(512) void gemm_tn_rows_kernel(const TnrArgs g) {
    using K = Cfg<TN>;
    constexpr int P = K::P, NS = K::NS, D = K::D, SLOT = K::SLOT, KPH = K::KPH, NPH = K::NPH, BROW = K::BROW, CB = 4 * TN;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int lr = lane & 31, lh = lane >> 5;
    // item -> (product, tile row, tile column): items are product-major, tile-row-major, so an XCD's contiguous run shares A panels
    const bool split = (int)blockIdx.x >= g.n_full;
    const int sidx = split ? xcd_order((int)blockIdx.x - g.n_full, (g.nitems - g.n_full) * g.slices) : 0;      // (an XCD's run: slices of neighbouring tiles)
    const int item = split ? g.n_full + sidx / g.slices : xcd_order((int)blockIdx.x, g.n_full), kslice = split ? sidx % g.slices : 0;
    const int kb0 = split ? kslice * g.ktper : 0;
    int q = 0;
#pragma unroll
    for (int i = 1; i < TNR_MAX; ++i)
        if (i < g.n && item >= g.first[i]) q = i;
    const int local = item - g.first[q], tile_m = local / g.tiles_n[q], tile_n = local % g.tiles_n[q];
    const int M = g.M[q], N = g.N[q], KT = split ? min(g.ktper, g.KT - kb0) : g.KT;
    const long lda = g.lda[q], ldb = g.ldb[q];

    // ---- LDS-DMA sources.  Piece pc < 16: k-rows 2 pc, 2 pc + 1 of the A block (32 chunks of 16 B each); else piece pb = pc - 16 of the B
    // block: 1024 / BROW rows of CB chunks.  LDS position (row r, chunk p) holds the operand's chunk p ^ 4 (r & 3); a chunk past the
    // operand's last column re-reads the last one (its products land in result columns that are never stored)
    const char *src[P];
    long kstride[P];
    int dst[P];
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const int pc = wave + 8 * i;                         // (NPIECE % 8 == 0: every wave has exactly P pieces)
        if (pc < 16) {
            const int r = 2 * pc + (lane >> 5), chunk = (lane & 31) ^ (4 * (r & 3));
            const int m0 = min(tile_m * 256 + 8 * chunk, M - 8);
            src[i] = reinterpret_cast<const char *>(g.a[q] + ((long)kb0 * 32 + r) * lda + m0);
            kstride[i] = 64 * lda;                           // 32 rows x lda elements x 2 B
            dst[i] = pc * 1024;
        } else {
            const int pb = pc - 16, r = pb * (1024 / BROW) + lane / CB, chunk = (lane % CB) ^ (4 * (r & 3));
            const int n0 = min(tile_n * K::BN + 8 * chunk, N - 8);
            src[i] = reinterpret_cast<const char *>(g.b[q] + ((long)kb0 * 32 + r) * ldb + n0);
            kstride[i] = 64 * ldb;
            dst[i] = A_BYTES + pb * 1024;
        }
    }
    auto issue = [&](int i, int kb, int slot) { dma16(src[i] + (long)min(kb, KT - 1) * kstride[i], lds + slot * SLOT + dst[i]); };

    // ---- fragment addresses: a transposing read of lane i = lane & 15 of its 16-lane group takes 8 bytes of row base + (i >> 2) at element
    // 4 (i & 3) of the group's 16 columns and returns column i of those four rows.  k-step ks, half h: base = 16 ks + 8 lh + 4 h.
    // Of a read's address only (slot) + (8 lh + rp) rows + the swizzled chunk depend on the lane or the iteration: ONE VALU add per operand
    // block and k-block (slot base + abase / bbase[t]); the k-step and the half -- rows +16 ks, +4 h -- are the instruction's immediate offset.
    const int i16 = lane & 15, grp = (lane >> 4) & 1, rp = i16 >> 2, clow = 2 * grp + ((i16 & 3) >> 1), sub = (i16 & 1) * 8;
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)lds;
    // A wave multiplies 64 result rows (two 32-row blocks) by HALF the tile's columns -- waves 0-3 the left half, 4-7 the right: 2 + TN / 2
    // fragments per k-step for TN MFMAs (one row block x all columns would be 1 + TN: every wave re-reading every B fragment).
    constexpr int TH = TN / 2;
    const int wr = wave & 3, wc = wave >> 2;
    unsigned abase[2], bbase[TH];
#pragma unroll
    for (int i = 0; i < 2; ++i) abase[i] = lds_base + (8 * lh + rp) * 512 + ((((2 * wr + i) ^ rp) & 7) << 6) + (clow << 4) + sub;     // chunk 4 ((2 wr + i) ^ rp) + clow
#pragma unroll
    for (int t = 0; t < TH; ++t) bbase[t] = lds_base + (8 * lh + rp) * BROW + (((wc * TH + t) ^ rp) << 6) + (clow << 4) + sub;         // chunk 4 ((wc TH + t) ^ rp) + clow
    // (inline assembly, not the builtin: the compiler cannot see that these reads never touch a ring slot an LDS-DMA is still filling, and
    //  puts vmcnt(0) in front of the builtin's reads -- every k-block would wait for the whole prefetch.  The reads are waited for by the
    //  explicit lgkmcnt(0) behind the phase's first barrier.)
    auto tr8 = [&](unsigned addr, auto off0, auto off1) {
        bf16x4 x0, x1;
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(x0) : "v"(addr), "n"(decltype(off0)::value) : "memory");
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(x1) : "v"(addr), "n"(decltype(off1)::value) : "memory");
        return bf16x8{x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
    };

    f32x16 acc[2][TH];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < TH; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.f;

    // ---- prologue and loop: gemm_rows.h's schedule (k-block j + D issued during k-block j; vmcnt((D - 1) P) in front of the phase's first
    // barrier retires this wave's loads up to k-block j + 1; waves 4-7 one barrier behind waves 0-3)
#pragma unroll
    for (int kb = 0; kb < D; ++kb)
#pragma unroll
        for (int i = 0; i < P; ++i) issue(i, kb, kb);
    wait_vm<(D - 1) * P>();
    __builtin_amdgcn_s_barrier();
    if (wave >= 4) __builtin_amdgcn_s_barrier();
    int slot = 0, pslot = D;
    for (int j = 0; j < KT; ++j) {
        const unsigned cur = slot * SLOT;
        const unsigned acur[2] = {abase[0] + cur, abase[1] + cur};
        unsigned bcur[TH];
#pragma unroll
        for (int t = 0; t < TH; ++t) bcur[t] = bbase[t] + cur;
        static_for<NPH>([&](auto phc) {
            constexpr int ph = decltype(phc)::value;
            bf16x8 fa[KPH][2], fb[KPH][TH];
            static_for<KPH>([&](auto uc) {
                constexpr int u = decltype(uc)::value, ks = ph * KPH + u;
                if (!(TNR_LAB & 2) || j == 0) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        fa[u][i] = tr8(acur[i], std::integral_constant<int, 16 * 512 * ks>{}, std::integral_constant<int, 16 * 512 * ks + 4 * 512>{});
#pragma unroll
                    for (int t = 0; t < TH; ++t)
                        fb[u][t] = tr8(bcur[t], std::integral_constant<int, A_BYTES + 16 * BROW * ks>{}, std::integral_constant<int, A_BYTES + 16 * BROW * ks + 4 * BROW>{});
                } else {
#pragma unroll
                    for (int i = 0; i < 2; ++i) asm volatile("" : "=v"(fa[u][i]));
#pragma unroll
                    for (int t = 0; t < TH; ++t) asm volatile("" : "=v"(fb[u][t]));
                }
            });
#pragma unroll
            for (int i = (ph * P + NPH - 1) / NPH; i < ((ph + 1) * P + NPH - 1) / NPH; ++i)
                if (!(TNR_LAB & 1)) issue(i, j + D, pslot);
            if (ph == NPH - 1 && !(TNR_LAB & 1)) wait_vm<(D - 1) * P>();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int u = 0; u < KPH; ++u)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int t = 0; t < TH; ++t) {
                        if (!(TNR_LAB & 8)) acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[u][t], fa[u][i], acc[i][t], 0, 0, 0);
                        else asm volatile("" : "+v"(acc[i][t]) : "v"(fb[u][t]), "v"(fa[u][i]));
                    }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
        });
        slot = slot + 1 == NS ? 0 : slot + 1;
        pslot = pslot + 1 == NS ? 0 : pslot + 1;
    }
    if (wave < 4) __builtin_amdgcn_s_barrier();
    wait_vm<0>();

    // ---- epilogue (gemm_rows.h's fp32 form): this lane's result row m; element (t, r): column 32 t + (r & 3) + 8 (r >> 2) + 4 lh of the tile
    // this lane's result rows 64 wr + 32 i + lr; element (i, t, r): column 32 (wc TH + t) + (r & 3) + 8 (r >> 2) + 4 lh of the tile
    const int colw = 32 * wc * TH + 4 * lh;
    if (split) {                                             // the whole 256 x BN tile of plain sums (the sum launch knows the bounds)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float *srow = g.slab + ((long)sidx * 256 + 64 * wr + 32 * i + lr) * K::BN + colw;
#pragma unroll
            for (int t = 0; t < TH; ++t)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
                    *reinterpret_cast<f32x4v *>(srow + 32 * t + 8 * gq) = f32x4v{acc[i][t][4 * gq], acc[i][t][4 * gq + 1], acc[i][t][4 * gq + 2], acc[i][t][4 * gq + 3]};
        }
        return;
    }
    const int ncol0 = tile_n * K::BN + colw;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = tile_m * 256 + 64 * wr + 32 * i + lr;
        if (m >= M) continue;
        float *crow = g.c[q] + (long)m * g.ldc[q] + ncol0;
#pragma unroll
        for (int t = 0; t < TH; ++t)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const f32x4v v = {acc[i][t][4 * gq], acc[i][t][4 * gq + 1], acc[i][t][4 * gq + 2], acc[i][t][4 * gq + 3]};
                if (ncol0 + 32 * t + 8 * gq < N) *reinterpret_cast<f32x4v *>(crow + 32 * t + 8 * gq) = v;
            }
    }
}

// C tile <- the sum of its S slabs in slice order; grid (split tiles, 16 row groups of 16), a thread = 4 columns of 16 rows
template <int TN>
__global__ __launch_bounds__(256) void tn_rows_slab_sum_kernel(const TnrArgs g) {
    constexpr int BN = 32 * TN;
    const int item = g.n_full + (int)blockIdx.x;
    int q = 0;
#pragma unroll
    for (int i = 1; i < TNR_MAX; ++i)
        if (i < g.n && item >= g.first[i]) q = i;
    const int local = item - g.first[q], tile_m = local / g.tiles_n[q], tile_n = local % g.tiles_n[q];
    const float *slab = g.slab + (long)blockIdx.x * g.slices * 256 * BN;
    for (int u = threadIdx.x; u < 16 * (BN / 4); u += 256) {
        const int r = (int)blockIdx.y * 16 + u / (BN / 4), c = (u % (BN / 4)) * 4;
        const int m = tile_m * 256 + r, n = tile_n * BN + c;
        if (m >= g.M[q] || n >= g.N[q]) continue;
        f32x4v sum = *reinterpret_cast<const f32x4v *>(slab + (long)r * BN + c);
        for (int k = 1; k < g.slices; ++k) sum += *reinterpret_cast<const f32x4v *>(slab + ((long)k * 256 + r) * BN + c);
        *reinterpret_cast<f32x4v *>(g.c[q] + (long)m * g.ldc[q] + n) = sum;
    }
}

void *halo_scratch_ptr() { void *p; size_t b; halo_get_scratch(&p, &b); return p; }

// Which tiles of a launch of `nitems` tiles run as S K-slices (+ the sum launch): ALL of them when they fill at most half a round of the CUs
// (a GPT block's four gradients on 256 x 256 tiles are 108), else the last round's when that is at most half full (the lm_head's 591 = two
// rounds and 79); slices of at least 16 k-blocks, at most four, and only with scratch for the slabs (halo_set_scratch).  S = 1: none.
void plan_slices(int nitems, int KT, int BN, int &n_full, int &S) {
    n_full = nitems; S = 1;
    const char *e = getenv("HALO_GEMM_TN_ROWS_TAIL");
    if (e && atoi(e) == 0) return;
    const int cus = halo_cu_count();
    int n_split = 0;
    if (2 * nitems <= cus) n_split = nitems;
    else if (nitems > cus && nitems % cus > 0 && 2 * (nitems % cus) <= cus) n_split = nitems % cus;
    if (!n_split) return;
    const int s = min(min(4, cus / n_split), KT / 16);
    void *scratch; size_t bytes;
    halo_get_scratch(&scratch, &bytes);
    if (s >= 2 && scratch && bytes >= (size_t)n_split * s * 256 * BN * sizeof(float)) { n_full = nitems - n_split; S = s; }
}

template <int TN>
hipError_t launch(TnrArgs &g, hipStream_t st, int slot_id) {
    if (!halo_func_attr_done(slot_id)) {
        const hipError_t e = hipFuncSetAttribute((const void *)gemm_tn_rows_kernel<TN>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg<TN>::LDS_BYTES);
        if (e != hipSuccess) return e;
        halo_func_attr_set(slot_id);
    }
    g.n_full = g.nitems; g.slices = 1; g.ktper = g.KT; g.slab = nullptr;
    int n_full, S;
    plan_slices(g.nitems, g.KT, Cfg<TN>::BN, n_full, S);
    if (S > 1) { g.n_full = n_full; g.slices = S; g.ktper = (g.KT + S - 1) / S; g.slab = (float *)halo_scratch_ptr(); }
    const int split_wgs = (g.nitems - g.n_full) * g.slices;
    hipLaunchKernelGGL((gemm_tn_rows_kernel<TN>), dim3((unsigned)(g.n_full + split_wgs)), dim3(512), Cfg<TN>::LDS_BYTES, st, g);
    if (split_wgs) hipLaunchKernelGGL((tn_rows_slab_sum_kernel<TN>), dim3((unsigned)(g.nitems - g.n_full), 16), dim3(256), 0, st, g);
    return hipGetLastError();
}

// tile columns for a group: the rounds of the CUs its whole tiles take + the sliced tiles' share of a round (+ a quarter for the sum launch),
// x the time of a tile (a 256 x 256 tile streams 32 KiB per k-block, a 256 x 128 one 24)
int pick_tn(int n, const int *M, const int *N, int KT) {
    const char *e = getenv("HALO_GEMM_TN_ROWS_TN");
    if (e && (atoi(e) == 4 || atoi(e) == 8)) return atoi(e);
    const int cus = halo_cu_count();
    double best = -1;
    int pick = 4;
    for (int tn : {4, 8}) {
        long items = 0;
        for (int i = 0; i < n; ++i) items += (long)((M[i] + 255) / 256) * ((N[i] + 32 * tn - 1) / (32 * tn));
        int n_full, S;
        plan_slices((int)items, KT, 32 * tn, n_full, S);
        const double cost = (tn == 4 ? 3.0 : 4.0) * ((n_full + cus - 1) / cus + (S > 1 ? 1.0 / S + 0.25 : 0.0));
        if (best < 0 || cost < best) { best = cost; pick = tn; }
    }
    return pick;
}

}  // namespace

extern "C" {

int halo_gemm_tn_rows_supported(int n, const int *M, const int *N, int K) {
    if (halo_math_mode() != HALO_MATH_BF16 || n < 1 || n > TNR_MAX || !M || !N || K <= 0 || K % 32) return 0;
    for (int i = 0; i < n; ++i)
        if (M[i] < 8 || N[i] < 8 || M[i] % 8 || N[i] % 8) return 0;
    return 1;
}

/* measured on MI355X (tools/time_tn_group.py): the lm_head's gradient (591 tiles of 256 x 256) 644 us against 752 on 128 x 128 tiles; a GPT
 * block's four (216 tiles of 256 x 128: one round) 145-154 against 147 -- the group launch of gemm_bf16x3.hip keeps those */
int halo_gemm_tn_rows_preferred(int n, const int *M, const int *N, int K) {
    if (!halo_gemm_tn_rows_supported(n, M, N, K)) return 0;
    long t8 = 0;
    for (int i = 0; i < n; ++i) t8 += (long)((M[i] + 255) / 256) * ((N[i] + 255) / 256);
    int n_full, S;
    plan_slices((int)t8, K / 32, 256, n_full, S);
    return t8 >= 2L * halo_cu_count() || (n_full == 0 && S > 1 && 4 * t8 * S >= 3L * halo_cu_count());    // (sliced: at least three quarters of a round)
}

int halo_gemm_tn_rows_group(int n, const void *const *a, const long *lda, const void *const *b, const long *ldb, const int *M, const int *N, int K,
                            float *const *C, const long *ldc, halo_stream_t stream) {
    HALO_CHECK_ARG(n >= 1 && n <= TNR_MAX && a && lda && b && ldb && M && N && C && ldc && K > 0 && K % 32 == 0);
    if (halo_math_mode() != HALO_MATH_BF16) return HALO_ENOTSUP;
    TnrArgs g = {};
    g.n = n; g.KT = K / 32;
    const int tn = pick_tn(n, M, N, K / 32);
    int items = 0;
    for (int i = 0; i < n; ++i) {
        HALO_CHECK_ARG(a[i] && b[i] && C[i] && M[i] >= 8 && N[i] >= 8 && M[i] % 8 == 0 && N[i] % 8 == 0);
        HALO_CHECK_ARG(lda[i] >= M[i] && ldb[i] >= N[i] && ldc[i] >= N[i] && lda[i] % 8 == 0 && ldb[i] % 8 == 0 && ldc[i] % 4 == 0);
        HALO_CHECK_ARG(((uintptr_t)a[i] | (uintptr_t)b[i] | (uintptr_t)C[i]) % 16 == 0);
        g.a[i] = (const __bf16 *)a[i]; g.b[i] = (const __bf16 *)b[i]; g.c[i] = C[i];
        g.lda[i] = lda[i]; g.ldb[i] = ldb[i]; g.ldc[i] = ldc[i]; g.M[i] = M[i]; g.N[i] = N[i];
        g.tiles_n[i] = (N[i] + 32 * tn - 1) / (32 * tn);
        g.first[i] = items;
        items += ((M[i] + 255) / 256) * g.tiles_n[i];
    }
    g.nitems = items;
    const hipError_t e = tn == 8 ? launch<8>(g, (hipStream_t)stream, 38) : launch<4>(g, (hipStream_t)stream, 39);
    return e == hipSuccess ? HALO_OK : HALO_ELAUNCH;
}

}  // extern "C"
