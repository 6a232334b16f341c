// Exact-f32 GEMM on v_mfma_f32_32x32x2_f32 (gfx950).  C[M,N] = opA(A) * opB(B) + epilogue.
//
// Tile BM x BN x 32, 256 threads = 4 waves in a 2x2 grid, each wave (BM/2) x (BN/2) of 32x32 MFMA
// tiles.  Operands are staged global -> registers -> LDS (double buffered, one barrier per
// K-step).  The LDS image depends on which dimension of the operand is contiguous in memory:
//   K-contiguous operand  -> [rows][32+1]  (float4 global loads along K, padded rows so that the
//                                           MFMA operand read -- 32 consecutive rows, one k -- is
//                                           bank-conflict free)
//   row-contiguous operand -> [32][rows]   (float4 global loads along the row dimension, b128 LDS
//                                           stores, operand read = 32 consecutive floats)
// The f32 MFMA is an exact k-ordered fmaf chain (MI355X_MICROARCH.md, Matrix cores), so results
// differ from a CPU sgemm only by summation order.
#include <stdlib.h>
#include "halo_common.h"
#include "halo_internal.h"

namespace {

constexpr int BK = 32;

struct GemmArgs {
    const float *A;
    const float *B;
    float *C;
    const float *bias1;
    const float *bias2;
    int M, N, K;
    int lda, ldb, ldc;
    int a_vec, b_vec;   // float4 global loads legal (alignment + leading dimension)
    int relu;
    int tiles_n;
    DropoutCfg drop;
    int use_drop;
    int ntiles;         // output tiles; grid = ntiles * ksplit
    int ksplit, kper;   // split-K: slice s covers k in [s*kper, min(K, (s+1)*kper)), raw sums go to slab[s]
    float *slab;
};

// One operand tile loader.  KC: memory is [rows][K]; else memory is [K][rows].
template <bool KC, int ROWS>
struct TileIO {
    static constexpr int UNITS = ROWS * BK / 4;          // float4 units per tile
    static constexpr int PER_THREAD = UNITS / 256;
    static constexpr int LDS_FLOATS = KC ? ROWS * (BK + 1) : BK * ROWS;
    static_assert(PER_THREAD >= 1, "tile too small for 256 threads");

    __device__ static __forceinline__ void unit_coords(int u, int &r, int &k) {
        if (KC) { r = u / (BK / 4); k = (u % (BK / 4)) * 4; }
        else    { k = u / (ROWS / 4); r = (u % (ROWS / 4)) * 4; }
    }

    __device__ static __forceinline__ void load(const float *base, int ld, int row0, int k0, int nrows, int K,
                                                bool vec, f32x4 (&regs)[PER_THREAD]) {
#pragma unroll
        for (int i = 0; i < PER_THREAD; ++i) {
            int r, k;
            unit_coords(threadIdx.x + 256 * i, r, k);
            const int gr = row0 + r, gk = k0 + k;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (KC) {
                if (gr < nrows) {
                    const float *p = base + (long)gr * ld + gk;
                    if (vec && gk + 3 < K) v = *reinterpret_cast<const f32x4 *>(p);
                    else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (gk + e < K) v[e] = p[e];
                    }
                }
            } else {
                if (gk < K) {
                    const float *p = base + (long)gk * ld + gr;
                    if (vec && gr + 3 < nrows) v = *reinterpret_cast<const f32x4 *>(p);
                    else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (gr + e < nrows) v[e] = p[e];
                    }
                }
            }
            regs[i] = v;
        }
    }

    __device__ static __forceinline__ void store(float *lds, const f32x4 (&regs)[PER_THREAD]) {
#pragma unroll
        for (int i = 0; i < PER_THREAD; ++i) {
            int r, k;
            unit_coords(threadIdx.x + 256 * i, r, k);
            if (KC) {
                float *p = lds + r * (BK + 1) + k;
                p[0] = regs[i][0]; p[1] = regs[i][1]; p[2] = regs[i][2]; p[3] = regs[i][3];
            } else {
                *reinterpret_cast<f32x4 *>(lds + k * ROWS + r) = regs[i];
            }
        }
    }

    __device__ static __forceinline__ float read(const float *lds, int r, int k) {
        return KC ? lds[r * (BK + 1) + k] : lds[k * ROWS + r];
    }
};

template <bool A_KC, bool B_KC, int BM, int BN>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmArgs p) {
    using IOA = TileIO<A_KC, BM>;
    using IOB = TileIO<B_KC, BN>;
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr int TM = WM / 32, TN = WN / 32;

    __shared__ __attribute__((aligned(16))) float lds[2 * (IOA::LDS_FLOATS + IOB::LDS_FLOATS)];
    float *const As0 = lds;
    float *const Bs0 = lds + 2 * IOA::LDS_FLOATS;

    const int tile = blockIdx.x % p.ntiles, kslice = blockIdx.x / p.ntiles;
    const int tile_m = tile / p.tiles_n, tile_n = tile % p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kbeg = kslice * p.kper, kend = min(p.K, kbeg + p.kper);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, lk = lane >> 5;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 ra[IOA::PER_THREAD], rb[IOB::PER_THREAD];
    const int nk = (kend - kbeg + BK - 1) / BK;

    IOA::load(p.A, p.lda, m0, kbeg, p.M, kend, p.a_vec, ra);
    IOB::load(p.B, p.ldb, n0, kbeg, p.N, kend, p.b_vec, rb);
    IOA::store(As0, ra);
    IOB::store(Bs0, rb);
    __syncthreads();

    for (int t = 0; t < nk; ++t) {
        const int cur = t & 1;
        if (t + 1 < nk) {
            IOA::load(p.A, p.lda, m0, kbeg + (t + 1) * BK, p.M, kend, p.a_vec, ra);
            IOB::load(p.B, p.ldb, n0, kbeg + (t + 1) * BK, p.N, kend, p.b_vec, rb);
        }
        const float *as = As0 + cur * IOA::LDS_FLOATS, *bs = Bs0 + cur * IOB::LDS_FLOATS;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = IOA::read(as, wm * WM + i * 32 + lr, kk + lk);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = IOB::read(bs, wn * WN + j * 32 + lr, kk + lk);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (t + 1 < nk) {
            IOA::store(As0 + (cur ^ 1) * IOA::LDS_FLOATS, ra);
            IOB::store(Bs0 + (cur ^ 1) * IOB::LDS_FLOATS, rb);
        }
        __syncthreads();
    }

    // epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * WN + j * 32 + lr;
            if (col >= p.N) continue;
            if (p.ksplit > 1) {
                float *slab = p.slab + (long)kslice * p.M * p.N;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                    if (row < p.M) slab[(long)row * p.N + col] = acc[i][j][r];
                }
                continue;
            }
            float bias = 0.f;
            if (p.bias1) bias += p.bias1[col];
            if (p.bias2) bias += p.bias2[col];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (row >= p.M) continue;
                float v = acc[i][j][r] + bias;
                v = gemm_activation(v, p.relu);
                const long e = (long)row * p.ldc + col;
                if (p.use_drop) v *= dropout_mult(p.drop, (uint64_t)e);
                if (p.relu & 4) v += p.C[e];
                p.C[e] = v;
            }
        }
    }
}

}  // namespace

// C = epilogue(sum_s slab[s]); shared with the split-bf16 GEMM
__global__ __launch_bounds__(256) void halo_splitk_reduce_kernel(const float *__restrict__ slab, int ksplit, int M, int N,
                                                                 float *__restrict__ C, int ldc, const float *bias1,
                                                                 const float *bias2, int relu, DropoutCfg drop, int use_drop) {
    const long total = (long)M * N;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int row = (int)(e / N), col = (int)(e % N);
        float v = 0.f;
        for (int s = 0; s < ksplit; ++s) v += slab[(long)s * total + e];
        if (bias1) v += bias1[col];
        if (bias2) v += bias2[col];
        v = gemm_activation(v, relu);
        const long o = (long)row * ldc + col;
        if (use_drop) v *= dropout_mult(drop, (uint64_t)o);
        if (relu & 4) v += C[o];
        C[o] = v;
    }
}

// the same sums four columns per thread (N, ldc multiples of 4, 16-byte aligned operands, fewer than 2^31 elements): one 16-byte load
// per slab and thread, 32-bit index arithmetic, one Philox call per four dropout multipliers; the slabs are summed in the same order,
// so the results are identical
__global__ __launch_bounds__(256) void halo_splitk_reduce4_kernel(const float *__restrict__ slab, int ksplit, int M, int N,
                                                                  float *__restrict__ C, int ldc, const float *bias1,
                                                                  const float *bias2, int relu, DropoutCfg drop, int use_drop) {
    const int total4 = M * (N / 4), n4 = N / 4;
    const int u = blockIdx.x * 256 + threadIdx.x;
    if (u >= total4) return;
    const int row = u / n4, col = (u - row * n4) * 4;
    const long total = (long)M * N;
    f32x4 v = *reinterpret_cast<const f32x4 *>(slab + (long)u * 4);
    for (int s = 1; s < ksplit; ++s) {
        const f32x4 w = *reinterpret_cast<const f32x4 *>(slab + (long)s * total + (long)u * 4);
        v[0] += w[0]; v[1] += w[1]; v[2] += w[2]; v[3] += w[3];
    }
    if (bias1) { const f32x4 b = *reinterpret_cast<const f32x4 *>(bias1 + col); v[0] += b[0]; v[1] += b[1]; v[2] += b[2]; v[3] += b[3]; }
    if (bias2) { const f32x4 b = *reinterpret_cast<const f32x4 *>(bias2 + col); v[0] += b[0]; v[1] += b[1]; v[2] += b[2]; v[3] += b[3]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = gemm_activation(v[i], relu);
    float *o = C + (long)row * ldc + col;
    if (use_drop) {
        const f32x4 dm = dropout_mult4(drop, (uint64_t)((long)row * ldc + col));
        v[0] *= dm[0]; v[1] *= dm[1]; v[2] *= dm[2]; v[3] *= dm[3];
    }
    if (relu & 4) { const f32x4 c = *reinterpret_cast<const f32x4 *>(o); v[0] += c[0]; v[1] += c[1]; v[2] += c[2]; v[3] += c[3]; }
    *reinterpret_cast<f32x4 *>(o) = v;
}

int halo_splitk_reduce(const float *slab, int ksplit, int M, int N, float *C, int ldc, const float *bias1,
                       const float *bias2, int relu, const DropoutCfg &drop, int use_drop, hipStream_t st) {
    const bool vec = N % 4 == 0 && ldc % 4 == 0 && (long)M * N < (1L << 31) &&
                     (((uintptr_t)slab | (uintptr_t)C | (uintptr_t)bias1 | (uintptr_t)bias2) % 16) == 0;
    if (vec) {
        const int total4 = M * (N / 4);
        hipLaunchKernelGGL(halo_splitk_reduce4_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, st, slab, ksplit, M, N, C, ldc,
                           bias1, bias2, relu, drop, use_drop);
        return halo_launch_status();
    }
    long blocks = ((long)M * N + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(halo_splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, slab, ksplit, M, N, C, ldc, bias1,
                       bias2, relu, drop, use_drop);
    return halo_launch_status();
}

// how many K slices make an under-filled grid cover the chip, given the caller-provided scratch
int halo_pick_ksplit(long tiles, int k_steps, long out_elems) {
    { static int force = -1; if (force < 0) { const char *e = getenv("HALO_KSPLIT"); force = e ? atoi(e) : 0; } if (force > 0 && tiles < 256 && k_steps >= 8) return force; }
    if (tiles >= 512 || k_steps < 8) return 1;
    if (tiles >= 128 && k_steps < 64) return 1;
    void *scratch; size_t bytes;
    halo_get_scratch(&scratch, &bytes);
    if (!scratch) return 1;
    if (tiles >= 256) {
        // 256 .. 511 tiles are 1-2 workgroups per CU with three resident: a very long contraction gains from two slices, which fill the
        // third slot (the lm_head's input gradient [8192 x 768 x 50304], 384 tiles: 843 -> 780 us with the reduce; at K = 3072 two slices
        // LOSE, 59 -> 67 us)
        return k_steps >= 1024 && bytes >= 2 * sizeof(float) * (size_t)out_elems ? 2 : 1;
    }
    // as many slices as still fit ONE round of the 256 CUs (tiles * s <= 256): [1280 x 1024 x 4096], 80 tiles: 3 slices 45.0 us, 4 slices
    // (320 workgroups: 64 CUs hold two) 53.6 us, 6 slices 46.1 us, GEMM + reduce (tools/ksplit_probe.py)
    long s = 256 / tiles;
    // one or two slices leave a long product (K >= 2048) on 100-255 tiles with half-idle CUs or stragglers: go to ~1.7 workgroups per
    // CU instead, all co-resident (88 tiles, the LSTM input gradient: 2 or 5 slices, the same step time, interleaved x4 -- left at 2).  [768 x 3072 x 8192] (144 tiles, bf16 / bf16x3): 1 slice 85 / 185 us, 2: 77 / 173, 3: 65 / 148,
    // 4: 73 / 160; [768 x 2304 x 8192] (108 tiles): 2 slices 55.5 / 118, 4: 51.2 / 117.5
    if (tiles >= 100 && k_steps >= 64 && 448 / tiles > s) s = 448 / tiles;
    if (s > k_steps / 4) s = k_steps / 4;
    const long cap = (long)(bytes / (sizeof(float) * (size_t)out_elems));
    if (s > cap) s = cap;
    if (s > 32) s = 32;
    return s < 2 ? 1 : (int)s;
}

namespace {

template <bool A_KC, bool B_KC, int BM, int BN>
int launch_tile(GemmArgs &p, hipStream_t st) {
    p.tiles_n = (p.N + BN - 1) / BN;
    p.ntiles = ((p.M + BM - 1) / BM) * p.tiles_n;
    const int k_steps = (p.K + BK - 1) / BK;
    p.ksplit = halo_pick_ksplit(p.ntiles, k_steps, (long)p.M * p.N);
    p.kper = ((k_steps + p.ksplit - 1) / p.ksplit) * BK;
    p.ksplit = (p.K + p.kper - 1) / p.kper;
    void *scratch; size_t bytes;
    halo_get_scratch(&scratch, &bytes);
    p.slab = (float *)scratch;
    hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC, BM, BN>), dim3((unsigned)(p.ntiles * p.ksplit)), dim3(256), 0, st, p);
    int rc = halo_launch_status();
    if (rc != HALO_OK || p.ksplit == 1) return rc;
    return halo_splitk_reduce(p.slab, p.ksplit, p.M, p.N, p.C, p.ldc, p.bias1, p.bias2, p.relu, p.drop, p.use_drop, st);
}

template <bool A_KC, bool B_KC>
int launch_gemm(GemmArgs &p, hipStream_t st) {
    // pick the tile so that the grid covers the chip (256 CUs) when the problem allows it
    const long tiles128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
    if (tiles128 >= 200 && p.M >= 128 && p.N >= 128) return launch_tile<A_KC, B_KC, 128, 128>(p, st);
    return launch_tile<A_KC, B_KC, 64, 64>(p, st);
}

}  // namespace

extern "C" int halo_gemm_f32(int a_kcontig, int b_kcontig, int M, int N, int K, const float *A, int lda,
                             const float *B, int ldb, float *C, int ldc, const float *bias1,
                             const float *bias2, int flags, float p_drop, uint64_t seed, uint32_t stream_id,
                             uint32_t offset, const uint32_t *offset_dev, halo_stream_t stream) {
    HALO_CHECK_ARG(A && B && C);
    HALO_CHECK_ARG(M > 0 && N > 0 && K > 0);
    HALO_CHECK_ARG(lda >= (a_kcontig ? K : M) && ldb >= (b_kcontig ? K : N) && ldc >= N);
    GemmArgs p;
    p.A = A; p.B = B; p.C = C; p.bias1 = bias1; p.bias2 = bias2;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.a_vec = (lda % 4 == 0) && ((uintptr_t)A % 16 == 0);
    p.b_vec = (ldb % 4 == 0) && ((uintptr_t)B % 16 == 0);
    p.relu = flags & 15;
    p.use_drop = p_drop > 0.f;
    p.drop = make_dropout(p_drop, seed, stream_id, offset, offset_dev);
    p.tiles_n = 0;
    hipStream_t st = (hipStream_t)stream;
    if (a_kcontig && b_kcontig) return launch_gemm<true, true>(p, st);
    if (a_kcontig && !b_kcontig) return launch_gemm<true, false>(p, st);
    if (!a_kcontig && b_kcontig) return launch_gemm<false, true>(p, st);
    return launch_gemm<false, false>(p, st);
}
