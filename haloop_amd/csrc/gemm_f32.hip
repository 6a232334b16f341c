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
#include "halo_common.h"

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

    const int tile_m = blockIdx.x / p.tiles_n, tile_n = blockIdx.x % p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
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
    const int nk = (p.K + BK - 1) / BK;

    IOA::load(p.A, p.lda, m0, 0, p.M, p.K, p.a_vec, ra);
    IOB::load(p.B, p.ldb, n0, 0, p.N, p.K, p.b_vec, rb);
    IOA::store(As0, ra);
    IOB::store(Bs0, rb);
    __syncthreads();

    for (int t = 0; t < nk; ++t) {
        const int cur = t & 1;
        if (t + 1 < nk) {
            IOA::load(p.A, p.lda, m0, (t + 1) * BK, p.M, p.K, p.a_vec, ra);
            IOB::load(p.B, p.ldb, n0, (t + 1) * BK, p.N, p.K, p.b_vec, rb);
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
            float bias = 0.f;
            if (p.bias1) bias += p.bias1[col];
            if (p.bias2) bias += p.bias2[col];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
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

template <bool A_KC, bool B_KC>
int launch_gemm(GemmArgs &p, hipStream_t st) {
    // pick the tile so that the grid covers the chip (256 CUs) when the problem allows it
    const long tiles128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
    if (tiles128 >= 200 && p.M >= 128 && p.N >= 128) {
        p.tiles_n = (p.N + 127) / 128;
        hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC, 128, 128>), dim3((unsigned)tiles128), dim3(256), 0, st, p);
    } else {
        p.tiles_n = (p.N + 63) / 64;
        const long tiles = (long)((p.M + 63) / 64) * p.tiles_n;
        hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC, 64, 64>), dim3((unsigned)tiles), dim3(256), 0, st, p);
    }
    return halo_launch_status();
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
    p.relu = (flags & HALO_GEMM_RELU) ? 1 : 0;
    p.use_drop = p_drop > 0.f;
    p.drop = make_dropout(p_drop, seed, stream_id, offset, offset_dev);
    p.tiles_n = 0;
    hipStream_t st = (hipStream_t)stream;
    if (a_kcontig && b_kcontig) return launch_gemm<true, true>(p, st);
    if (a_kcontig && !b_kcontig) return launch_gemm<true, false>(p, st);
    if (!a_kcontig && b_kcontig) return launch_gemm<false, true>(p, st);
    return launch_gemm<false, false>(p, st);
}
