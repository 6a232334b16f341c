// Laboratory build of the 256 x 256 workgroup-tile single-pass bf16 product over the tiled operand images (csrc/tiled_image.h):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I haloop_amd/csrc tools/gemm256_lab.hip -o gpurun_out/gemm256_lab && gpurun_out/gemm256_lab
// One launch over a list of problems (the LSTM's two weight-gradient products + the carried input-gradient slices), verified against a
// plain device product, timed with HIP events.  The kernel under test is csrc/gemm256.h, the one the library links.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include "gemm256.h"

// (the library keeps these per device in abi.hip; a one-device lab binary needs no more)
static bool g_attr_done[8];
int halo_cu_count() { return 256; }
bool halo_func_attr_done(int slot) { return g_attr_done[slot]; }
void halo_func_attr_set(int slot) { g_attr_done[slot] = true; }

// ==== lab only: the four-wave variant of the 256 x 256 tile (measured equal to the eight-wave kernel the library links: DESIGN.md 3.1e) ====
namespace halo_g256 {

// ---- the same tile on FOUR waves of 128 x 128 (one wave per SIMD, the 256 accumulator registers in AGPRs) ----------------------------------
// LDS traffic per k-block: 4 waves x 16 KiB of fragment reads + 32 KiB of LDS-DMA = 96 KiB against the eight-wave kernel's 128 KiB (= the
// LDS pipe's whole bandwidth for the time the MFMAs take).  No second wave on a SIMD to cover for a waiting one, so the wave software-
// pipelines itself: the fragments of phase f + 1 are read into a second register set BETWEEN the 16 MFMAs of phase f, the LDS-DMA of
// k-block j + 4 is issued there too, and there is ONE barrier per k-block:
//   phase (j, 0): MFMAs on set 0;  reads (j, 1) -> set 1;  issue B(j + 3)... see the schedule below
//   start of phase (j, 1): lgkmcnt(0) [every read of k-block j is in registers]; vmcnt(16) [k-block j + 1 landed: behind it A, B of j + 2 and
//   j + 3 = 16 loads]; barrier B_j.  Behind B_j every wave has k-block j in registers -> slot j % 4 is free for k-block j + 4, and k-block
//   j + 1 has landed for everyone -> its first reads follow in this phase.
//   phase (j, 1): MFMAs on set 1;  reads (j + 1, 0) -> set 0;  issue A(j + 4) -> slot j % 4
//   phase (j + 1, 0): MFMAs on set 0;  reads (j + 1, 1) -> set 1;  issue B(j + 4) -> slot j % 4
// A k-block is requested three k-blocks (6 phases x 512 MFMA cycles) before its first read.
// the instruction order of one phase (hipcc would otherwise put the 16 MFMAs first and the reads behind them, where they land too late):
// one fragment read behind each of the first eight MFMAs, one LDS-DMA issue behind each of the next four
__device__ __forceinline__ void w4_phase_order() {
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // 1 DS read
    }
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // 1 VMEM read (the LDS-DMA)
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
}

template <int LAB = 0>
__global__ __launch_bounds__(256) void gemm256w4_kernel(const Args a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int wr = wave >> 1, wc = wave & 1;
    const int lr = lane & 31, lh = lane >> 5;
    int q = 0;
    if (a.nprob > 1 && (int)blockIdx.x >= a.p[1].first) q = 1;
    if (a.nprob > 2 && (int)blockIdx.x >= a.p[2].first) q = 2;
    const Prob &g = a.p[q];
    const int local = (int)blockIdx.x - g.first, ntile = g.tiles_m * g.tiles_n;
    const int kslice = local / ntile;
    const int tile = xcd_order(local % ntile, ntile);
    const int grp = tile / (4 * g.tiles_n), gm0 = grp * 4, gh = min(4, g.tiles_m - gm0), ing = tile % (4 * g.tiles_n);
    const int tile_m = gm0 + ing % gh, tile_n = ing / gh;
    const int KT = g.KT;
    const int kt0 = kslice * g.ktper, nkb = min(KT, kt0 + g.ktper) - kt0;
    // LDS-DMA: a pair of parts = 16 wave-instructions of 1 KiB, four per wave: wave w takes the 1-KiB pieces 2 w, 2 w + 1 of both parts
    const char *srcA[2] = {g.A + ((long)min(2 * tile_m, g.rbA - 1) * KT + kt0) * BLOCK, g.A + ((long)min(2 * tile_m + 1, g.rbA - 1) * KT + kt0) * BLOCK};
    const char *srcB[2] = {g.B + ((long)min(2 * tile_n, g.rbB - 1) * KT + kt0) * BLOCK, g.B + ((long)min(2 * tile_n + 1, g.rbB - 1) * KT + kt0) * BLOCK};
    const int dma_off = wave * 2048 + lane * 16;
    auto issue = [&](const char *const (&src)[2], int part0, int kb, int slot) {
        const long o = (long)min(kb, nkb - 1) * BLOCK + dma_off;
#pragma unroll
        for (int part = 0; part < 2; ++part)
#pragma unroll
            for (int u = 0; u < 2; ++u) dma16(src[part] + o + u * 1024, lds + slot * SLOT + (part0 + part) * PART + wave * 2048 + u * 1024);
    };
    int aoff[2][4], boff[2][4];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            aoff[p][i] = wr * PART + swz(32 * i + lr, 2 * p + lh);
            boff[p][i] = (2 + wc) * PART + swz(32 * i + lr, 2 * p + lh);
        }
    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jn = 0; jn < 4; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jn][r] = 0.f;

    issue(srcA, 0, 0, 0); issue(srcB, 2, 0, 0); issue(srcA, 0, 1, 1); issue(srcB, 2, 1, 1);
    issue(srcA, 0, 2, 2); issue(srcB, 2, 2, 2); issue(srcA, 0, 3, 3); issue(srcB, 2, 3, 3);
    asm volatile("s_waitcnt vmcnt(24)\n\ts_barrier" ::: "memory");
    bf16x8 f0a[4], f0b[4], f1a[4], f1b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { f0a[i] = *reinterpret_cast<const bf16x8 *>(lds + aoff[0][i]); f0b[i] = *reinterpret_cast<const bf16x8 *>(lds + boff[0][i]); }
    for (int j = 0; j < nkb; ++j) {
        const char *cur = lds + (j & 3) * SLOT, *nxt = lds + ((j + 1) & 3) * SLOT;
        // ---- phase (j, 0): MFMAs on set 0; between them the reads of (j, 1) into set 1 and the B pieces of k-block j + 3
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jn = 0; jn < 4; ++jn) {
                acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f0a[i], f0b[jn], acc[i][jn], 0, 0, 0);
                const int n = 4 * i + jn;
                if (!(LAB & 4)) {
                    if (n < 4) f1a[n] = *reinterpret_cast<const bf16x8 *>(cur + aoff[1][n]);
                    else if (n < 8) f1b[n - 4] = *reinterpret_cast<const bf16x8 *>(cur + boff[1][n - 4]);
                }
                if (n == 8 && j >= 1 && !(LAB & 1)) issue(srcB, 2, j + 3, (j + 3) & 3);
            }
        w4_phase_order();
        __builtin_amdgcn_sched_barrier(0);
        // ---- start of phase (j, 1): k-block j is in registers, k-block j + 1 has landed -- for every wave behind the barrier
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (!(LAB & 3)) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jn = 0; jn < 4; ++jn) {
                acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1a[i], f1b[jn], acc[i][jn], 0, 0, 0);
                const int n = 4 * i + jn;
                if (!(LAB & 4)) {
                    if (n < 4) f0a[n] = *reinterpret_cast<const bf16x8 *>(nxt + aoff[0][n]);
                    else if (n < 8) f0b[n - 4] = *reinterpret_cast<const bf16x8 *>(nxt + boff[0][n - 4]);
                }
                if (n == 8 && !(LAB & 1)) issue(srcA, 0, j + 4, j & 3);
            }
        w4_phase_order();
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

    const int m0 = tile_m * 256 + wr * 128, n0 = tile_n * 256 + wc * 128;
    const bool second = tile_n * 256 >= g.n_split;
    float *cq = (second ? g.C2 : g.C) + (long)kslice * g.slab_stride;
    const int ldq = second ? g.ldc2 : g.ldc, cshift = second ? g.n_split : 0;
    const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(cq, 0, 0x40000000, 0x00020000);
    constexpr int FAR = 0x7ffffff0;
    float ssj[4] = {0.f, 0.f, 0.f, 0.f};
    bool colok[4];
    int voff[4];
#pragma unroll
    for (int jn = 0; jn < 4; ++jn) {
        const int col = n0 + 32 * jn + lr;
        colok[jn] = col < g.N;
        voff[jn] = colok[jn] ? (4 * lh * ldq + col - cshift) * 4 : FAR;
    }
    const bool inner = m0 + 128 <= g.M;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + 32 * i + (r & 3) + 8 * (r >> 2);
            const int soff = row * ldq * 4;
            const bool rowok = inner || row + 4 * lh < g.M;
#pragma unroll
            for (int jn = 0; jn < 4; ++jn) {
                const float v = acc[i][jn][r];
                if (!(LAB & 8)) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), crs, rowok ? voff[jn] : FAR, soff, 0);
                ssj[jn] += rowok ? v * v : 0.f;
            }
        }
    float ss = ((colok[0] ? ssj[0] : 0.f) + (colok[1] ? ssj[1] : 0.f)) + ((colok[2] ? ssj[2] : 0.f) + (colok[3] ? ssj[3] : 0.f));
    if (g.sumsq) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) ss += __shfl_xor(ss, d, 64);
        __syncthreads();
        float *red = reinterpret_cast<float *>(lds);
        if (lane == 0) red[wave] = ss;
        __syncthreads();
        if (threadIdx.x == 0) g.sumsq[local] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}

template <int LAB = 0>
static inline hipError_t launch_w4(const Args &a, int nwg, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        const hipError_t e = hipFuncSetAttribute((const void *)gemm256w4_kernel<LAB>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return e;
        attr = true;
    }
    hipLaunchKernelGGL(gemm256w4_kernel<LAB>, dim3((unsigned)nwg), dim3(256), LDS_BYTES, st, a);
    return hipGetLastError();
}


}  // namespace halo_g256

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static unsigned short f2bf(float f) {
    unsigned u; memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static float bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

// X [R][K] row-major floats (already bf16-representable) -> tiled image (hi parts; lo parts left zero)
static void make_image(const std::vector<float> &x, int R, int K, std::vector<char> &img) {
    const int RB = (R + 127) / 128, KT = (K + 31) / 32;
    img.assign((size_t)RB * KT * 16384, 0);
    for (int r = 0; r < RB * 128; ++r)
        for (int k = 0; k < KT * 32; ++k) {
            const float v = (r < R && k < K) ? x[(size_t)r * K + k] : 0.f;
            const int rb = r / 128, kt = k / 32, rr = r % 128, c = (k % 32) / 8, e = k % 8;
            const size_t off = ((size_t)rb * KT + kt) * 16384 + rr * 64 + ((c ^ ((rr >> 2) & 3)) << 4) + e * 2;
            const unsigned short h = f2bf(v);
            memcpy(&img[off], &h, 2);
        }
}

__global__ void ref_kernel(const float *A, const float *B, float *C, int M, int N, int K) {
    const int col = blockIdx.x * 16 + threadIdx.x, row = blockIdx.y * 16 + threadIdx.y;
    if (row >= M || col >= N) return;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += A[(size_t)row * K + k] * B[(size_t)col * K + k];
    C[(size_t)row * N + col] = s;
}

struct Prob { int M, N, K, kslices; };

int main(int argc, char **argv) {
    const int TB = argc > 1 ? atoi(argv[1]) : 1344;
    std::vector<Prob> probs = {{4096, 2048, TB, 1}, {4096, 1152, TB, 1}, {TB, 128, 4096, 8}};
    if (argc > 2 && atoi(argv[2]) == 1) probs = {{8192, 3072, 768, 1}};
    if (argc > 2 && atoi(argv[2]) == 2) probs = {{4096, 4096, 4096, 1}};
    if (argc > 2 && atoi(argv[2]) == 3) probs = {{300, 200, 96, 1}, {256, 256, 64, 2}};
    halo_g256::Args a = {};
    a.nprob = (int)probs.size();
    std::vector<float *> dC(probs.size()), dRef(probs.size());
    std::vector<float *> dA(probs.size()), dB(probs.size());
    int first = 0;
    double flops = 0;
    srand(1);
    for (size_t q = 0; q < probs.size(); ++q) {
        const Prob &pr = probs[q];
        std::vector<float> A((size_t)pr.M * pr.K), B((size_t)pr.N * pr.K);
        for (auto &v : A) v = bf2f(f2bf((rand() / (float)RAND_MAX) * 2.f - 1.f));
        for (auto &v : B) v = bf2f(f2bf((rand() / (float)RAND_MAX) * 2.f - 1.f));
        std::vector<char> ia, ib;
        make_image(A, pr.M, pr.K, ia);
        make_image(B, pr.N, pr.K, ib);
        char *da, *db;
        CK(hipMalloc(&da, ia.size())); CK(hipMalloc(&db, ib.size()));
        CK(hipMemcpy(da, ia.data(), ia.size(), hipMemcpyHostToDevice));
        CK(hipMemcpy(db, ib.data(), ib.size(), hipMemcpyHostToDevice));
        CK(hipMalloc(&dA[q], A.size() * 4)); CK(hipMalloc(&dB[q], B.size() * 4));
        CK(hipMemcpy(dA[q], A.data(), A.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dB[q], B.data(), B.size() * 4, hipMemcpyHostToDevice));
        const size_t csz = (size_t)pr.M * pr.N * pr.kslices;
        CK(hipMalloc(&dC[q], csz * 4)); CK(hipMalloc(&dRef[q], (size_t)pr.M * pr.N * 4));
        CK(hipMemset(dC[q], 0xff, csz * 4));
        halo_g256::Prob &g = a.p[q];
        g.A = da; g.B = db; g.M = pr.M; g.N = pr.N; g.KT = (pr.K + 31) / 32;
        g.C = dC[q]; g.ldc = pr.N; g.n_split = pr.N; g.C2 = nullptr; g.ldc2 = 0;
        g.kslices = pr.kslices; g.slab_stride = (long)pr.M * pr.N; g.sumsq = nullptr;
        halo_g256::finish(g, first);
        first += g.tiles_m * g.tiles_n * g.kslices;
        flops += 2.0 * pr.M * pr.N * pr.K;
        hipLaunchKernelGGL(ref_kernel, dim3((pr.N + 15) / 16, (pr.M + 15) / 16), dim3(16, 16), 0, 0, dA[q], dB[q], dRef[q], pr.M, pr.N, pr.K);
    }
    printf("%d workgroups, %.2f GFLOP\n", first, flops * 1e-9);
    CK(halo_g256::launch<0>(a, first, 0));
    CK(hipDeviceSynchronize());
    for (size_t q = 0; q < probs.size(); ++q) {
        const Prob &pr = probs[q];
        std::vector<float> c((size_t)pr.M * pr.N * pr.kslices), ref((size_t)pr.M * pr.N);
        CK(hipMemcpy(c.data(), dC[q], c.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(ref.data(), dRef[q], ref.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0, scale = 0;
        for (size_t i = 0; i < ref.size(); ++i) {
            double s = 0;
            for (int k = 0; k < pr.kslices; ++k) s += c[(size_t)k * ref.size() + i];
            worst = fmax(worst, fabs(s - ref[i]));
            scale = fmax(scale, fabs((double)ref[i]));
        }
        printf("problem %zu [%d x %d x %d, %d K-slices]: max |diff| %.3e of max |ref| %.3e %s\n", q, pr.M, pr.N, pr.K, pr.kslices, worst, scale,
               worst <= 2e-5 * scale * sqrt((double)pr.K) ? "ok" : "MISMATCH");
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, auto fn) {
        float best = 1e9f, sum = 0;
        for (int rep = 0; rep < 4; ++rep) {
            const int iters = 30;
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < iters; ++i) CK(fn());
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = fminf(best, ms / iters); sum += ms / iters;
        }
        printf("%-44s best %.2f us (%.0f TFLOP/s), mean %.2f us\n", name, best * 1e3, flops / (best * 1e-3) * 1e-12, sum / 4 * 1e3);
    };
    {   // the four-wave kernel: verify, then time
        for (size_t q = 0; q < probs.size(); ++q) CK(hipMemset(dC[q], 0xff, (size_t)probs[q].M * probs[q].N * probs[q].kslices * 4));
        CK(halo_g256::launch_w4<0>(a, first, 0));
        CK(hipDeviceSynchronize());
        for (size_t q = 0; q < probs.size(); ++q) {
            const Prob &pr = probs[q];
            std::vector<float> c((size_t)pr.M * pr.N * pr.kslices), ref((size_t)pr.M * pr.N);
            CK(hipMemcpy(c.data(), dC[q], c.size() * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(ref.data(), dRef[q], ref.size() * 4, hipMemcpyDeviceToHost));
            double worst = 0, scale = 0;
            for (size_t i = 0; i < ref.size(); ++i) {
                double s = 0;
                for (int k = 0; k < pr.kslices; ++k) s += c[(size_t)k * ref.size() + i];
                worst = fmax(worst, fabs(s - ref[i]));
                scale = fmax(scale, fabs((double)ref[i]));
            }
            printf("4 waves: problem %zu max |diff| %.3e of %.3e %s\n", q, worst, scale, worst <= 2e-5 * scale * sqrt((double)pr.K) ? "ok" : "MISMATCH");
        }
    }
    timeit("4 waves: the product", [&] { return halo_g256::launch_w4<0>(a, first, 0); });
    timeit("4 waves: no epilogue stores", [&] { return halo_g256::launch_w4<8>(a, first, 0); });
    timeit("4 waves: no prefetch", [&] { return halo_g256::launch_w4<1 | 8>(a, first, 0); });
    timeit("4 waves: no fragment reads", [&] { return halo_g256::launch_w4<4 | 8>(a, first, 0); });
    timeit("4 waves: neither", [&] { return halo_g256::launch_w4<1 | 4 | 8>(a, first, 0); });
    timeit("the product", [&] { return halo_g256::launch<0>(a, first, 0); });
    timeit("no epilogue stores", [&] { return halo_g256::launch<8>(a, first, 0); });
    timeit("no prefetch issues (and no wait)", [&] { return halo_g256::launch<1 | 8>(a, first, 0); });
    timeit("no counted wait", [&] { return halo_g256::launch<2 | 8>(a, first, 0); });
    timeit("no fragment reads", [&] { return halo_g256::launch<4 | 8>(a, first, 0); });
    timeit("no fragment reads, no prefetch", [&] { return halo_g256::launch<1 | 4 | 8>(a, first, 0); });
    timeit("the product (again)", [&] { return halo_g256::launch<0>(a, first, 0); });
    return 0;
}
