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
