// What a barrier among G co-resident workgroups costs inside one launch on MI355X (the question behind a one-launch greedy decode:
// 62 stages per token, each an all-to-all over the 64 rows' features):
//   counter : one monotonic counter, arrive = agent-scope release fence + atomic add, wait = relaxed polls + acquire fence
//   words   : one epoch word per workgroup (write-through store), everybody polls all G words (no atomic, no serial arrivals)
// each with and without a payload (every workgroup writes 2 KB that every other one reads behind the barrier).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/grid_barrier_probe.hip -o tools/grid_barrier_probe.bin && tools/grid_barrier_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr unsigned long long TIMEOUT_TICKS = 20000000ull;       // 0.2 s of s_memrealtime (100 MHz)

__device__ __forceinline__ void sync_counter(unsigned *ctr, unsigned target, unsigned *status) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
            if (__builtin_amdgcn_s_memrealtime() - t0 > TIMEOUT_TICKS) { *status = 1; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

// words [G]: workgroup b stores epoch into words[b]; wave 0 polls all G words
__device__ __forceinline__ void sync_words(unsigned *words, int G, unsigned epoch, unsigned *status) {
    __syncthreads();
    if (threadIdx.x < 64) {
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_store(words + blockIdx.x, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        bool ok;
        do {
            ok = true;
            for (int i = threadIdx.x; i < G; i += 64)
                ok = ok && (int)(__hip_atomic_load(words + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - epoch) >= 0;
            ok = __all(ok);
            if (!ok) {
                if (__builtin_amdgcn_s_memrealtime() - t0 > TIMEOUT_TICKS) { if (threadIdx.x == 0) *status = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        } while (!ok);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

template <int MODE, bool PAYLOAD>
__global__ __launch_bounds__(256) void probe_kernel(unsigned *ctr, unsigned *words, float *buf, int rounds, unsigned *status, float *sink) {
    const int G = gridDim.x;
    float acc = 0.f;
    for (int r = 0; r < rounds; ++r) {
        if (PAYLOAD) {
            // this workgroup's 2 KB of round r (two planes: round r reads plane r & 1 while round r + 1 is written into the other)
            float *mine = buf + ((long)(r & 1) * G + blockIdx.x) * 512;
            mine[threadIdx.x] = (float)(r + blockIdx.x);
            mine[256 + threadIdx.x] = (float)(r - (int)blockIdx.x);
        }
        if (MODE == 0) sync_counter(ctr, (unsigned)(r + 1) * G, status);
        else sync_words(words, G, (unsigned)(r + 1), status);
        if (PAYLOAD) {
            const float *all = buf + (long)(r & 1) * G * 512;
            for (int i = threadIdx.x; i < G * 512; i += 256 * 8) acc += all[i];
        }
    }
    if (acc == 1234.5f) *sink = acc;
}

template <int MODE, bool PAYLOAD>
static void run(const char *name, int G, int rounds) {
    unsigned *ctr, *words, *status; float *buf, *sink;
    CK(hipMalloc(&ctr, 256)); CK(hipMalloc(&words, 4096)); CK(hipMalloc(&status, 4)); CK(hipMalloc(&buf, 2L * 256 * 2048)); CK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    unsigned hs = 0;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipMemset(ctr, 0, 256)); CK(hipMemset(words, 0, 4096)); CK(hipMemset(status, 0, 4));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((probe_kernel<MODE, PAYLOAD>), dim3(G), dim3(256), 0, 0, ctr, words, buf, rounds, status, sink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
        CK(hipMemcpy(&hs, status, 4, hipMemcpyDeviceToHost));
        if (hs) break;
    }
    printf("%-28s G %3d: %6.2f us per round%s\n", name, G, best * 1e3f / rounds, hs ? "  TIMED OUT" : "");
    CK(hipFree(ctr)); CK(hipFree(words)); CK(hipFree(status)); CK(hipFree(buf)); CK(hipFree(sink));
}

int main() {
    const int rounds = 500;
    for (int G : {32, 64, 128, 256}) {
        run<0, false>("counter", G, rounds);
        run<1, false>("words", G, rounds);
        run<0, true>("counter + 2 KB payload", G, rounds);
        run<1, true>("words + 2 KB payload", G, rounds);
    }
    return 0;
}
