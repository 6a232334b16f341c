// Direct peer exchange for the data-parallel step ("prefer direct reduce-scatter + all-gather across the 7 links
// over ring"; the gradient semantics are DistributedDataParallel's, ha/attention_loop.py:67-82,154,203): every rank owns an ARENA of device
// memory that its peers map through HIP IPC, and a collective is three small launches on the rank's own stream --
//     push   : this rank's 1/N pieces written STRAIGHT into the owners' arenas, all peers at once (xGMI is point-to-point: seven links carry
//              seven pieces in parallel; no ring, no intermediate hops), then a system-scope fence;
//     signal : one epoch word per peer, stored into the PEER's flag block behind a second system-scope fence (its own launch: the push's
//              stores have all been acknowledged when it starts);
//     wait   : one workgroup polls this rank's flag block (system-scope loads, s_sleep between polls) until every peer's word carries the
//              epoch -- a BOUNDED wait: on a timeout it raises the caller's sticky status word (halo_set_status_word) and returns, so a
//              missing peer fails the step loudly (the clip launch applies no update, LstmCtcTrainer.check_status() raises) instead of
//              hanging the GPU --
// followed by the consumer's own kernel (the reduction of the received pieces in RANK ORDER, the same sum on whichever rank forms it; or
// the expansion of the gathered bf16 records).  Epochs only grow (the step count); a slot is rewritten by the next step's push only
// after its owner has signalled a LATER collective of the step before (the all-gather closes every step), so no back-signal is needed.
// The arena is allocated uncached (hipDeviceMallocUncached: what RCCL uses for the buffers its peers write), falling back to
// fine-grained memory where the runtime refuses the flag -- never to plain (L2-cached, non-coherent) device memory.
#include <string.h>
#include "halo_common.h"
#include "halo_internal.h"

namespace {

constexpr int DX_MAX_WORLD = 16;
constexpr unsigned long long DX_TIMEOUT_TICKS = 500000000ull;      // 5 s of s_memrealtime (100 MHz)

struct DxPeers {
    char *base[DX_MAX_WORLD];      // every rank's arena as mapped into THIS process (own rank: the local pointer)
};

// piece p of src (elems elements of esize bytes each, piece stride piece_stride elements) -> peer p's arena at dst_off + rank * elems * esize,
// for every peer p != rank (same = 1: the SAME piece, src itself, to every peer: the all-gather's push).  16-byte units; grid.y = peer slot.
__global__ __launch_bounds__(256) void dx_push_kernel(const char *__restrict__ src, long piece_bytes, long piece_stride_bytes, int same, DxPeers peers,
                                                      long dst_off, int world, int rank) {
    int p = blockIdx.y;
    if (p >= rank) ++p;                                     // the world - 1 peers, skipping this rank
    if (p >= world) return;
    const uint4 *s = reinterpret_cast<const uint4 *>(src + (same ? 0 : (long)p * piece_stride_bytes));
    uint4 *d = reinterpret_cast<uint4 *>(peers.base[p] + dst_off + (long)rank * piece_bytes);
    const long n16 = piece_bytes / 16;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n16; i += (long)gridDim.x * 256) d[i] = s[i];
    __threadfence_system();
}

// flag word [kind][rank] of every peer's flag block <- epoch
__global__ __launch_bounds__(64) void dx_signal_kernel(DxPeers peers, long flag_off, int world, int rank, unsigned epoch) {
    __threadfence_system();
    const int p = threadIdx.x;
    if (p < world && p != rank)
        __hip_atomic_store(reinterpret_cast<unsigned *>(peers.base[p] + flag_off) + rank, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// every peer's word in this rank's own flag block carries `epoch` (or a later one); bounded
__global__ __launch_bounds__(64) void dx_wait_kernel(const unsigned *flags, int world, int rank, unsigned epoch, unsigned *status) {
    const int p = threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    bool ok = p >= world || p == rank;
    bool timed_out = false;
    while (!__all(ok)) {
        if (!ok) ok = (int)(__hip_atomic_load(flags + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - epoch) >= 0;
        if (__builtin_amdgcn_s_memrealtime() - t0 > DX_TIMEOUT_TICKS) { timed_out = true; break; }
        __builtin_amdgcn_s_sleep(8);
    }
    __threadfence_system();
    if (timed_out && threadIdx.x == 0 && status) __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// own [elems] <- (sum over the ranks in RANK ORDER of: own (at position `rank`), inbox piece p (p != rank)) * scale; four elements per thread
__global__ __launch_bounds__(256) void dx_reduce_kernel(float *__restrict__ own, const float *__restrict__ inbox, long elems, int world, int rank, float scale) {
    const long n4 = elems / 4;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int p = 0; p < world; ++p) {
            const f32x4 v = p == rank ? reinterpret_cast<const f32x4 *>(own)[i]
                                      : __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(inbox + (long)p * elems) + i);
            s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
        }
        reinterpret_cast<f32x4 *>(own)[i] = f32x4{s[0] * scale, s[1] * scale, s[2] * scale, s[3] * scale};
    }
}

}  // namespace

extern "C" {

int halo_dx_alloc(size_t bytes, void **ptr, void *handle64) {
    HALO_CHECK_ARG(bytes > 0 && ptr && handle64);
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "the IPC handle travels as 64 bytes");
    void *p = nullptr;
    // (never plain device memory: a peer's stores over xGMI would not be coherent with this GPU's L2 -- a single-GPU test would pass
    //  and eight GPUs would sum stale pieces; a runtime that refuses both coherent kinds fails the construction, loudly)
    if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached) != hipSuccess) {
        (void)hipGetLastError();
        if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained) != hipSuccess) {
            (void)hipGetLastError();
            return HALO_ELAUNCH;
        }
    }
    if (hipMemset(p, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { (void)hipFree(p); return HALO_ELAUNCH; }
    if (hipIpcGetMemHandle(reinterpret_cast<hipIpcMemHandle_t *>(handle64), p) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(p); return HALO_ELAUNCH; }
    *ptr = p;
    return HALO_OK;
}

int halo_dx_open(const void *handle64, void **ptr) {
    HALO_CHECK_ARG(handle64 && ptr);
    hipIpcMemHandle_t h;
    memcpy(&h, handle64, sizeof(h));
    if (hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); return HALO_ELAUNCH; }
    return HALO_OK;
}

int halo_dx_close(void *ptr) { return ptr && hipIpcCloseMemHandle(ptr) == hipSuccess ? HALO_OK : HALO_ELAUNCH; }
int halo_dx_free(void *ptr) { return ptr && hipFree(ptr) == hipSuccess ? HALO_OK : HALO_ELAUNCH; }

static int dx_peers(void *const *bases, int world, DxPeers &q) {
    if (!bases || world < 1 || world > DX_MAX_WORLD) return HALO_EINVAL;
    for (int i = 0; i < world; ++i) {
        if (!bases[i] || (uintptr_t)bases[i] % 16) return HALO_EINVAL;
        q.base[i] = (char *)bases[i];
    }
    return HALO_OK;
}

int halo_dx_push(const void *src, size_t piece_bytes, size_t piece_stride_bytes, int same, void *const *peer_bases, size_t dst_offset, int world,
                 int rank, halo_stream_t stream) {
    HALO_CHECK_ARG(src && piece_bytes > 0 && piece_bytes % 16 == 0 && piece_stride_bytes % 16 == 0 && dst_offset % 16 == 0 && (uintptr_t)src % 16 == 0);
    HALO_CHECK_ARG(rank >= 0 && rank < world);
    DxPeers q = {};
    const int rc = dx_peers(peer_bases, world, q);
    if (rc != HALO_OK) return rc;
    if (world == 1) return HALO_OK;
    const long n16 = (long)(piece_bytes / 16);
    const unsigned bx = (unsigned)((n16 + 255) / 256 > 256 ? 256 : (n16 + 255) / 256);
    hipLaunchKernelGGL(dx_push_kernel, dim3(bx, (unsigned)(world - 1)), dim3(256), 0, (hipStream_t)stream, (const char *)src, (long)piece_bytes,
                       (long)piece_stride_bytes, same, q, (long)dst_offset, world, rank);
    return halo_launch_status();
}

int halo_dx_signal(void *const *peer_bases, size_t flag_offset, int world, int rank, uint32_t epoch, halo_stream_t stream) {
    HALO_CHECK_ARG(flag_offset % 16 == 0 && rank >= 0 && rank < world);
    DxPeers q = {};
    const int rc = dx_peers(peer_bases, world, q);
    if (rc != HALO_OK) return rc;
    hipLaunchKernelGGL(dx_signal_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, q, (long)flag_offset, world, rank, epoch);
    return halo_launch_status();
}

int halo_dx_wait(const void *own_flags, int world, int rank, uint32_t epoch, halo_stream_t stream) {
    HALO_CHECK_ARG(own_flags && world >= 1 && world <= DX_MAX_WORLD && rank >= 0 && rank < world);
    hipLaunchKernelGGL(dx_wait_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const unsigned *)own_flags, world, rank, epoch, halo_ctx_cur().status);
    return halo_launch_status();
}

int halo_dx_reduce(float *own, const float *inbox, size_t elems, int world, int rank, float scale, halo_stream_t stream) {
    HALO_CHECK_ARG(own && inbox && elems > 0 && elems % 4 == 0 && ((uintptr_t)own | (uintptr_t)inbox) % 16 == 0 && rank >= 0 && rank < world);
    const long n4 = (long)(elems / 4);
    const unsigned bx = (unsigned)((n4 + 255) / 256 > 2048 ? 2048 : (n4 + 255) / 256);
    hipLaunchKernelGGL(dx_reduce_kernel, dim3(bx), dim3(256), 0, (hipStream_t)stream, own, inbox, (long)elems, world, rank, scale);
    return halo_launch_status();
}

}  // extern "C"
