// ABI bookkeeping: version, error strings, device probe.
#include <string.h>
#include <new>
#include "halo_common.h"
#include "halo_internal.h"

static HaloCtx g_default_ctx;
static thread_local HaloCtx *t_ctx = nullptr;
HaloCtx &halo_ctx_cur() { return t_ctx ? *t_ctx : g_default_ctx; }
int halo_lstm_fusion() { return halo_ctx_cur().lstm_fusion; }
// the scratch is cut in two halves so that work forked onto a side stream (slot 1) never shares
// split-K slabs with the main stream (slot 0)
void halo_set_scratch_slot(int slot) { halo_ctx_cur().scratch_slot = slot ? 1 : 0; }
void halo_get_scratch(void **ptr, size_t *bytes) {
    const HaloCtx &c = halo_ctx_cur();
    const size_t half = (c.scratch_bytes / 2) & ~(size_t)255;
    *ptr = c.scratch ? (char *)c.scratch + (size_t)c.scratch_slot * half : nullptr;
    *bytes = c.scratch ? half : 0;
}

// side stream + events for fork/join inside one C-ABI call (created once, on first use, outside capture)
static hipStream_t g_side_stream = nullptr;
static hipEvent_t g_fork_event = nullptr, g_join_event = nullptr;
int halo_side_stream(hipStream_t *side, hipEvent_t *fork_ev, hipEvent_t *join_ev) {
    if (!g_side_stream) {
        int lo = 0, hi = 0;
        hipDeviceGetStreamPriorityRange(&lo, &hi);      // lo = least urgent
        if (hipStreamCreateWithPriority(&g_side_stream, hipStreamNonBlocking, lo) != hipSuccess) return HALO_ELAUNCH;
        if (hipEventCreateWithFlags(&g_fork_event, hipEventDisableTiming) != hipSuccess) return HALO_ELAUNCH;
        if (hipEventCreateWithFlags(&g_join_event, hipEventDisableTiming) != hipSuccess) return HALO_ELAUNCH;
    }
    *side = g_side_stream; *fork_ev = g_fork_event; *join_ev = g_join_event;
    return HALO_OK;
}
int halo_math_mode() { return halo_ctx_cur().math_mode; }

extern "C" {

int halo_set_math_mode(int mode) {
    if (mode != HALO_MATH_F32 && mode != HALO_MATH_BF16X3 && mode != HALO_MATH_BF16) return HALO_EINVAL;
    halo_ctx_cur().math_mode = mode;
    return HALO_OK;
}
int halo_get_math_mode(void) { return halo_ctx_cur().math_mode; }

int halo_set_lstm_fusion(int on) {
    halo_ctx_cur().lstm_fusion = on ? 1 : 0;
    return HALO_OK;
}

int halo_set_scratch(void *device_ptr, size_t bytes) {
    if (device_ptr && ((uintptr_t)device_ptr % 16 != 0)) return HALO_EINVAL;
    halo_ctx_cur().scratch = device_ptr;
    halo_ctx_cur().scratch_bytes = device_ptr ? bytes : 0;
    return HALO_OK;
}

halo_ctx *halo_ctx_create(void) { return reinterpret_cast<halo_ctx *>(new (std::nothrow) HaloCtx(halo_ctx_cur())); }
void halo_ctx_destroy(halo_ctx *ctx) {
    HaloCtx *c = reinterpret_cast<HaloCtx *>(ctx);
    if (t_ctx == c) t_ctx = nullptr;
    delete c;
}
int halo_ctx_use(halo_ctx *ctx) {
    t_ctx = reinterpret_cast<HaloCtx *>(ctx);
    return HALO_OK;
}
int halo_set_status_word(uint32_t *device_word) {
    if (device_word && ((uintptr_t)device_word % 4 != 0)) return HALO_EINVAL;
    halo_ctx_cur().status = device_word;
    return HALO_OK;
}

int halo_abi_version(void) { return HALO_ABI_VERSION; }

const char *halo_strerror(int code) {
    switch (code) {
        case HALO_OK: return "ok";
        case HALO_EINVAL: return "invalid argument";
        case HALO_ENOTSUP: return "shape not supported by the gfx950 kernels";
        case HALO_ELAUNCH: return "HIP launch failed";
        default: return "unknown halo error";
    }
}

int halo_device_info(int device, char *arch, int arch_len, int *cu_count) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return HALO_ELAUNCH;
    if (arch && arch_len > 0) {
        strncpy(arch, prop.gcnArchName, (size_t)arch_len - 1);
        arch[arch_len - 1] = 0;
    }
    if (cu_count) *cu_count = prop.multiProcessorCount;
    return HALO_OK;
}

}  // extern "C"
