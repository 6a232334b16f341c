// ABI bookkeeping: version, error strings, device probe.
#include <string.h>
#include "halo_common.h"
#include "halo_internal.h"

static int g_math_mode = 0;
static void *g_scratch = nullptr;
static size_t g_scratch_bytes = 0;
void halo_get_scratch(void **ptr, size_t *bytes) { *ptr = g_scratch; *bytes = g_scratch_bytes; }
int halo_math_mode() { return g_math_mode; }

extern "C" {

int halo_set_math_mode(int mode) {
    if (mode != HALO_MATH_F32 && mode != HALO_MATH_BF16X3) return HALO_EINVAL;
    g_math_mode = mode;
    return HALO_OK;
}
int halo_get_math_mode(void) { return g_math_mode; }

int halo_set_scratch(void *device_ptr, size_t bytes) {
    if (device_ptr && ((uintptr_t)device_ptr % 16 != 0)) return HALO_EINVAL;
    g_scratch = device_ptr;
    g_scratch_bytes = device_ptr ? bytes : 0;
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
