// ABI bookkeeping: version, error strings, device probe.
#include <string.h>
#include "halo_common.h"

extern "C" {

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
