// Library-level entry points: version, error strings, device check.
#include "cfm_common.h"
#include <string.h>

extern "C" int cfm_version(void) { return 1; }
extern "C" int cfm_abi_version(void) { return CFM_ABI_VERSION; }

extern "C" const char* cfm_strerror(int s) {
    switch (s) {
        case CFM_OK: return "ok";
        case CFM_ERR_BAD_SHAPE: return "bad shape";
        case CFM_ERR_UNSUPPORTED: return "unsupported configuration";
        case CFM_ERR_NULL: return "null pointer";
        case CFM_ERR_LAUNCH: return "kernel launch failed";
        case CFM_ERR_DEVICE: return "device is not gfx950";
        case CFM_ERR_ALIGN: return "pointer or leading dimension not 16-byte aligned";
        default: return "unknown status";
    }
}

extern "C" int cfm_device_check(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return CFM_ERR_DEVICE;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) return CFM_ERR_DEVICE;
    return strncmp(p.gcnArchName, "gfx950", 6) == 0 ? CFM_OK : CFM_ERR_DEVICE;
}

extern "C" int64_t cfm_subsampled_length(int64_t n) { return ((n - 1) / 2 - 1) / 2; }
