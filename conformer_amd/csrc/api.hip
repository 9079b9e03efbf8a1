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

// The same on a device array (convolution.py:55 applied to `lengths`): floor division, as torch.div(rounding_mode="floor")
// -- an arithmetic shift also for negative values.  One launch instead of four stock elementwise kernels per forward.
__global__ void subsampled_lengths_kernel(const int64_t* __restrict__ in, int64_t* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (((in[i] - 1) >> 1) - 1) >> 1;
}
extern "C" int cfm_subsampled_lengths_i64(const int64_t* lengths, int64_t* out, int n, cfm_stream_t stream) {
    CFM_REQUIRE(lengths && out, CFM_ERR_NULL);
    CFM_REQUIRE(n > 0, CFM_ERR_BAD_SHAPE);
    hipLaunchKernelGGL(subsampled_lengths_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), lengths, out, n);
    return cfm_launch_status();
}
