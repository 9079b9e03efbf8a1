// Stand-alone dropout mask (the backward applies it to incoming gradients).  The round-1 glue kernels of the materialising
// attention backward (q-bias, row-dot, softmax-backward over (B,H,T,T) tensors, strided add) lived here; the fused kernel
// in attention_bwd_flash_f32.hip replaced them.
#include "cfm_common.h"

namespace {

// y = x * keep(seed, flat index): the stand-alone form of the epilogue dropout (backward: mask the incoming gradient)
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n4,
                                                      float p, unsigned long long seed) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const float inv_keep = 1.0f / (1.0f - p);
    f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
    float keep[4];
    dropout_keep4(seed, (unsigned long long)(4 * i), p, inv_keep, keep);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] *= keep[e];
    reinterpret_cast<f32x4*>(y)[i] = v;
}

}  // namespace

extern "C" int cfm_dropout_f32(const float* x, float* y, int64_t n, float p, uint64_t seed, cfm_stream_t stream) {
    CFM_REQUIRE(x && y, CFM_ERR_NULL);
    CFM_REQUIRE(n > 0 && (n & 3) == 0 && p >= 0.f && p < 1.f, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(CFM_ALIGNED16(x) && CFM_ALIGNED16(y), CFM_ERR_ALIGN);
    hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       x, y, n / 4, p, seed);
    return cfm_launch_status();
}
