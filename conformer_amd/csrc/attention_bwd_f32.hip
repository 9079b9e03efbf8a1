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

// ... with y stored in a 16-bit matrix-pipe type: a masked gradient whose only consumers are GEMM operands (they would round it
// to that type anyway).  p = 0: a plain cast.
template <typename T16>
__global__ __launch_bounds__(256) void dropout_out16_kernel(const float* __restrict__ x, T16* __restrict__ y, int64_t n8, float p,
                                                            unsigned long long seed) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    const f32x4 v0 = reinterpret_cast<const f32x4*>(x)[2 * i], v1 = reinterpret_cast<const f32x4*>(x)[2 * i + 1];
    float k0[4] = {1.f, 1.f, 1.f, 1.f}, k1[4] = {1.f, 1.f, 1.f, 1.f};
    if (p > 0.f) {
        const float inv_keep = 1.0f / (1.0f - p);
        dropout_keep4(seed, (unsigned long long)(8 * i), p, inv_keep, k0);
        dropout_keep4(seed, (unsigned long long)(8 * i + 4), p, inv_keep, k1);
    }
    typename Lowp<T16>::x8 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) { r[e] = (T16)(v0[e] * k0[e]); r[4 + e] = (T16)(v1[e] * k1[e]); }
    reinterpret_cast<typename Lowp<T16>::x8*>(y)[i] = r;
}

}  // namespace

extern "C" int cfm_dropout_out16_f32(int prec, const float* x, void* y16, int64_t n, float p, uint64_t seed, cfm_stream_t stream) {
    CFM_REQUIRE(x && y16, CFM_ERR_NULL);
    CFM_REQUIRE(n > 0 && (n & 7) == 0 && p >= 0.f && p < 1.f, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(CFM_ALIGNED16(x) && CFM_ALIGNED16(y16), CFM_ERR_ALIGN);
    const dim3 grid((unsigned)((n / 8 + 255) / 256));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (prec == CFM_PREC_BF16) hipLaunchKernelGGL(dropout_out16_kernel<__bf16>, grid, dim3(256), 0, s, x, static_cast<__bf16*>(y16), n / 8, p, seed);
    else if (prec == CFM_PREC_FP16) hipLaunchKernelGGL(dropout_out16_kernel<_Float16>, grid, dim3(256), 0, s, x, static_cast<_Float16*>(y16), n / 8, p, seed);
    else return CFM_ERR_UNSUPPORTED;
    return cfm_launch_status();
}

extern "C" int cfm_dropout_f32(const float* x, float* y, int64_t n, float p, uint64_t seed, cfm_stream_t stream) {
    CFM_REQUIRE(x && y, CFM_ERR_NULL);
    CFM_REQUIRE(n > 0 && (n & 3) == 0 && p >= 0.f && p < 1.f, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(CFM_ALIGNED16(x) && CFM_ALIGNED16(y), CFM_ERR_ALIGN);
    hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       x, y, n / 4, p, seed);
    return cfm_launch_status();
}
