// LayerNorm forward, fp32.  HBM-bound: one wave64 per row, 16-byte loads, shuffle reductions,
// two-pass (mean, then centred variance) in registers so the numerics track nn.LayerNorm.
#include "cfm_common.h"

// TOUT = float, or a 16-bit matrix-pipe type: under autocast the LayerNorm output only feeds GEMM A operands (and their
// weight-gradient GEMMs), which round it to that type anyway -- writing it rounded halves the bytes with identical results.
template <int VPL, typename TOUT>  // float4 vectors per lane; row length d <= VPL*256
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
    TOUT* __restrict__ y, float* __restrict__ mean_out, float* __restrict__ rstd_out,
    int64_t rows, int d, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nvec = d >> 2;
    const f32x4* xr = reinterpret_cast<const f32x4*>(x + row * d);
    f32x4 v[VPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = lane + i * 64;
        if (c < nvec) { v[i] = xr[c]; s += (v[i].x + v[i].y) + (v[i].z + v[i].w); }
        else v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = lane + i * 64;
        if (c < nvec) {
            v[i] = v[i] - mean;
            q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
        }
    }
    const float var = wave_sum(q) / (float)d;
    const float rstd = 1.0f / sqrtf(var + eps);
    const f32x4* g4 = reinterpret_cast<const f32x4*>(gamma);
    const f32x4* b4 = reinterpret_cast<const f32x4*>(beta);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = lane + i * 64;
        if (c < nvec) {
            const f32x4 o = v[i] * rstd * g4[c] + b4[c];
            if constexpr (sizeof(TOUT) == 4) reinterpret_cast<f32x4*>(y + row * d)[c] = o;
            else *reinterpret_cast<typename Lowp<TOUT>::x4*>(y + row * d + 4 * c) = Lowp<TOUT>::cvt4(o);
        }
    }
    if (lane == 0) {
        if (mean_out) mean_out[row] = mean;
        if (rstd_out) rstd_out[row] = rstd;
    }
}

template <typename TOUT>
static int layernorm_launch(const float* x, const float* gamma, const float* beta, TOUT* y, float* mean_or_null,
                            float* rstd_or_null, int64_t rows, int d, float eps, hipStream_t s) {
    const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
#define LN_LAUNCH(V) hipLaunchKernelGGL((layernorm_fwd_kernel<V, TOUT>), grid, block, 0, s, x, gamma, beta, y, \
                                        mean_or_null, rstd_or_null, rows, d, eps)
    if (d <= 256) LN_LAUNCH(1);
    else if (d <= 512) LN_LAUNCH(2);
    else if (d <= 1024) LN_LAUNCH(4);
    else if (d <= 2048) LN_LAUNCH(8);
    else LN_LAUNCH(32);
#undef LN_LAUNCH
    return cfm_launch_status();
}

extern "C" int cfm_layernorm_fwd_f32(const float* x, const float* gamma, const float* beta, float* y,
                                     float* mean_or_null, float* rstd_or_null,
                                     int64_t rows, int d, float eps, cfm_stream_t stream) {
    CFM_REQUIRE(x && gamma && beta && y, CFM_ERR_NULL);
    CFM_REQUIRE(rows > 0 && d > 0 && (d & 3) == 0, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(d <= 8192, CFM_ERR_UNSUPPORTED);
    CFM_REQUIRE(CFM_ALIGNED16(x) && CFM_ALIGNED16(y) && CFM_ALIGNED16(gamma) && CFM_ALIGNED16(beta), CFM_ERR_ALIGN);
    return layernorm_launch<float>(x, gamma, beta, y, mean_or_null, rstd_or_null, rows, d, eps, static_cast<hipStream_t>(stream));
}

// LayerNorm whose output is written in the 16-bit matrix-pipe type `prec` (CFM_PREC_BF16 | CFM_PREC_FP16), for consumers
// that are 16-bit GEMM operands (a_is_16bit / b_is_16bit); statistics and arithmetic are fp32, one RNE rounding at the store.
extern "C" int cfm_layernorm_fwd_out16_f32(int prec, const float* x, const float* gamma, const float* beta, void* y16,
                                           float* mean_or_null, float* rstd_or_null, int64_t rows, int d, float eps,
                                           cfm_stream_t stream) {
    CFM_REQUIRE(x && gamma && beta && y16, CFM_ERR_NULL);
    CFM_REQUIRE(rows > 0 && d > 0 && (d & 3) == 0, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(d <= 8192, CFM_ERR_UNSUPPORTED);
    CFM_REQUIRE(CFM_ALIGNED16(x) && CFM_ALIGNED16(gamma) && CFM_ALIGNED16(beta) && (reinterpret_cast<uintptr_t>(y16) & 7) == 0,
                CFM_ERR_ALIGN);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (prec == CFM_PREC_BF16)
        return layernorm_launch<__bf16>(x, gamma, beta, static_cast<__bf16*>(y16), mean_or_null, rstd_or_null, rows, d, eps, s);
    if (prec == CFM_PREC_FP16)
        return layernorm_launch<_Float16>(x, gamma, beta, static_cast<_Float16*>(y16), mean_or_null, rstd_or_null, rows, d, eps, s);
    return CFM_ERR_UNSUPPORTED;
}
