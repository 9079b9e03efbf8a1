// LayerNorm forward, fp32.  HBM-bound: one wave64 per row, 16-byte loads, shuffle reductions,
// two-pass (mean, then centred variance) in registers so the numerics track nn.LayerNorm.
#include <algorithm>

#include "cfm_common.h"

// TOUT = float, or a 16-bit matrix-pipe type: under autocast the LayerNorm output only feeds GEMM A operands (and their
// weight-gradient GEMMs), which round it to that type anyway -- writing it rounded halves the bytes with identical results.
// A wave walks `rows_per_wave` rows (stride 4 inside the workgroup): gamma / beta are fetched once per wave instead of once per
// row (they are as many bytes as the row itself) and the next row is requested before the current one is reduced.
template <int VPL, typename TOUT>  // float4 vectors per lane; row length d <= VPL*256
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
    TOUT* __restrict__ y, float* __restrict__ mean_out, float* __restrict__ rstd_out,
    int64_t rows, int d, float eps, int rows_per_wave, float* __restrict__ stats_out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_wave * 4 + wave;
    const int64_t rend = min(rows, (int64_t)(blockIdx.x + 1) * rows_per_wave * 4);
    if (r0 >= rend) return;
    const int nvec = d >> 2;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    constexpr bool KEEP_GB = VPL <= 8;                 // wider rows re-read gamma / beta per row (register budget)
    f32x4 gam[KEEP_GB ? VPL : 1], bet[KEEP_GB ? VPL : 1];
    const f32x4* g4 = reinterpret_cast<const f32x4*>(gamma);
    const f32x4* b4 = reinterpret_cast<const f32x4*>(beta);
    if constexpr (KEEP_GB) {
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = min(lane + i * 64, nvec - 1);
            gam[i] = g4[c]; bet[i] = b4[c];
        }
    }
    f32x4 nxt[VPL];
    auto fetch = [&](int64_t row) {
        const f32x4* xr = reinterpret_cast<const f32x4*>(x + min(row, rows - 1) * d);
#pragma unroll
        for (int i = 0; i < VPL; ++i) nxt[i] = xr[min(lane + i * 64, nvec - 1)];
    };
    fetch(r0);
    for (int64_t row = r0; row < rend; row += 4) {
        f32x4 v[VPL];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            v[i] = lane + i * 64 < nvec ? nxt[i] : z4;
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
        if (row + 4 < rend) fetch(row + 4);
        const float mean = wave_sum(s) / (float)d;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            if (lane + i * 64 < nvec) {
                v[i] = v[i] - mean;
                q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
            }
        }
        const float var = wave_sum(q) / (float)d;
        const float rstd = 1.0f / sqrtf(var + eps);
        float so = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = lane + i * 64;
            if (c < nvec) {
                const f32x4 o = KEEP_GB ? v[i] * rstd * gam[i] + bet[i] : v[i] * rstd * g4[c] + b4[c];
                if constexpr (sizeof(TOUT) == 4) reinterpret_cast<f32x4*>(y + row * d)[c] = o;
                else *reinterpret_cast<typename Lowp<TOUT>::x4*>(y + row * d + 4 * c) = Lowp<TOUT>::cvt4(o);
                v[i] = o;
                so += (o.x + o.y) + (o.z + o.w);
            }
        }
        if (stats_out) {     // (kernel-uniform) LN fold: statistics of the OUTPUT row for the next LayerNorm: ONE partial (sum, M2)
            const float sum_o = wave_sum(so), mo = sum_o / (float)d;
            float qo = 0.f;
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                if (lane + i * 64 < nvec) {
                    const f32x4 t = v[i] - mo;
                    qo += (t.x * t.x + t.y * t.y) + (t.z * t.z + t.w * t.w);
                }
            }
            qo = wave_sum(qo);
            if (lane == 0) *reinterpret_cast<float2*>(stats_out + 2 * row) = float2{sum_o, qo};
        }
        if (lane == 0) {
            if (mean_out) mean_out[row] = mean;
            if (rstd_out) rstd_out[row] = rstd;
        }
    }
}

template <typename TOUT>
static int layernorm_launch(const float* x, const float* gamma, const float* beta, TOUT* y, float* mean_or_null,
                            float* rstd_or_null, int64_t rows, int d, float eps, hipStream_t s, float* stats_or_null = nullptr) {
    // rows per wave: 1 up to 16 k rows -- measured at 7968 x 512: 6.8 us with one row per wave, 7.7 us with three (fewer waves in
    // flight costs more than the shared gamma / beta fetch saves); more only for very tall inputs
    const int rpw = (int)std::max<int64_t>(1, std::min<int64_t>(4, rows / 16384));
    const dim3 grid((unsigned)((rows + 4 * rpw - 1) / (4 * rpw))), block(256);
#define LN_LAUNCH(V) hipLaunchKernelGGL((layernorm_fwd_kernel<V, TOUT>), grid, block, 0, s, x, gamma, beta, y, \
                                        mean_or_null, rstd_or_null, rows, d, eps, rpw, stats_or_null)
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

// LN fold (inference): the closing LayerNorm of a block (block.py:27) also emits the statistics partial of its OUTPUT row,
// [rows][1][2] = (sum, M2 about the mean), which the next block's first GEMM consumes (cfm_gemm_lnfold_f32, ln_parts = 1)
// in place of that block's ffn.py:16 LayerNorm.
extern "C" int cfm_layernorm_fwd_stats_f32(const float* x, const float* gamma, const float* beta, float* y,
                                           float* stats_out, int64_t rows, int d, float eps, cfm_stream_t stream) {
    CFM_REQUIRE(x && gamma && beta && y && stats_out, CFM_ERR_NULL);
    CFM_REQUIRE(rows > 0 && d > 0 && (d & 3) == 0, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(d <= 8192, CFM_ERR_UNSUPPORTED);
    CFM_REQUIRE(CFM_ALIGNED16(x) && CFM_ALIGNED16(y) && CFM_ALIGNED16(gamma) && CFM_ALIGNED16(beta) &&
                (reinterpret_cast<uintptr_t>(stats_out) & 7u) == 0, CFM_ERR_ALIGN);
    return layernorm_launch<float>(x, gamma, beta, y, nullptr, nullptr, rows, d, eps, static_cast<hipStream_t>(stream), stats_out);
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
