// Depthwise Conv1d (k=K, "same" zero padding per utterance) + BatchNorm1d(eval) + Swish, channel-last.
// HBM-bound.  Lanes run along channels (coalesced rows of the (B,T,C) tensor); each wave owns TT
// consecutive frames of 64 channels and slides over the TT+K-1 input frames with the K taps and the TT
// accumulators held in registers (both loops fully unrolled => static register indexing).
#include "cfm_common.h"

namespace {

// TOUT = float, or a 16-bit matrix-pipe type: under autocast the module's output only feeds the pointwise_conv_2 GEMM (and, in
// training, its weight-gradient GEMM), which round it to that type anyway.
template <int K, int TT, typename TOUT = float>
__global__ __launch_bounds__(256) void dwconv_bn_swish_kernel(
    const float* __restrict__ g, const float* __restrict__ w, const float* __restrict__ bias,
    const float* __restrict__ bn_w, const float* __restrict__ bn_b, const float* __restrict__ bn_mean,
    const float* __restrict__ bn_var, float eps, TOUT* __restrict__ y, int T, int C) {
    constexpr int HALF = (K - 1) / 2;
    __shared__ float taps[64 * K];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int t0 = (blockIdx.y * 4 + wave) * TT;
    const int b = blockIdx.z;
    float wr[K];
    load_taps<K>(w, blockIdx.x * 64, C, taps, wr);
    if (t0 >= T) return;                               // wave-uniform (after the block-wide tap staging)
    const bool cok = c < C;
    const int cc = cok ? c : C - 1;
    float acc[TT];
    const float bi = bias[cc];
#pragma unroll
    for (int o = 0; o < TT; ++o) acc[o] = bi;
    const float* gb = g + (int64_t)b * T * C + cc;
    float win[TT + K - 1];
    load_window<TT + K - 1>(gb, t0 - HALF, T, C, win);
#pragma unroll
    for (int o = 0; o < TT; ++o)
#pragma unroll
        for (int j = 0; j < K; ++j) acc[o] = fmaf(wr[j], win[o + j], acc[o]);
    const float inv = 1.0f / sqrtf(bn_var[cc] + eps);
    const float mu = bn_mean[cc], ga = bn_w[cc], be = bn_b[cc];
    TOUT* yb = y + (int64_t)b * T * C + cc;
#pragma unroll
    for (int o = 0; o < TT; ++o) {
        const int t = t0 + o;
        if (t < T && cok) yb[(int64_t)t * C] = (TOUT)swishf_acc((acc[o] - mu) * inv * ga + be);
    }
}

// any odd K <= 63: one thread per output element, taps streamed from L1/L2
__global__ __launch_bounds__(256) void dwconv_bn_swish_generic_kernel(
    const float* __restrict__ g, const float* __restrict__ w, const float* __restrict__ bias,
    const float* __restrict__ bn_w, const float* __restrict__ bn_b, const float* __restrict__ bn_mean,
    const float* __restrict__ bn_var, float eps, float* __restrict__ y, int B, int T, int C, int K) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)B * T * C) return;
    const int c = (int)(idx % C);
    const int64_t bt = idx / C;
    const int t = (int)(bt % T);
    const int half = (K - 1) / 2;
    float acc = bias[c];
    for (int j = 0; j < K; ++j) {
        const int tt = t + j - half;
        if (tt >= 0 && tt < T) acc = fmaf(w[(int64_t)c * K + j], g[idx + (int64_t)(j - half) * C], acc);
    }
    const float inv = 1.0f / sqrtf(bn_var[c] + eps);
    y[idx] = swishf_acc((acc - bn_mean[c]) * inv * bn_w[c] + bn_b[c]);
}

}  // namespace

extern "C" int cfm_dwconv_bn_swish_fwd_f32(const float* g, const float* w, const float* bias, const float* bn_weight,
                                           const float* bn_bias, const float* bn_mean, const float* bn_var,
                                           float bn_eps, float* y, int B, int T, int C, int K, cfm_stream_t stream) {
    CFM_REQUIRE(g && w && bias && bn_weight && bn_bias && bn_mean && bn_var && y, CFM_ERR_NULL);
    CFM_REQUIRE(B > 0 && T > 0 && C > 0 && K > 0 && (K & 1) == 1, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(K <= 63, CFM_ERR_UNSUPPORTED);
    hipStream_t s = static_cast<hipStream_t>(stream);
    constexpr int TT = 16;
    const dim3 grid((unsigned)((C + 63) / 64), (unsigned)((T + 4 * TT - 1) / (4 * TT)), (unsigned)B), block(256);
#define DW_LAUNCH(KK) hipLaunchKernelGGL((dwconv_bn_swish_kernel<KK, TT>), grid, block, 0, s, g, w, bias, bn_weight, \
                                         bn_bias, bn_mean, bn_var, bn_eps, y, T, C)
    switch (K) {
        case 31: DW_LAUNCH(31); break;
        case 15: DW_LAUNCH(15); break;
        case 7: DW_LAUNCH(7); break;
        case 3: DW_LAUNCH(3); break;
        default: {
            const int64_t total = (int64_t)B * T * C;
            hipLaunchKernelGGL(dwconv_bn_swish_generic_kernel, dim3((unsigned)((total + 255) / 256)), block, 0, s, g, w,
                               bias, bn_weight, bn_bias, bn_mean, bn_var, bn_eps, y, B, T, C, K);
        }
    }
#undef DW_LAUNCH
    return cfm_launch_status();
}

// cfm_dwconv_bn_swish_fwd_f32 with y stored in the 16-bit type `prec` (K in {3, 7, 15, 31}; CFM_ERR_UNSUPPORTED otherwise).
extern "C" int cfm_dwconv_bn_swish_fwd_out16_f32(int prec, const float* g, const float* w, const float* bias,
                                                 const float* bn_weight, const float* bn_bias, const float* bn_mean,
                                                 const float* bn_var, float bn_eps, void* y16, int B, int T, int C, int K,
                                                 cfm_stream_t stream) {
    CFM_REQUIRE(g && w && bias && bn_weight && bn_bias && bn_mean && bn_var && y16, CFM_ERR_NULL);
    CFM_REQUIRE(B > 0 && T > 0 && C > 0, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(K == 31 || K == 15 || K == 7 || K == 3, CFM_ERR_UNSUPPORTED);
    CFM_REQUIRE(prec == CFM_PREC_BF16 || prec == CFM_PREC_FP16, CFM_ERR_UNSUPPORTED);
    hipStream_t s = static_cast<hipStream_t>(stream);
    constexpr int TT = 16;
    const dim3 grid((unsigned)((C + 63) / 64), (unsigned)((T + 4 * TT - 1) / (4 * TT)), (unsigned)B), block(256);
#define DW16(KK, TY) hipLaunchKernelGGL((dwconv_bn_swish_kernel<KK, TT, TY>), grid, block, 0, s, g, w, bias, bn_weight, bn_bias, \
                                        bn_mean, bn_var, bn_eps, static_cast<TY*>(y16), T, C)
#define DW16K(TY) switch (K) { case 31: DW16(31, TY); break; case 15: DW16(15, TY); break; case 7: DW16(7, TY); break; default: DW16(3, TY); }
    if (prec == CFM_PREC_BF16) { DW16K(__bf16) } else { DW16K(_Float16) }
#undef DW16K
#undef DW16
    return cfm_launch_status();
}

