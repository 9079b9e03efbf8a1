// Convolution-subsampling stem, first half: 3x3/stride-2 conv (C_in = 1) + ReLU writing a CHANNEL-LAST
// activation h1 (B,T1,F1,C), plus the two one-time weight re-layouts that let the second conv and the input
// Linear run as plain K-contiguous MFMA GEMMs (gemm_f32.hip).  conv1 is write-bound (C*4 bytes per 9 FMAs):
// each thread owns 4 consecutive channels (one 16-byte store) and keeps its 36 taps in registers.
#include "cfm_common.h"

namespace {

__global__ __launch_bounds__(256) void conv1_relu_kernel(const float* __restrict__ x, const float* __restrict__ w1,
                                                         const float* __restrict__ b1, float* __restrict__ h1,
                                                         int B, int F, int T, int C, int F1, int T1, int ppb) {
    const int c4n = C >> 2;
    const int c4 = threadIdx.x % c4n;
    const int pl = threadIdx.x / c4n;
    if (pl >= ppb) return;
    float w[4][9];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 9; ++j) w[i][j] = w1[(c4 * 4 + i) * 9 + j];
    const f32x4 bb = *reinterpret_cast<const f32x4*>(b1 + c4 * 4);
    const int64_t npos = (int64_t)B * T1 * F1;
    for (int64_t pos = (int64_t)blockIdx.x * ppb + pl; pos < npos; pos += (int64_t)gridDim.x * ppb) {
        const int f1 = (int)(pos % F1);
        const int64_t bt = pos / F1;
        const int t1 = (int)(bt % T1);
        const int64_t b = bt / T1;
        const float* xp = x + (b * F + 2 * f1) * (int64_t)T + 2 * t1;
        float xv[9];
#pragma unroll
        for (int kf = 0; kf < 3; ++kf)
#pragma unroll
            for (int kt = 0; kt < 3; ++kt) xv[kf * 3 + kt] = xp[(int64_t)kf * T + kt];
        f32x4 o = bb;
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            o.x = fmaf(w[0][j], xv[j], o.x);
            o.y = fmaf(w[1][j], xv[j], o.y);
            o.z = fmaf(w[2][j], xv[j], o.z);
            o.w = fmaf(w[3][j], xv[j], o.w);
        }
        o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
        *reinterpret_cast<f32x4*>(h1 + pos * C + c4 * 4) = o;
    }
}

// w2 (Co, Ci, 3, 3) -> w2p (Co, 3(kf), 3(kt), Ci)
__global__ __launch_bounds__(256) void pack_conv2_kernel(const float* __restrict__ w2, float* __restrict__ w2p, int C) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)C * C * 9;
    if (idx >= total) return;
    const int ci = (int)(idx % C);
    const int tap = (int)((idx / C) % 9);
    const int64_t co = idx / ((int64_t)C * 9);
    w2p[idx] = w2[(co * C + ci) * 9 + tap];
}

// wl (d, C*F2) with column c*F2+f  ->  wlp (d, F2*C) with column f*C+c
__global__ __launch_bounds__(256) void pack_linear_kernel(const float* __restrict__ wl, float* __restrict__ wlp,
                                                          int d_out, int C, int F2) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t kk = (int64_t)C * F2;
    if (idx >= kk * d_out) return;
    const int c = (int)(idx % C);
    const int f = (int)((idx / C) % F2);
    const int64_t n = idx / kk;
    wlp[idx] = wl[n * kk + (int64_t)c * F2 + f];
}

}  // namespace

extern "C" int cfm_subsample_conv1_relu_f32(const float* x, const float* w1, const float* b1, float* h1, int B, int F,
                                            int T, int C, cfm_stream_t stream) {
    CFM_REQUIRE(x && w1 && b1 && h1, CFM_ERR_NULL);
    CFM_REQUIRE(B > 0 && F >= 3 && T >= 3 && C > 0 && (C & 3) == 0, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(C <= 1024, CFM_ERR_UNSUPPORTED);
    CFM_REQUIRE(CFM_ALIGNED16(h1) && CFM_ALIGNED16(b1), CFM_ERR_ALIGN);
    const int F1 = (F - 1) / 2, T1 = (T - 1) / 2;
    const int ppb = 256 / (C / 4);
    const int64_t npos = (int64_t)B * T1 * F1;
    int64_t blocks = (npos + ppb - 1) / ppb;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(conv1_relu_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), x, w1,
                       b1, h1, B, F, T, C, F1, T1, ppb);
    return cfm_launch_status();
}

extern "C" int cfm_pack_conv2_weight_f32(const float* w2, float* w2p, int C, cfm_stream_t stream) {
    CFM_REQUIRE(w2 && w2p, CFM_ERR_NULL);
    CFM_REQUIRE(C > 0, CFM_ERR_BAD_SHAPE);
    const int64_t total = (int64_t)C * C * 9;
    hipLaunchKernelGGL(pack_conv2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), w2, w2p, C);
    return cfm_launch_status();
}

extern "C" int cfm_pack_linear_weight_f32(const float* wl, float* wlp, int d_out, int C, int F2, cfm_stream_t stream) {
    CFM_REQUIRE(wl && wlp, CFM_ERR_NULL);
    CFM_REQUIRE(d_out > 0 && C > 0 && F2 > 0, CFM_ERR_BAD_SHAPE);
    const int64_t total = (int64_t)d_out * C * F2;
    hipLaunchKernelGGL(pack_linear_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), wl, wlp, d_out, C, F2);
    return cfm_launch_status();
}
