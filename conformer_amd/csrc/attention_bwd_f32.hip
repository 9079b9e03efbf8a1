// Backward of the relative-position attention core -- round-1 form: the five products run as batched MFMA GEMMs
// (gemm_bwd_f32.hip) over re-materialised (B,H,T,T) probability / score-gradient tensors; this file holds the glue
// kernels between them.  (The forward never materialises scores; a flash-style fused backward that recomputes tiles
// the same way is the planned replacement -- DESIGN.md section 7.)
//
//   Qu = q + u_h, Qv = q + v_h                                      attn_qbias
//   content = Qu.K^T, posfull = Qv.Pm_h^T, dP = dO.V^T              batched GEMMs (caller)
//   D_i = dO_i . O_i                                                attn_rowdot
//   s = (content[i,k] + posfull[i, k-i+T-1]) * scale; P = exp(s - lse_i) (0 for masked keys)
//   dS = P * (dP - D_i) * scale; dposfull[i, k-i+T-1] = dS[i,k], 0 elsewhere      attn_softmax_bwd (in place)
//   dV = P^T.dO, dK = dS^T.Qu, dQu = dS.K, dQv = dposfull.Pm_h, dPm_h = sum_b dposfull^T.Qv   batched GEMMs (caller)
#include "cfm_common.h"

namespace {

__global__ __launch_bounds__(256) void attn_qbias_kernel(const float* __restrict__ q, int64_t ld,
                                                         const float* __restrict__ u, const float* __restrict__ vb,
                                                         float* __restrict__ qu, float* __restrict__ qv, int64_t rows,
                                                         int d) {
    const int nv = d >> 2;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * nv) return;
    const int64_t r = idx / nv;
    const int c = (int)(idx - r * nv) * 4;
    const f32x4 x = *reinterpret_cast<const f32x4*>(q + r * ld + c);
    *reinterpret_cast<f32x4*>(qu + r * d + c) = x + *reinterpret_cast<const f32x4*>(u + c);
    *reinterpret_cast<f32x4*>(qv + r * d + c) = x + *reinterpret_cast<const f32x4*>(vb + c);
}

// D[b,h,i] = sum_c dO[b,i,h,c] * O[b,i,h,c]; one wave per (b,i) row, 64/H... lanes split over heads
__global__ __launch_bounds__(256) void attn_rowdot_kernel(const float* __restrict__ dO, const float* __restrict__ O,
                                                          float* __restrict__ D, int B, int T, int H, int dh) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (int64_t)B * T) return;
    const int b = (int)(row / T), i = (int)(row % T);
    const int d = H * dh;
    for (int h = 0; h < H; ++h) {
        float s = 0.f;
        for (int c = lane; c < dh; c += 64) s += dO[row * d + h * dh + c] * O[row * d + h * dh + c];
        s = wave_sum(s);
        if (lane == 0) D[((int64_t)b * H + h) * T + i] = s;
    }
}

// one wave per (b,h,i) row
__global__ __launch_bounds__(256) void attn_softmax_bwd_kernel(float* __restrict__ content,   // (B,H,T,T4) -> P
                                                               float* __restrict__ posfull,   // (H,B,T,P4) -> dposfull
                                                               float* __restrict__ dP,        // (B,H,T,T4) -> dS
                                                               const float* __restrict__ lse, const float* __restrict__ D,
                                                               const int64_t* __restrict__ lengths, float scale, int B,
                                                               int T, int H, int T4, int P4, float drop_p,
                                                               unsigned long long drop_seed) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);     // (b*H + h)*T + i
    if (row >= (int64_t)B * H * T) return;
    const int i = (int)(row % T);
    const int64_t bh = row / T;
    const int h = (int)(bh % H), b = (int)(bh / H);
    int klen = T;
    if (lengths) { const int64_t L = lengths[b]; if (L < T) klen = (int)L; }
    float* crow = content + row * T4;
    float* drow = dP + row * T4;
    float* prow = posfull + (((int64_t)h * B + b) * T + i) * P4;
    const float l = lse[row], Di = D[row];
    const int jlo = T - 1 - i;                                               // band: j = k + jlo, k in [0,T)
    for (int j = lane; j < P4; j += 64)
        if (j < jlo || j >= jlo + T) prow[j] = 0.f;                          // outside the band: no gradient
    for (int k = lane; k < T4; k += 64) {
        float p = 0.f, ds = 0.f;
        if (k < T) {
            if (k < klen) {
                const float s = (crow[k] + prow[k + jlo]) * scale;
                p = exp_fast(s - l);
                // forward dropped the weights: ctx = sum_k a*m*v, m in {0, 1/(1-p)}: d(a) = dP*m and the P that feeds dV is a*m
                const float m = drop_p > 0.f ? dropout_keep(drop_seed, (unsigned long long)row * (unsigned long long)T + (unsigned)k,
                                                           drop_p, 1.0f / (1.0f - drop_p)) : 1.0f;
                ds = p * (drow[k] * m - Di) * scale;
                p *= m;
            }
            prow[k + jlo] = ds;
        }
        crow[k] = p;
        drow[k] = ds;
    }
}

// y = x * keep(seed, flat index): the stand-alone form of the epilogue dropout (backward: mask the incoming gradient)
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n4,
                                                      float p, unsigned long long seed) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const float inv_keep = 1.0f / (1.0f - p);
    f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] *= dropout_keep(seed, (unsigned long long)(4 * i + e), p, inv_keep);
    reinterpret_cast<f32x4*>(y)[i] = v;
}

__global__ __launch_bounds__(256) void add_strided_kernel(float* __restrict__ dst, int64_t ldd,
                                                          const float* __restrict__ src, int64_t lds_, int64_t rows,
                                                          int cols) {
    const int nv = cols >> 2;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * nv) return;
    const int64_t r = idx / nv;
    const int c = (int)(idx - r * nv) * 4;
    f32x4* d4 = reinterpret_cast<f32x4*>(dst + r * ldd + c);
    *d4 = *d4 + *reinterpret_cast<const f32x4*>(src + r * lds_ + c);
}

}  // namespace

extern "C" int cfm_attn_qbias_f32(const float* q, int64_t ld, const float* u, const float* vbias, float* qu, float* qv,
                                  int64_t rows, int d, cfm_stream_t stream) {
    CFM_REQUIRE(q && u && vbias && qu && qv, CFM_ERR_NULL);
    CFM_REQUIRE(rows > 0 && d > 0 && (d & 3) == 0 && (ld & 3) == 0, CFM_ERR_BAD_SHAPE);
    const int64_t total = rows * (d / 4);
    hipLaunchKernelGGL(attn_qbias_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), q, ld, u, vbias, qu, qv, rows, d);
    return cfm_launch_status();
}

extern "C" int cfm_attn_rowdot_f32(const float* dO, const float* O, float* D, int B, int T, int H, int dh,
                                   cfm_stream_t stream) {
    CFM_REQUIRE(dO && O && D, CFM_ERR_NULL);
    CFM_REQUIRE(B > 0 && T > 0 && H > 0 && dh > 0, CFM_ERR_BAD_SHAPE);
    const int64_t rows = (int64_t)B * T;
    hipLaunchKernelGGL(attn_rowdot_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), dO, O, D, B, T, H, dh);
    return cfm_launch_status();
}

extern "C" int cfm_attn_softmax_bwd_f32(float* content_to_p, float* posfull_to_dposfull, float* dp_to_ds,
                                        const float* lse, const float* D, const int64_t* lengths_or_null, float scale,
                                        int B, int T, int H, int T4, int P4, float drop_p, uint64_t drop_seed,
                                        cfm_stream_t stream) {
    CFM_REQUIRE(content_to_p && posfull_to_dposfull && dp_to_ds && lse && D, CFM_ERR_NULL);
    CFM_REQUIRE(B > 0 && T > 0 && H > 0 && T4 >= T && P4 >= 2 * T - 1, CFM_ERR_BAD_SHAPE);
    const int64_t rows = (int64_t)B * H * T;
    hipLaunchKernelGGL(attn_softmax_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), content_to_p, posfull_to_dposfull, dp_to_ds, lse, D,
                       lengths_or_null, scale, B, T, H, T4, P4, drop_p, drop_seed);
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

extern "C" int cfm_add_strided_f32(float* dst, int64_t ld_dst, const float* src, int64_t ld_src, int64_t rows, int cols,
                                   cfm_stream_t stream) {
    CFM_REQUIRE(dst && src, CFM_ERR_NULL);
    CFM_REQUIRE(rows > 0 && cols > 0 && (cols & 3) == 0 && (ld_dst & 3) == 0 && (ld_src & 3) == 0, CFM_ERR_BAD_SHAPE);
    const int64_t total = rows * (cols / 4);
    hipLaunchKernelGGL(add_strided_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), dst, ld_dst, src, ld_src, rows, cols);
    return cfm_launch_status();
}
