// Convolution-subsampling stem, first half: 3x3/stride-2 conv (C_in = 1) + ReLU writing a CHANNEL-LAST
// activation h1 (B,T1,F1,C), plus the two one-time weight re-layouts that let the second conv and the input
// Linear run as plain K-contiguous MFMA GEMMs (gemm_f32.hip).  conv1 is write-bound (C*4 bytes per 9 FMAs):
// each thread owns 4 consecutive channels (one 16-byte store) and keeps its 36 taps in registers.
#include <stdlib.h>

#include "cfm_common.h"

namespace {

// A workgroup owns one (utterance, output frame t1) row of h1 -- F1 x C contiguous outputs -- and walks f1: no index
// arithmetic per position.  (The first version flattened (b, t1, f1) into one 64-bit position index and paid two 64-bit
// divisions per position and thread: it ran at 1.9 TB/s of stores, ALU-bound on the divisions, not write-bound.)
// h1 (1.28 GB at cfg-2, read back only after the whole tensor is written) is stored NON-TEMPORALLY: 356 -> 245 us (3.6 -> 5.2 TB/s);
// the 16-bit h1 of the autocast path: 256 -> 172 us (tools/conv1_time.py; NT = false restores plain stores for an A/B).
// CPT channels per thread: 4 (one 16-byte fp32 store) or, for a 16-bit h1, 8 (one 16-byte store of eight values: the write-out
// is bound by the number of store instructions, not by their width).
template <typename TOUT, int CPT, bool NT = true>   // TOUT: float, or the 16-bit matrix-pipe type when h1 only feeds the 16-bit conv2 GEMM
__global__ __launch_bounds__(256) void conv1_relu_kernel(const float* __restrict__ x, const float* __restrict__ w1,
                                                         const float* __restrict__ b1, TOUT* __restrict__ h1,
                                                         int B, int F, int T, int C, int F1, int T1, int ppb) {
    const int cgn = C / CPT;
    const int cg = threadIdx.x % cgn;
    const int pl = threadIdx.x / cgn;
    if (pl >= ppb) return;
    float w[CPT][9], bb[CPT];
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
#pragma unroll
        for (int j = 0; j < 9; ++j) w[i][j] = w1[(cg * CPT + i) * 9 + j];
        bb[i] = b1[cg * CPT + i];
    }
    const int t1 = blockIdx.x, b = blockIdx.y;
    const float* xb = x + (int64_t)b * F * T + 2 * t1;
    TOUT* hrow = h1 + ((int64_t)b * T1 + t1) * F1 * C + cg * CPT;
    constexpr int U = CPT == 4 ? 4 : 2;                  // positions in flight per thread: their 9-tap windows are requested together
    for (int f0 = pl; f0 < F1; f0 += U * ppb) {
        float xv[U][9];
#pragma unroll
        for (int u = 0; u < U; ++u) {                      // (positions beyond F1 re-read the last one and are not stored)
            const float* xp = xb + (int64_t)(2 * min(f0 + u * ppb, F1 - 1)) * T;
#pragma unroll
            for (int kf = 0; kf < 3; ++kf)
#pragma unroll
                for (int kt = 0; kt < 3; ++kt) xv[u][kf * 3 + kt] = xp[kf * T + kt];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int f1 = f0 + u * ppb;
            if (f1 >= F1) break;
            float o[CPT];
#pragma unroll
            for (int i = 0; i < CPT; ++i) {
                o[i] = bb[i];
#pragma unroll
                for (int j = 0; j < 9; ++j) o[i] = fmaf(w[i][j], xv[u][j], o[i]);
                o[i] = fmaxf(o[i], 0.f);
            }
            if constexpr (sizeof(TOUT) == 4) {
                static_assert(sizeof(TOUT) != 4 || CPT == 4, "fp32 h1: 4 channels per thread");
                __builtin_nontemporal_store(f32x4{o[0], o[1], o[2], o[3]}, reinterpret_cast<f32x4*>(hrow + (int64_t)f1 * C));
            } else if constexpr (CPT == 8) {
                typename Lowp<TOUT>::x8 r;
#pragma unroll
                for (int i = 0; i < 8; ++i) r[i] = (TOUT)o[i];
                if constexpr (NT) __builtin_nontemporal_store(r, reinterpret_cast<typename Lowp<TOUT>::x8*>(hrow + (int64_t)f1 * C));
                else *reinterpret_cast<typename Lowp<TOUT>::x8*>(hrow + (int64_t)f1 * C) = r;
            } else {
                *reinterpret_cast<typename Lowp<TOUT>::x4*>(hrow + (int64_t)f1 * C) = Lowp<TOUT>::cvt4(f32x4{o[0], o[1], o[2], o[3]});
            }
        }
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

// ---- backward pieces ----------------------------------------------------------------------------------------------
// class-packed transpose-conv weights: class q = 2*pt+pf, taps (kt,kf) with kt=pt, kf=pf (mod 2) in (kt outer, kf inner)
// order; w2c[q][ci][tap*C + co] = w2[co][ci][kf][kt]
__global__ __launch_bounds__(256) void pack_conv2_t_kernel(const float* __restrict__ w2, float* __restrict__ w2c, int C) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t cc = (int64_t)C * C;
    if (idx >= 9 * cc) return;
    const int64_t blk = idx / cc;                       // 0..8 in units of C*C
    const int q = blk < 4 ? 0 : (blk < 6 ? 1 : (blk < 8 ? 2 : 3));
    const int64_t qoff = (q == 0 ? 0 : (q == 1 ? 4 : (q == 2 ? 6 : 8))) * cc;
    const int pt = q >> 1, pf = q & 1;
    const int nkt = pt == 0 ? 2 : 1, nkf = pf == 0 ? 2 : 1, nt = nkt * nkf;
    const int64_t r = idx - qoff;                       // within the class block: [ci][tap][co]
    const int co = (int)(r % C);
    const int tap = (int)((r / C) % nt);
    const int64_t ci = r / ((int64_t)C * nt);
    const int kt = pt + 2 * (tap / nkf), kf = pf + 2 * (tap % nkf);
    w2c[idx] = w2[((co * (int64_t)C + ci) * 3 + kf) * 3 + kt];
}

// dz = dy where y > 0 else 0 (ReLU backward on the stored OUTPUT), float4
__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                       float* __restrict__ dz, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const f32x4 a = reinterpret_cast<const f32x4*>(y)[i];
    f32x4 d = reinterpret_cast<const f32x4*>(dy)[i];
    d.x = a.x > 0.f ? d.x : 0.f; d.y = a.y > 0.f ? d.y : 0.f; d.z = a.z > 0.f ? d.z : 0.f; d.w = a.w > 0.f ? d.w : 0.f;
    reinterpret_cast<f32x4*>(dz)[i] = d;
}

// the same with dz written in a 16-bit matrix-pipe type (its only consumers are 16-bit GEMM operands)
template <typename T16>
__global__ __launch_bounds__(256) void relu_bwd_out16_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                             T16* __restrict__ dz, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const f32x4 a = reinterpret_cast<const f32x4*>(y)[i];
    f32x4 d = reinterpret_cast<const f32x4*>(dy)[i];
    d.x = a.x > 0.f ? d.x : 0.f; d.y = a.y > 0.f ? d.y : 0.f; d.z = a.z > 0.f ? d.z : 0.f; d.w = a.w > 0.f ? d.w : 0.f;
    *reinterpret_cast<typename Lowp<T16>::x4*>(dz + 4 * i) = Lowp<T16>::cvt4(d);
}

// conv1 parameter gradients: dw1[c][kf][kt] += sum dz1 * x[b][2f1+kf][2t1+kt], db1[c] += sum dz1 with
// dz1 = dh1 where relu(conv1) > 0 (the pre-activation is recomputed from x: 9 FMAs, h1 need not be re-read).
// Same thread mapping as the forward kernel (4 channels per thread); block partials through LDS, one atomic per value.
// TD: the type dh1 is stored in (float, or a 16-bit matrix-pipe type under autocast).  The f1 loop runs four positions at a time
// with their dh1 / x loads issued together: one 16-byte load in flight per thread made the 2.5 GB pass latency-bound (1.2 ms
// at cfg-3, 2 TB/s).
template <typename TD>
__global__ __launch_bounds__(256) void conv1_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w1,
                                                        const float* __restrict__ b1, const TD* __restrict__ dh1,
                                                        float* __restrict__ dw1, float* __restrict__ db1, int B, int F,
                                                        int T, int C, int F1, int T1, int ppb) {
    const int c4n = C >> 2;
    const int c4 = threadIdx.x % c4n;
    const int pl = threadIdx.x / c4n;
    const bool act = pl < ppb;
    float w[4][9], acc[4][10];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 9; ++j) { w[i][j] = act ? w1[(c4 * 4 + i) * 9 + j] : 0.f; acc[i][j] = 0.f; }
        acc[i][9] = 0.f;
    }
    const f32x4 bb = act ? *reinterpret_cast<const f32x4*>(b1 + c4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    // grid = (frame groups, utterances): a workgroup walks the frames t1 = blockIdx.x, blockIdx.x + gridDim.x, ... of one
    // utterance and all f1 of each -- no per-position index arithmetic (the flattened 64-bit position index of the first
    // version cost two 64-bit divisions per position and thread)
    const int b = blockIdx.y;
    constexpr int U = 4;
    typedef TD td4 __attribute__((ext_vector_type(4)));
    if (act)
        for (int t1 = blockIdx.x; t1 < T1; t1 += gridDim.x)
        for (int f0 = pl; f0 < F1; f0 += U * ppb) {
            float xv[U][9];
            td4 dv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {                              // (positions beyond F1 read position f1 = F1-1: never used)
                const int f1 = min(f0 + u * ppb, F1 - 1);
                const float* xp = x + ((int64_t)b * F + 2 * f1) * T + 2 * t1;
#pragma unroll
                for (int kf = 0; kf < 3; ++kf)
#pragma unroll
                    for (int kt = 0; kt < 3; ++kt) xv[u][kf * 3 + kt] = xp[kf * T + kt];
                dv[u] = *reinterpret_cast<const td4*>(dh1 + (((int64_t)b * T1 + t1) * F1 + f1) * C + c4 * 4);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool ok = f0 + u * ppb < F1;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float pre = bb[i];
#pragma unroll
                    for (int j = 0; j < 9; ++j) pre = fmaf(w[i][j], xv[u][j], pre);
                    const float dz = (ok && pre > 0.f) ? (float)dv[u][i] : 0.f;
#pragma unroll
                    for (int j = 0; j < 9; ++j) acc[i][j] = fmaf(dz, xv[u][j], acc[i][j]);
                    acc[i][9] += dz;
                }
            }
        }
    // threads with the same c4 (different pl) hold partials of the same channels: combine via atomics on global
    if (act) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 9; ++j) atomicAdd(dw1 + (c4 * 4 + i) * 9 + j, acc[i][j]);
            atomicAdd(db1 + c4 * 4 + i, acc[i][9]);
        }
    }
}

}  // namespace

extern "C" int cfm_pack_conv2_weight_t_f32(const float* w2, float* w2c, int C, cfm_stream_t stream) {
    CFM_REQUIRE(w2 && w2c, CFM_ERR_NULL);
    CFM_REQUIRE(C > 0, CFM_ERR_BAD_SHAPE);
    const int64_t total = 9 * (int64_t)C * C;
    hipLaunchKernelGGL(pack_conv2_t_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), w2, w2c, C);
    return cfm_launch_status();
}

extern "C" int cfm_relu_bwd_f32(const float* y, const float* dy, float* dz, int64_t n, cfm_stream_t stream) {
    CFM_REQUIRE(y && dy && dz, CFM_ERR_NULL);
    CFM_REQUIRE(n > 0 && (n & 3) == 0, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(CFM_ALIGNED16(y) && CFM_ALIGNED16(dy) && CFM_ALIGNED16(dz), CFM_ERR_ALIGN);
    const int64_t n4 = n / 4;
    hipLaunchKernelGGL(relu_bwd_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), y, dy, dz, n4);
    return cfm_launch_status();
}

// cfm_relu_bwd_f32 with dz stored in the 16-bit type `prec` (a gradient that only feeds 16-bit GEMM operands: the stem's conv2
// backward under autocast)
extern "C" int cfm_relu_bwd_out16_f32(int prec, const float* y, const float* dy, void* dz16, int64_t n, cfm_stream_t stream) {
    CFM_REQUIRE(y && dy && dz16, CFM_ERR_NULL);
    CFM_REQUIRE(n > 0 && (n & 3) == 0, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(CFM_ALIGNED16(y) && CFM_ALIGNED16(dy) && (reinterpret_cast<uintptr_t>(dz16) & 7) == 0, CFM_ERR_ALIGN);
    const int64_t n4 = n / 4;
    const dim3 grid((unsigned)((n4 + 255) / 256));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (prec == CFM_PREC_BF16) hipLaunchKernelGGL(relu_bwd_out16_kernel<__bf16>, grid, dim3(256), 0, s, y, dy, static_cast<__bf16*>(dz16), n4);
    else if (prec == CFM_PREC_FP16) hipLaunchKernelGGL(relu_bwd_out16_kernel<_Float16>, grid, dim3(256), 0, s, y, dy, static_cast<_Float16*>(dz16), n4);
    else return CFM_ERR_UNSUPPORTED;
    return cfm_launch_status();
}

// dw1 (C,1,3,3), db1 (C) accumulated (caller zero-fills); dh1: gradient w.r.t. h1 = relu(conv1(x)) (B,T1,F1,C)
extern "C" int cfm_subsample_conv1_bwd_f32(const float* x, const float* w1, const float* b1, const float* dh1, float* dw1,
                                           float* db1, int B, int F, int T, int C, cfm_stream_t stream) {
    CFM_REQUIRE(x && w1 && b1 && dh1 && dw1 && db1, CFM_ERR_NULL);
    CFM_REQUIRE(B > 0 && F >= 3 && T >= 3 && C > 0 && (C & 3) == 0, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(C <= 1024, CFM_ERR_UNSUPPORTED);
    const int F1 = (F - 1) / 2, T1 = (T - 1) / 2;
    const int ppb = 256 / (C / 4);
    CFM_REQUIRE(B <= 65535, CFM_ERR_UNSUPPORTED);
    int groups = (768 + B - 1) / B;                     // ~768 workgroups in all: 40 atomics per thread at the end
    groups = groups < 1 ? 1 : (groups > T1 ? T1 : groups);
    hipLaunchKernelGGL(conv1_bwd_kernel<float>, dim3((unsigned)groups, (unsigned)B), dim3(256), 0, static_cast<hipStream_t>(stream),
                       x, w1, b1, dh1, dw1, db1, B, F, T, C, F1, T1, ppb);
    return cfm_launch_status();
}

// cfm_subsample_conv1_bwd_f32 with dh1 stored in the 16-bit type `prec` (cfm_subsample_conv2_bwd_input_fwdkernel_out16_mfma16_f32)
extern "C" int cfm_subsample_conv1_bwd_d16_f32(int prec, const float* x, const float* w1, const float* b1, const void* dh1_16,
                                               float* dw1, float* db1, int B, int F, int T, int C, cfm_stream_t stream) {
    CFM_REQUIRE(x && w1 && b1 && dh1_16 && dw1 && db1, CFM_ERR_NULL);
    CFM_REQUIRE(B > 0 && F >= 3 && T >= 3 && C > 0 && (C & 3) == 0, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(C <= 1024 && B <= 65535, CFM_ERR_UNSUPPORTED);
    CFM_REQUIRE((reinterpret_cast<uintptr_t>(dh1_16) & 7) == 0, CFM_ERR_ALIGN);
    const int F1 = (F - 1) / 2, T1 = (T - 1) / 2;
    const int ppb = 256 / (C / 4);
    int groups = (768 + B - 1) / B;
    groups = groups < 1 ? 1 : (groups > T1 ? T1 : groups);
    const dim3 grid((unsigned)groups, (unsigned)B);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (prec == CFM_PREC_BF16)
        hipLaunchKernelGGL(conv1_bwd_kernel<__bf16>, grid, dim3(256), 0, s, x, w1, b1, static_cast<const __bf16*>(dh1_16), dw1, db1, B, F,
                           T, C, F1, T1, ppb);
    else if (prec == CFM_PREC_FP16)
        hipLaunchKernelGGL(conv1_bwd_kernel<_Float16>, grid, dim3(256), 0, s, x, w1, b1, static_cast<const _Float16*>(dh1_16), dw1, db1,
                           B, F, T, C, F1, T1, ppb);
    else return CFM_ERR_UNSUPPORTED;
    return cfm_launch_status();
}

extern "C" int cfm_subsample_conv1_relu_f32(const float* x, const float* w1, const float* b1, float* h1, int B, int F,
                                            int T, int C, cfm_stream_t stream) {
    CFM_REQUIRE(x && w1 && b1 && h1, CFM_ERR_NULL);
    CFM_REQUIRE(B > 0 && F >= 3 && T >= 3 && C > 0 && (C & 3) == 0, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(C <= 1024, CFM_ERR_UNSUPPORTED);
    CFM_REQUIRE(CFM_ALIGNED16(h1) && CFM_ALIGNED16(b1), CFM_ERR_ALIGN);
    const int F1 = (F - 1) / 2, T1 = (T - 1) / 2;
    const int ppb = 256 / (C / 4);
    CFM_REQUIRE(B <= 65535, CFM_ERR_UNSUPPORTED);
    hipLaunchKernelGGL((conv1_relu_kernel<float, 4>), dim3((unsigned)T1, (unsigned)B), dim3(256), 0, static_cast<hipStream_t>(stream), x,
                       w1, b1, h1, B, F, T, C, F1, T1, ppb);
    return cfm_launch_status();
}

// cfm_subsample_conv1_relu_f32 with h1 written in the 16-bit type `prec` (the A operand of cfm_subsample_conv2_relu_mfma16_f32
// with a_is_16bit): 0.64 GB instead of 1.28 GB at cfg-2.
extern "C" int cfm_subsample_conv1_relu_out16_f32(int prec, const float* x, const float* w1, const float* b1, void* h1, int B,
                                                  int F, int T, int C, cfm_stream_t stream) {
    CFM_REQUIRE(x && w1 && b1 && h1, CFM_ERR_NULL);
    CFM_REQUIRE(B > 0 && F >= 3 && T >= 3 && C > 0 && (C & 3) == 0 && C <= 1024, CFM_ERR_BAD_SHAPE);
    const int F1 = (F - 1) / 2, T1 = (T - 1) / 2;
    const bool wide = (C & 7) == 0 && (reinterpret_cast<uintptr_t>(h1) & 15) == 0;     // eight channels per thread, 16-byte stores
    const int cpt = wide ? 8 : 4;
    const int ppb = 256 / (C / cpt) > 0 ? 256 / (C / cpt) : 0;
    CFM_REQUIRE(ppb > 0, CFM_ERR_UNSUPPORTED);
    CFM_REQUIRE(B <= 65535, CFM_ERR_UNSUPPORTED);
    const dim3 grid((unsigned)T1, (unsigned)B);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (prec == CFM_PREC_BF16) {
        if (wide) hipLaunchKernelGGL((conv1_relu_kernel<__bf16, 8>), grid, dim3(256), 0, s, x, w1, b1, static_cast<__bf16*>(h1), B, F, T, C, F1, T1, ppb);
        else hipLaunchKernelGGL((conv1_relu_kernel<__bf16, 4>), grid, dim3(256), 0, s, x, w1, b1, static_cast<__bf16*>(h1), B, F, T, C, F1, T1, ppb);
    } else if (prec == CFM_PREC_FP16) {
        if (wide) hipLaunchKernelGGL((conv1_relu_kernel<_Float16, 8>), grid, dim3(256), 0, s, x, w1, b1, static_cast<_Float16*>(h1), B, F, T, C, F1, T1, ppb);
        else hipLaunchKernelGGL((conv1_relu_kernel<_Float16, 4>), grid, dim3(256), 0, s, x, w1, b1, static_cast<_Float16*>(h1), B, F, T, C, F1, T1, ppb);
    } else return CFM_ERR_UNSUPPORTED;
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
