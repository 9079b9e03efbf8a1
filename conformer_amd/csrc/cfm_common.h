// Internal helpers shared by the gfx950 kernels of libconformer_hip.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/conformer_hip.h"

#define CFM_WAVE 64

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CFM_REQUIRE(cond, code) do { if (!(cond)) return (code); } while (0)
#define CFM_ALIGNED16(p) ((reinterpret_cast<uintptr_t>(p) & 15u) == 0)

static inline int cfm_launch_status() {
    return hipGetLastError() == hipSuccess ? CFM_OK : CFM_ERR_LAUNCH;
}

// full-wave (64 lane) butterfly reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// exp / sigmoid / swish on the hardware transcendental pipe: v_exp_f32 (2^x, 1 ulp) and v_rcp_f32 (1 ulp) instead of the
// ~15-instruction libm expf and the ~10-instruction IEEE divide.  exp(x) = 2^(x*log2e): the rounding of x*log2e adds
// |x|*2^-24 to the exponent, i.e. a relative error <= ~1.2e-7*(1+|x|) -- 1e-6 at |x| = 8, far inside the 1e-5 per-op
// budget.  These sit in GEMM epilogues and the attention inner loop, where the accurate forms cost
// up to a quarter of a short-K GEMM's run time.
__device__ __forceinline__ float exp_fast(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ float rcp_fast(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float sigmoidf_acc(float x) { return rcp_fast(1.0f + exp_fast(-x)); }
__device__ __forceinline__ float swishf_acc(float x) { return x * sigmoidf_acc(x); }

// Dropout keep-factor for element `idx` of a tensor under (seed): a counter-based generator, so the backward regenerates exactly
// the forward's mask from (seed, idx) with no stored mask.  ONE splitmix64 finaliser serves the FOUR elements of an aligned
// group (idx >> 2); element idx & 3 takes 16 of its 64 bits: a GEMM epilogue's four consecutive columns cost one hash (the
// one-hash-per-element form was 3.4 ms of a 52 ms cfg-3 step at the reference's p = 0.1).  Returns 0 (dropped, probability
// ceil(65536 p) / 65536) or 1/(1-p).  The stream differs from torch's Philox -- parity runs use p = 0.
__device__ __forceinline__ unsigned long long dropout_hash(unsigned long long seed, unsigned long long group) {
    unsigned long long z = seed + group * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ unsigned dropout_threshold(float p) { return (unsigned)ceilf(p * 65536.0f); }   // drop iff bits < threshold
__device__ __forceinline__ float dropout_keep(unsigned long long seed, unsigned long long idx, float p, float inv_keep) {
    const unsigned long long z = dropout_hash(seed, idx >> 2);
    const unsigned bits = (unsigned)(z >> (16 * (unsigned)(idx & 3))) & 0xffffu;
    return bits < dropout_threshold(p) ? 0.f : inv_keep;
}
// the four elements idx4 .. idx4 + 3 of an ALIGNED group (idx4 % 4 == 0): one hash
__device__ __forceinline__ void dropout_keep4(unsigned long long seed, unsigned long long idx4, float p, float inv_keep, float (&keep)[4]) {
    const unsigned long long z = dropout_hash(seed, idx4 >> 2);
    const unsigned th = dropout_threshold(p), lo = (unsigned)z, hi = (unsigned)(z >> 32);
    keep[0] = (lo & 0xffffu) < th ? 0.f : inv_keep;
    keep[1] = (lo >> 16) < th ? 0.f : inv_keep;
    keep[2] = (hi & 0xffffu) < th ? 0.f : inv_keep;
    keep[3] = (hi >> 16) < th ? 0.f : inv_keep;
}
// four CONSECUTIVE elements a .. a + 3 at any alignment (attention rows: T is odd): the two groups they touch, two hashes
__device__ __forceinline__ void dropout_keep4u(unsigned long long seed, unsigned long long a, float p, float inv_keep, float (&keep)[4]) {
    const unsigned long long z0 = dropout_hash(seed, a >> 2), z1 = dropout_hash(seed, (a >> 2) + 1);
    const unsigned th = dropout_threshold(p), sh = 16 * (unsigned)(a & 3);
    // 128-bit window z1:z0 shifted right by sh bits (sh in {0,16,32,48})
    const unsigned long long w = sh == 0 ? z0 : (z0 >> sh) | (z1 << (64 - sh));
    const unsigned lo = (unsigned)w, hi = (unsigned)(w >> 32);
    keep[0] = (lo & 0xffffu) < th ? 0.f : inv_keep;
    keep[1] = (lo >> 16) < th ? 0.f : inv_keep;
    keep[2] = (hi & 0xffffu) < th ? 0.f : inv_keep;
    keep[3] = (hi >> 16) < th ? 0.f : inv_keep;
}

// Bijective XCD-aware remap of a 1-D block id: blocks that share an XCD (id % 8 under the observed
// round-robin placement) get a contiguous chunk of the logical grid so neighbouring tiles reuse that
// XCD's L2.  Speed only -- never correctness.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
    const unsigned q = nwg >> 3, r = nwg & 7u, x = bid & 7u, i = bid >> 3;
    const unsigned base = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + i;
}

// Depthwise-conv taps of the 64 channels a workgroup owns: the (C,K) weight block [c0*K, (c0+64)*K) is contiguous, so
// it is read with coalesced loads into LDS and each lane then picks its K taps (stride K words: conflict-free for odd K)
// -- the direct per-lane read w[c*K + j] touches 64 cache lines per instruction.  All 256 threads must call it.
template <int K, bool FLIP = false>
__device__ __forceinline__ void load_taps(const float* __restrict__ w, int c0, int C, float* taps_lds /* [64*K] */,
                                          float (&wr)[K]) {
    const int64_t base = (int64_t)c0 * K, lim = (int64_t)C * K;
    for (int f = threadIdx.x; f < 64 * K; f += blockDim.x) taps_lds[f] = base + f < lim ? w[base + f] : 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int j = 0; j < K; ++j) wr[j] = taps_lds[lane * K + (FLIP ? K - 1 - j : j)];
}

// Sliding window of N frames of one channel column (frame stride C) starting at frame t_first: out-of-range frames
// read a clamped address and are zeroed by a select, so all N loads are unconditional and issue back to back (a
// conditional load compiles to a branch + s_waitcnt per frame and serialises the memory latency).
template <int N>
__device__ __forceinline__ void load_window(const float* __restrict__ col, int t_first, int T, int C, float (&win)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) win[i] = col[(int64_t)min(max(t_first + i, 0), T - 1) * C];
    __builtin_amdgcn_sched_barrier(0);          // keep the scheduler from sinking each load next to its first use
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int t = t_first + i;
        win[i] = (t >= 0 && t < T) ? win[i] : 0.f;
    }
}

// 16-bit matrix-pipe element types (gemm_mfma16.hip, gemm_bwd_mfma16.hip): conversions are RNE, accumulation is fp32.
template <typename T16> struct Lowp;
template <> struct Lowp<__bf16> {
    typedef __bf16 x8 __attribute__((ext_vector_type(8)));
    typedef __bf16 x4 __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ x4 cvt4(const f32x4 v) {
        x4 r;
        r[0] = (__bf16)v.x; r[1] = (__bf16)v.y; r[2] = (__bf16)v.z; r[3] = (__bf16)v.w;
        return r;
    }
    static __device__ __forceinline__ f32x16 mfma(const x8 a, const x8 b, const f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct Lowp<_Float16> {
    typedef _Float16 x8 __attribute__((ext_vector_type(8)));
    typedef _Float16 x4 __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ x4 cvt4(const f32x4 v) {
        x4 r;
        r[0] = (_Float16)v.x; r[1] = (_Float16)v.y; r[2] = (_Float16)v.z; r[3] = (_Float16)v.w;
        return r;
    }
    static __device__ __forceinline__ f32x16 mfma(const x8 a, const x8 b, const f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};
