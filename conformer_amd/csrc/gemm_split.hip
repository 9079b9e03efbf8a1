// fp32 GEMM on the bf16 matrix pipe by EXACT operand splitting ("bf16x6"; opt-in, conformer_amd.ops.set_fp32_matmul).
//
// gfx950 runs v_mfma_f32_32x32x16_bf16 at 16x the rate of v_mfma_f32_32x32x2_f32.  An fp32 value has a 24-bit
// significand = three 8-bit bf16 significands: x = x0 + x1 + x2 exactly (x0 = bf16(x), x1 = bf16(x - x0),
// x2 = bf16(x - x0 - x1); bf16 has fp32's exponent range: the expansion is exact for 2^-110 <= |x| < 3.39e38, i.e. for
// everything but values within 2^-9 of fp32's largest finite number and the last 16 binades above the denormals).  Then
//     a.w = sum_{i,j} a_i.w_j          (9 bf16 x bf16 products, each EXACT in the fp32 accumulator)
// and the three products with i + j >= 3 are below 2^-24 |a.w|: dropping them leaves a per-product relative error
// <= 2^-23, the same class as the fp32 FMA chain's own rounding (2^-24 of the accumulator per step).  Six bf16 MFMAs
// replace eight fp32 MFMAs' worth of K at 1/16 the cost each: 0.375x the matrix-pipe time of the native fp32 kernel,
// with fp32 inputs, fp32 accumulation, fp32 outputs and fp32-level error (measured in tests/test_split_gpu.py
// against the float64 oracle, side by side with the native fp32 kernel).  `planes` = 2 keeps x0 + x1 (three products,
// relative error <= 2^-15: the "3x" mode, still 60x inside the 1e-3 parity bar).
//
// Weights are split once per parameter version into bf16 planes [planes][N][K] (cfm_split_bf16_f32, cached by the
// host side like the 16-bit weight copies); activations are split on their way into LDS.  Block tile 128x128 / 128x64
// / 64x64, K-tile 32, 4 waves (2x2), LDS rows of 80 B (conflict-free ds_read_b128), single LDS stage with the next
// tile's global loads in flight across the MFMAs, two workgroups per CU; shared 16-byte epilogue (gemm_shared.h).
// (A two-stage K-tile-16 pipeline with the staging interleaved between the MFMAs measured 5-15 % slower: one barrier per
// 24 MFMAs and 32/64-byte global segments cost more than the overlap gains; the kernel is bound by LDS traffic -- 144 KB
// per 48 MFMAs per wave quartet, ~75 % of the CU's 128 B/clk when the matrix pipe is saturated.)
#include "gemm_shared.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 widen(const bf16x4 h) { return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]}; }

template <int BM, int BN, int EPI, bool CONV, int NPL>
__global__ __launch_bounds__(256, 2) void gemm_split_kernel(const GemmArgs g) {
    constexpr int TM = BM / 64, TN = BN / 64, BK = 32;
    constexpr int ROWB = 40;                                  // LDS row in bf16 elements: 32 + 8 pad = 80 B
    constexpr int NA = BM / 32;                               // float4 loads per thread per K-tile (8 lanes per row)
    constexpr int NBH = BN / 64;                              // 16-byte plane loads per thread per K-tile and plane (4 lanes per row)
    static_assert(EPI != EPI_GLU || TN == 2, "GLU keeps value and gate tiles in one wave");
    __shared__ __attribute__((aligned(16))) __bf16 lds[NPL * (BM + BN) * ROWB];
    __bf16* As = lds;                          // [NPL][BM][ROWB]
    __bf16* Bs = lds + NPL * BM * ROWB;        // [NPL][BN][ROWB]

    const unsigned nwg = g.tiles_m * g.tiles_n;
    const unsigned tile = xcd_remap(blockIdx.x, nwg);
    const unsigned tm = tile / g.tiles_n, tn = tile % g.tiles_n;
    const int64_t m0 = (int64_t)tm * BM;
    const int n0 = (int)tn * (EPI == EPI_GLU ? BN / 2 : BN);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int li = lane & 31, hf = lane >> 5;

    const int srow = tid >> 3, sch = tid & 7;                 // A: fp32, 8 lanes x 16 B = the 128 B of one tile row
    const int hrow = tid >> 2, hch = tid & 3;                 // W planes: bf16, 4 lanes x 16 B = the 64 B of one tile row
    const float* a_ptr[NA];
    const __bf16* w_ptr[NBH];
#pragma unroll
    for (int p = 0; p < NA; ++p) a_ptr[p] = a_row_ptr<CONV>(g, m0 + srow + 32 * p);
#pragma unroll
    for (int p = 0; p < NBH; ++p)
        w_ptr[p] = reinterpret_cast<const __bf16*>(g.W) + (int64_t)w_row_index<EPI, BN>(g, n0, hrow + 64 * p) * g.K;
    const int64_t plane = (int64_t)g.N * g.K;                 // elements between two planes of the split weight

    // unconditional clamped loads; chunks beyond K are zeroed when they are staged (`kvalid*`)
    f32x4 ra[NA];
    bf16x8 rb[NPL][NBH];
    bool kvalid = true, kvalid_h = true;
    auto load_tile = [&](int kt) {
        const int k = kt * BK + sch * 4, kh = kt * BK + hch * 8;       // K % 8 == 0
        kvalid = k < g.K; kvalid_h = kh < g.K;
        const int kc = max(min(k, g.K - 4), 0), khc = max(min(kh, g.K - 8), 0);
        const int64_t aoff = a_k_offset<CONV>(g, kc - sch * 4) + sch * 4;
#pragma unroll
        for (int p = 0; p < NA; ++p) ra[p] = *reinterpret_cast<const f32x4*>(a_ptr[p] + aoff);
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
            for (int p = 0; p < NBH; ++p) rb[pl][p] = *reinterpret_cast<const bf16x8*>(w_ptr[p] + pl * plane + khc);
    };
    auto store_tile = [&]() {
        bf16x8 z8;
#pragma unroll
        for (int e = 0; e < 8; ++e) z8[e] = (__bf16)0.f;
#pragma unroll
        for (int p = 0; p < NA; ++p) {
            f32x4 r = kvalid ? ra[p] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) {
                const bf16x4 h = Lowp<__bf16>::cvt4(r);                                   // RNE: |r - h| <= 2^-9 |r|
                *reinterpret_cast<bf16x4*>(As + (pl * BM + srow + 32 * p) * ROWB + sch * 4) = h;
                if (pl + 1 < NPL) r = r - widen(h);                                       // exact
            }
        }
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
            for (int p = 0; p < NBH; ++p)
                *reinterpret_cast<bf16x8*>(Bs + (pl * BN + hrow + 64 * p) * ROWB + hch * 8) = kvalid_h ? rb[pl][p] : z8;
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int a_row = wr * (BM / 2) + li, b_row = wc * (BN / 2) + li;
    const int nkt = (g.K + BK - 1) / BK;
    load_tile(0);
    for (int kt = 0; kt < nkt; ++kt) {
        store_tile();
        __syncthreads();
        if (kt + 1 < nkt) load_tile(kt + 1);
        __builtin_amdgcn_sched_barrier(0);                    // the loads stay in flight across this tile's MFMAs
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 fa[TM][NPL], fb[TN][NPL];
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
                for (int t = 0; t < TM; ++t)
                    fa[t][pl] = *reinterpret_cast<const bf16x8*>(As + (pl * BM + a_row + 32 * t) * ROWB + 16 * s + 8 * hf);
#pragma unroll
                for (int t = 0; t < TN; ++t)
                    fb[t][pl] = *reinterpret_cast<const bf16x8*>(Bs + (pl * BN + b_row + 32 * t) * ROWB + 16 * s + 8 * hf);
            }
#pragma unroll
            for (int mt = 0; mt < TM; ++mt)
#pragma unroll
                for (int nt = 0; nt < TN; ++nt) {
                    f32x16 c = acc[mt][nt];
                    // smallest terms first; every product is exact, the sums are fp32
#pragma unroll
                    for (int order = NPL - 1; order >= 0; --order)
#pragma unroll
                        for (int i = 0; i <= order; ++i)
                            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[nt][order - i], fa[mt][i], c, 0, 0, 0);
                    acc[mt][nt] = c;
                }
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    }
    gemm_epilogue<BM, BN, EPI, TM, TN>(g, acc, m0, n0, wr, wc, li, hf);
}

template <int BM, int BN, int EPI, bool CONV>
int launch_cfg(GemmArgs g, int planes, hipStream_t s) {
    const int ncols = EPI == EPI_GLU ? g.n_out : g.N;
    const int bn = EPI == EPI_GLU ? BN / 2 : BN;
    g.tiles_m = (unsigned)((g.M + BM - 1) / BM);
    g.tiles_n = (unsigned)((ncols + bn - 1) / bn);
    const dim3 grid(g.tiles_m * g.tiles_n);
    if (planes == 3) hipLaunchKernelGGL((gemm_split_kernel<BM, BN, EPI, CONV, 3>), grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((gemm_split_kernel<BM, BN, EPI, CONV, 2>), grid, dim3(256), 0, s, g);
    return cfm_launch_status();
}

template <int EPI, bool CONV>
int launch(const GemmArgs& g, int planes, hipStream_t s) {
    const int ncols = EPI == EPI_GLU ? g.n_out : g.N;
    const int64_t rows128 = (g.M + 127) / 128;
    if constexpr (EPI == EPI_GLU) {
        return rows128 * ((ncols + 63) / 64) >= 400 ? launch_cfg<128, 128, EPI, CONV>(g, planes, s)
                                                    : launch_cfg<64, 128, EPI, CONV>(g, planes, s);
    } else {
        if (rows128 * ((ncols + 127) / 128) >= 400) return launch_cfg<128, 128, EPI, CONV>(g, planes, s);
        if (rows128 * ((ncols + 63) / 64) >= 400) return launch_cfg<128, 64, EPI, CONV>(g, planes, s);
        return launch_cfg<64, 64, EPI, CONV>(g, planes, s);
    }
}

// dst[pl][i] = pl-th bf16 term of src[i] (RNE residual chain), pl < planes
__global__ __launch_bounds__(256) void split_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, int64_t n4,
                                                         int planes) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    f32x4 r = reinterpret_cast<const f32x4*>(src)[i];
    for (int pl = 0; pl < planes; ++pl) {
        const bf16x4 h = Lowp<__bf16>::cvt4(r);
        *reinterpret_cast<bf16x4*>(dst + (int64_t)pl * n4 * 4 + 4 * i) = h;
        r = r - widen(h);
    }
}

}  // namespace

// planes: 3 (six products, fp32-level error) or 2 (three products, 2^-15).  W_split: [planes][N][K] bf16 from
// cfm_split_bf16_f32 (for epi 3 / GLU: N = 2*n_out rows).  Everything else as cfm_gemm_mfma16_f32 with fp32 A and C.
extern "C" int cfm_gemm_split_bf16_f32(int planes, int epi, const float* A, const void* W_split, const float* bias,
                                       const float* R_or_null, float alpha, float* C, int64_t M, int N, int K, int64_t lda,
                                       int64_t ldr, int64_t ldc, cfm_stream_t stream) {
    CFM_REQUIRE(A && W_split && bias && C, CFM_ERR_NULL);
    CFM_REQUIRE(planes == 2 || planes == 3, CFM_ERR_UNSUPPORTED);
    CFM_REQUIRE(M > 0 && N > 0 && K > 0 && (K & 7) == 0 && (lda & 3) == 0 && lda >= K && ldc >= N, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(CFM_ALIGNED16(A) && CFM_ALIGNED16(W_split), CFM_ERR_ALIGN);
    GemmArgs g{};
    g.A = A; g.W = static_cast<const float*>(W_split); g.bias = bias; g.R = R_or_null; g.C = C; g.M = M; g.K = K;
    g.lda = lda; g.ldr = ldr; g.ldc = ldc; g.alpha = alpha;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (epi == EPI_GLU) {
        g.n_out = N; g.N = 2 * N;
        return launch<EPI_GLU, false>(g, planes, s);
    }
    g.N = N;
    switch (epi) {
        case EPI_BIAS: return launch<EPI_BIAS, false>(g, planes, s);
        case EPI_SWISH: return launch<EPI_SWISH, false>(g, planes, s);
        case EPI_RELU: return launch<EPI_RELU, false>(g, planes, s);
        case EPI_RESID:
            CFM_REQUIRE(R_or_null != nullptr, CFM_ERR_NULL);
            CFM_REQUIRE(ldr >= N, CFM_ERR_BAD_SHAPE);
            return launch<EPI_RESID, false>(g, planes, s);
        default: return CFM_ERR_UNSUPPORTED;
    }
}

// split-operand form of cfm_subsample_conv2_relu_f32 (C % 64 == 0); w2p_split: [planes][C][9C] bf16 of the packed weight
extern "C" int cfm_subsample_conv2_relu_split_bf16_f32(int planes, const float* h1, const void* w2p_split, const float* b2,
                                                       float* h2, int B, int F1, int T1, int C, cfm_stream_t stream) {
    CFM_REQUIRE(h1 && w2p_split && b2 && h2, CFM_ERR_NULL);
    CFM_REQUIRE(planes == 2 || planes == 3, CFM_ERR_UNSUPPORTED);
    CFM_REQUIRE(B > 0 && F1 >= 3 && T1 >= 3 && C > 0, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(C % 64 == 0, CFM_ERR_UNSUPPORTED);
    CFM_REQUIRE(CFM_ALIGNED16(h1) && CFM_ALIGNED16(w2p_split) && CFM_ALIGNED16(h2), CFM_ERR_ALIGN);
    GemmArgs g{};
    g.cT1 = T1; g.cF1 = F1; g.cC = C; g.cT2 = (T1 - 1) / 2; g.cF2 = (F1 - 1) / 2;
    g.A = h1; g.W = static_cast<const float*>(w2p_split); g.bias = b2; g.C = h2;
    g.M = (int64_t)B * g.cT2 * g.cF2; g.N = C; g.K = 9 * C; g.lda = 0; g.ldc = C; g.alpha = 1.f;
    return launch<EPI_RELU, true>(g, planes, static_cast<hipStream_t>(stream));
}

// dst [planes][n] bf16 <- the exact bf16 expansion of src (n % 4 == 0)
extern "C" int cfm_split_bf16_f32(int planes, const float* src, void* dst, int64_t n, cfm_stream_t stream) {
    CFM_REQUIRE(src && dst, CFM_ERR_NULL);
    CFM_REQUIRE(planes == 2 || planes == 3, CFM_ERR_UNSUPPORTED);
    CFM_REQUIRE(n > 0 && (n & 3) == 0, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(CFM_ALIGNED16(src) && (reinterpret_cast<uintptr_t>(dst) & 7) == 0, CFM_ERR_ALIGN);
    hipLaunchKernelGGL(split_bf16_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       src, static_cast<__bf16*>(dst), n / 4, planes);
    return cfm_launch_status();
}
