// 16-bit-MFMA GEMM with fp32 storage: C = epilogue(r16(A) . r16(W)^T + bias), fp32 accumulate, fp32 in/out, where r16
// rounds to bfloat16 (CFM_PREC_BF16) or IEEE half (CFM_PREC_FP16: the reference's own AMP dtype, train.py:6,232).
//
// This is the arithmetic torch.autocast gives nn.Linear / Conv (inputs rounded to the 16-bit type, fp32 accumulation;
// SURVEY.md Appendix D).  Operands stay fp32 in HBM -- LayerNorm, residual stream, softmax and every epilogue keep fp32
// (so results are never rounded to 16 bits, unlike autocast's) -- and are rounded (RNE) on their way into LDS, so no
// 16-bit copies of activations or weights ever exist.
// v_mfma_f32_32x32x16_bf16 runs at 16x the fp32 MFMA rate: the kernel is bound by the fp32 operand stream
// (L2 -> LDS), not by the matrix pipe, so the staging is built for bytes: K-tile 64 (each row of a tile is 256
// contiguous bytes = two full cache lines, 16 lanes x 16 B), registers -> bf16 -> LDS rows of 144 B (conflict-free
// ds_read_b128: one read = the 8 k-values one lane feeds one MFMA), LDS double buffer, one barrier per K-tile.
// Block tile 128x128 or 64x64, 4 waves (2x2); accumulators transposed (lanes = rows) for the shared 16-byte epilogue.
#include "gemm_shared.h"

namespace {

// W16: the weight operand is already stored in the 16-bit type (a cached cast of the fp32 master weights, made once per
// optimizer step): half the L2->LDS bytes of the operand that every row tile re-reads, and no conversion.
// A16 (with W16): the activation operand is a 16-bit tensor too (LayerNorm / Swish / stem conv1 outputs written in the
// 16-bit type by their producers), lda in elements; the stem's implicit-GEMM gather uses the same element offsets.
template <typename T16, int BM, int BN, int EPI, bool CONV, bool W16, bool A16 = false>
__global__ __launch_bounds__(256, 2) void gemm_mfma16_kernel(const GemmArgs g) {
    static_assert(!A16 || W16, "16-bit A operand comes together with 16-bit weights");
    using x8 = typename Lowp<T16>::x8;
    using x4 = typename Lowp<T16>::x4;
    constexpr int TM = BM / 64, TN = BN / 64, BK = 64;
    constexpr int ROWB = 72;                                  // LDS row in bf16 elements: 64 + 8 pad = 144 B
    constexpr int NA = BM / 16, NB = BN / 16;                 // float4 loads per thread per K-tile (rows / 16 passes)
    static_assert(EPI != EPI_GLU || TN == 2, "GLU keeps value and gate tiles in one wave");
    __shared__ __attribute__((aligned(16))) T16 lds[2 * (BM + BN) * ROWB];
    T16* As = lds;                     // [2][BM][ROWB]
    T16* Bs = lds + 2 * BM * ROWB;     // [2][BN][ROWB]

    const unsigned nwg = g.tiles_m * g.tiles_n;
    const unsigned tile = xcd_remap(blockIdx.x, nwg);
    const unsigned tm = tile / g.tiles_n, tn = tile % g.tiles_n;
    const int64_t m0 = (int64_t)tm * BM;
    const int n0 = (int)tn * (EPI == EPI_GLU ? BN / 2 : BN);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int li = lane & 31, hf = lane >> 5;

    // ---- staging: 16 lanes cover the 256 B of one tile row; pass p handles rows srow + 16 p
    const int srow = tid >> 4, sch = tid & 15;
    constexpr int NBH = BN / 32;                              // W16: 16-byte loads per thread per K-tile (8 lanes per row)
    const int hrow = tid >> 3, hch = tid & 7;
    constexpr int NAH = BM / 32;
    const float* a_ptr[A16 ? 1 : NA];
    const T16* ah_ptr[A16 ? NAH : 1];
    const float* w_ptr[W16 ? 1 : NB];
    const T16* wh_ptr[W16 ? NBH : 1];
    if (A16) {
#pragma unroll
        for (int p = 0; p < NAH; ++p) {
            ah_ptr[p] = reinterpret_cast<const T16*>(g.A) + (a_row_ptr<CONV>(g, m0 + hrow + 32 * p) - g.A);   // element offset
        }
    } else {
#pragma unroll
        for (int p = 0; p < NA; ++p) a_ptr[p] = a_row_ptr<CONV>(g, m0 + srow + 16 * p);
    }
    if (W16) {
#pragma unroll
        for (int p = 0; p < NBH; ++p)
            wh_ptr[p] = reinterpret_cast<const T16*>(g.W) + (int64_t)w_row_index<EPI, BN>(g, n0, hrow + 32 * p) * g.K;
    } else {
#pragma unroll
        for (int p = 0; p < NB; ++p) w_ptr[p] = w_row_ptr<EPI, BN>(g, n0, srow + 16 * p);
    }
    // All loads are unconditional: a K-tile chunk beyond K (only possible in the last tile when K % 64 != 0) reads a clamped
    // address and is zeroed by a select when it is staged into LDS (`kvalid*`), after the MFMAs of the current tile.
    f32x4 ra[A16 ? 1 : NA], rb[W16 ? 1 : NB];
    x8 rah[A16 ? NAH : 1], rbh[W16 ? NBH : 1];
    bool kvalid = true, kvalid_h = true;
    auto load_tile = [&](int kt) {
        const int k = kt * BK + sch * 4, kh = kt * BK + hch * 8;       // K % 4 == 0 (K % 8 == 0 for 16-bit operands)
        kvalid = k < g.K; kvalid_h = kh < g.K;
        const int kc = min(k, g.K - 4), khc = min(kh, g.K - 8);
        if (A16) {
            const int64_t ahoff = a_k_offset<CONV>(g, khc - hch * 8) + hch * 8;
#pragma unroll
            for (int p = 0; p < NAH; ++p) rah[p] = *reinterpret_cast<const x8*>(ah_ptr[p] + ahoff);
        } else {
            const int64_t aoff = a_k_offset<CONV>(g, kc - sch * 4) + sch * 4;
#pragma unroll
            for (int p = 0; p < NA; ++p) ra[p] = *reinterpret_cast<const f32x4*>(a_ptr[p] + aoff);
        }
        if (W16) {
#pragma unroll
            for (int p = 0; p < NBH; ++p) rbh[p] = *reinterpret_cast<const x8*>(wh_ptr[p] + khc);
        } else {
#pragma unroll
            for (int p = 0; p < NB; ++p) rb[p] = *reinterpret_cast<const f32x4*>(w_ptr[p] + kc);
        }
    };
    auto store_tile = [&](int buf) {
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        x8 z8;
#pragma unroll
        for (int e = 0; e < 8; ++e) z8[e] = (T16)0.f;
        if (A16) {
#pragma unroll
            for (int p = 0; p < NAH; ++p)
                *reinterpret_cast<x8*>(As + (buf * BM + hrow + 32 * p) * ROWB + hch * 8) = kvalid_h ? rah[p] : z8;
        } else {
#pragma unroll
            for (int p = 0; p < NA; ++p)
                *reinterpret_cast<x4*>(As + (buf * BM + srow + 16 * p) * ROWB + sch * 4) = Lowp<T16>::cvt4(kvalid ? ra[p] : z4);
        }
        if (W16) {
#pragma unroll
            for (int p = 0; p < NBH; ++p)
                *reinterpret_cast<x8*>(Bs + (buf * BN + hrow + 32 * p) * ROWB + hch * 8) = kvalid_h ? rbh[p] : z8;
        } else {
#pragma unroll
            for (int p = 0; p < NB; ++p)
                *reinterpret_cast<x4*>(Bs + (buf * BN + srow + 16 * p) * ROWB + sch * 4) = Lowp<T16>::cvt4(kvalid ? rb[p] : z4);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // lane (row li, half hf) feeds k = 16 s + 8 hf + {0..7} of MFMA step s: one 16-byte read
    const int a_row = wr * (BM / 2) + li, b_row = wc * (BN / 2) + li;
    const int nkt = (g.K + BK - 1) / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nkt;
        if (more) load_tile(kt + 1);
        __builtin_amdgcn_sched_barrier(0);                    // the loads stay in flight across this tile's MFMAs
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            x8 fa[TM], fb[TN];
#pragma unroll
            for (int t = 0; t < TM; ++t)
                fa[t] = *reinterpret_cast<const x8*>(As + (cur * BM + a_row + 32 * t) * ROWB + 16 * s + 8 * hf);
#pragma unroll
            for (int t = 0; t < TN; ++t)
                fb[t] = *reinterpret_cast<const x8*>(Bs + (cur * BN + b_row + 32 * t) * ROWB + 16 * s + 8 * hf);
#pragma unroll
            for (int mt = 0; mt < TM; ++mt)
#pragma unroll
                for (int nt = 0; nt < TN; ++nt)
                    acc[mt][nt] = Lowp<T16>::mfma(fb[nt], fa[mt], acc[mt][nt]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (more) store_tile(cur ^ 1);
        __syncthreads();
    }
    gemm_epilogue<BM, BN, EPI, TM, TN>(g, acc, m0, n0, wr, wc, li, hf);
}

template <typename T16, int BM, int BN, int EPI, bool CONV>
int launch_cfg(GemmArgs g, int src16, hipStream_t s) {          // src16: 0 = fp32 operands, 1 = 16-bit W, 2 = 16-bit A and W
    const int ncols = EPI == EPI_GLU ? g.n_out : g.N;
    const int bn = EPI == EPI_GLU ? BN / 2 : BN;
    g.tiles_m = (unsigned)((g.M + BM - 1) / BM);
    g.tiles_n = (unsigned)((ncols + bn - 1) / bn);
    const dim3 grid(g.tiles_m * g.tiles_n);
    if (src16 == 2) hipLaunchKernelGGL((gemm_mfma16_kernel<T16, BM, BN, EPI, CONV, true, true>), grid, dim3(256), 0, s, g);
    else if (src16 == 1) hipLaunchKernelGGL((gemm_mfma16_kernel<T16, BM, BN, EPI, CONV, true>), grid, dim3(256), 0, s, g);
    else if (src16 == 0) hipLaunchKernelGGL((gemm_mfma16_kernel<T16, BM, BN, EPI, CONV, false>), grid, dim3(256), 0, s, g);
    else return CFM_ERR_UNSUPPORTED;
    return cfm_launch_status();
}

template <typename T16, int EPI, bool CONV>
int launch_t(const GemmArgs& g, int src16, hipStream_t s) {
    const int ncols = EPI == EPI_GLU ? g.n_out : g.N;
    const int bn = EPI == EPI_GLU ? 64 : 128;
    const int64_t t128 = ((g.M + 127) / 128) * ((ncols + bn - 1) / bn);
    if constexpr (EPI == EPI_GLU) {
        return t128 >= 512 ? launch_cfg<T16, 128, 128, EPI, CONV>(g, src16, s) : launch_cfg<T16, 64, 128, EPI, CONV>(g, src16, s);
    } else {
        return t128 >= 512 ? launch_cfg<T16, 128, 128, EPI, CONV>(g, src16, s) : launch_cfg<T16, 64, 64, EPI, CONV>(g, src16, s);
    }
}

template <typename T16>
__global__ __launch_bounds__(256) void cast16_kernel(const float* __restrict__ src, T16* __restrict__ dst, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    *reinterpret_cast<typename Lowp<T16>::x4*>(dst + 4 * i) = Lowp<T16>::cvt4(reinterpret_cast<const f32x4*>(src)[i]);
}

template <int EPI, bool CONV>
int launch(int prec, const GemmArgs& g, int src16, hipStream_t s) {
    if (prec == CFM_PREC_BF16) return launch_t<__bf16, EPI, CONV>(g, src16, s);
    if (prec == CFM_PREC_FP16) return launch_t<_Float16, EPI, CONV>(g, src16, s);
    return CFM_ERR_UNSUPPORTED;
}

}  // namespace

// prec: CFM_PREC_BF16 | CFM_PREC_FP16.  epi: 0 bias | 1 bias+swish | 2 bias+relu | 3 bias+GLU (N = n_out columns of C,
// W has 2*n_out rows) | 4 alpha*y + R.  Same layouts and argument rules as the fp32 entry points (cfm_gemm_train_f32 for
// Z_or_null / drop_p / drop_seed); results differ from them by the 16-bit rounding of A and W only.
extern "C" int cfm_gemm_mfma16_f32(int prec, int epi, const void* A, int a_is_16bit, const void* W, int w_is_16bit,
                                   const float* bias, const float* R_or_null, float alpha, void* C, int c_is_16bit,
                                   float* Z_or_null, int64_t M, int N, int K, int64_t lda, int64_t ldr, int64_t ldc,
                                   float drop_p, uint64_t drop_seed, cfm_stream_t stream) {
    CFM_REQUIRE(A && W && bias && C, CFM_ERR_NULL);
    CFM_REQUIRE(!w_is_16bit || (K & 7) == 0, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(!a_is_16bit || (w_is_16bit && (lda & 7) == 0), CFM_ERR_BAD_SHAPE);
    const int src16 = a_is_16bit ? 2 : (w_is_16bit ? 1 : 0);
    CFM_REQUIRE(M > 0 && N > 0 && K > 0 && (K & 3) == 0 && (lda & 3) == 0 && lda >= K && ldc >= N, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(drop_p >= 0.f && drop_p < 1.f, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(CFM_ALIGNED16(A) && CFM_ALIGNED16(W), CFM_ERR_ALIGN);
    GemmArgs g{};
    g.A = static_cast<const float*>(A); g.W = static_cast<const float*>(W); g.bias = bias; g.R = R_or_null;
    g.C = static_cast<float*>(C); g.c_prec = c_is_16bit ? prec : 0; g.M = M; g.K = K; g.lda = lda;
    g.ldr = ldr; g.ldc = ldc; g.alpha = alpha; g.Zsave = Z_or_null; g.drop_p = drop_p; g.drop_seed = drop_seed;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (epi == EPI_GLU) {
        g.n_out = N; g.N = 2 * N;
        return launch<EPI_GLU, false>(prec, g, src16, s);
    }
    g.N = N;
    switch (epi) {
        case EPI_BIAS: return launch<EPI_BIAS, false>(prec, g, src16, s);
        case EPI_SWISH: return launch<EPI_SWISH, false>(prec, g, src16, s);
        case EPI_RELU: return launch<EPI_RELU, false>(prec, g, src16, s);
        case EPI_RESID:
            CFM_REQUIRE(R_or_null != nullptr, CFM_ERR_NULL);
            CFM_REQUIRE(ldr >= N, CFM_ERR_BAD_SHAPE);
            return launch<EPI_RESID, false>(prec, g, src16, s);
        default: return CFM_ERR_UNSUPPORTED;
    }
}

// 16-bit-MFMA form of cfm_subsample_conv2_relu_f32 (C % 64 == 0).  h1_is_16bit (with w_is_16bit): h1 comes from
// cfm_subsample_conv1_relu_out16_f32; h2_is_16bit: h2 is stored in `prec` (it feeds the input Linear's 16-bit A operand).
extern "C" int cfm_subsample_conv2_relu_mfma16_f32(int prec, const void* h1, int h1_is_16bit, const void* w2p, int w_is_16bit,
                                                   const float* b2, void* h2, int h2_is_16bit, int B, int F1, int T1, int C,
                                                   cfm_stream_t stream) {
    CFM_REQUIRE(h1 && w2p && b2 && h2, CFM_ERR_NULL);
    CFM_REQUIRE(B > 0 && F1 >= 3 && T1 >= 3 && C > 0, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(C % 64 == 0, CFM_ERR_UNSUPPORTED);
    CFM_REQUIRE(!h1_is_16bit || w_is_16bit, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(CFM_ALIGNED16(h1) && CFM_ALIGNED16(w2p) && (reinterpret_cast<uintptr_t>(h2) & 7) == 0, CFM_ERR_ALIGN);
    GemmArgs g{};
    g.cT1 = T1; g.cF1 = F1; g.cC = C; g.cT2 = (T1 - 1) / 2; g.cF2 = (F1 - 1) / 2;
    g.A = static_cast<const float*>(h1); g.W = static_cast<const float*>(w2p); g.bias = b2; g.C = static_cast<float*>(h2);
    g.c_prec = h2_is_16bit ? prec : 0;
    g.M = (int64_t)B * g.cT2 * g.cF2; g.N = C; g.K = 9 * C; g.lda = 0; g.ldc = C; g.alpha = 1.f;
    return launch<EPI_RELU, true>(prec, g, h1_is_16bit ? 2 : (w_is_16bit ? 1 : 0), static_cast<hipStream_t>(stream));
}

// dst (16-bit, prec) <- RNE(src) for n fp32 values (n % 4 == 0): the per-optimizer-step cast of the master weights that
// the w_is_16bit / b_is_16bit operands of the 16-bit GEMM entries consume.
extern "C" int cfm_cast16_f32(int prec, const float* src, void* dst, int64_t n, cfm_stream_t stream) {
    CFM_REQUIRE(src && dst, CFM_ERR_NULL);
    CFM_REQUIRE(n > 0 && (n & 3) == 0, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(CFM_ALIGNED16(src) && (reinterpret_cast<uintptr_t>(dst) & 7) == 0, CFM_ERR_ALIGN);
    const dim3 grid((unsigned)((n / 4 + 255) / 256));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (prec == CFM_PREC_BF16) hipLaunchKernelGGL(cast16_kernel<__bf16>, grid, dim3(256), 0, s, src, static_cast<__bf16*>(dst), n / 4);
    else if (prec == CFM_PREC_FP16) hipLaunchKernelGGL(cast16_kernel<_Float16>, grid, dim3(256), 0, s, src, static_cast<_Float16*>(dst), n / 4);
    else return CFM_ERR_UNSUPPORTED;
    return cfm_launch_status();
}
