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
// Weights are split once per parameter version (cfm_split_pack_bf16_f32, cached by the host side like the 16-bit weight
// copies) and stored in MFMA fragment order, so they stream from L2 straight into registers; activations are split on
// their way into LDS.  Block tile 128x128 / 128x64 / 64x64, K-tile 32, 4 waves (2x2), two workgroups per CU; shared
// 16-byte epilogue (gemm_shared.h).
// Measured on the FFN GEMM (7968 x 2048 x 512, 112 us against 160 us native in the same harness): the six MFMAs alone
// take 51 us (1.95 PFLOP/s, 78 % of the nominal bf16 peak -- the practical ceiling of the pipe), ~36 us are fixed
// (65 MB epilogue, ramp, barriers), ~30 us are operand staging that is not yet hidden behind the MFMAs.  Three loop
// structures (both operands through one LDS stage; K-tile 16 with two stages; this one) land within 5 % of each other.
// Also measured and dropped: one LDS stage + three workgroups per CU for the 128x64 tile (the N = 512 GEMMs have only two
// tiles per CU: -8 %), 64x64 tiles for those GEMMs (-15 %), letting the compiler schedule across the term groups (-8 %).
#include "gemm_shared.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 widen(const bf16x4 h) { return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]}; }

// Weight operand: pre-split AND pre-shuffled into MFMA fragment order by cfm_split_pack_bf16_f32 --
//     [row block of 32][K-step of 16][plane][lane = 32*hf + li][8 bf16]   (row = 32*block + li, k = 16*step + 8*hf + e)
// so the fragment one wave feeds one MFMA is ONE fully coalesced 1 KB load straight into registers: the weights never
// touch LDS (half of the kernel's LDS traffic and LDS footprint gone), consecutive K-steps of a row block are contiguous.
// Activation operand: fp32 rows -> registers -> split chain -> LDS planes (two stages), read back as 16-byte fragments.
// Per K-tile (32 = two MFMA K-steps) and wave: 2*TM*NPL ds_read_b128, 2*TN*NPL 1 KB global loads (issued one K-tile
// ahead), 2*TM*TN*terms MFMAs with the staging of the next activation tile sliced in between; one barrier.
template <int BM, int BN, int EPI, bool CONV, int NPL>
__global__ __launch_bounds__(256, 2) void gemm_split_kernel(const GemmArgs g) {
    constexpr int TM = BM / 64, TN = BN / 64, BK = 32;
    constexpr int ROWB = 40;                                  // LDS row in bf16 elements: 32 + 8 pad = 80 B
    constexpr int NA = BM / 32;                               // float4 loads per thread per K-tile (8 lanes per row)
    constexpr int STAGE = NPL * BM * ROWB;                    // [NPL][BM][ROWB]
    static_assert(EPI != EPI_GLU || TN == 2, "GLU keeps value and gate tiles in one wave");
    __shared__ __attribute__((aligned(16))) __bf16 lds[2 * STAGE];

    const unsigned nwg = g.tiles_m * g.tiles_n;
    const unsigned tile = xcd_remap(blockIdx.x, nwg);
    const unsigned tm = tile / g.tiles_n, tn = tile % g.tiles_n;
    const int64_t m0 = (int64_t)tm * BM;
    const int n0 = (int)tn * (EPI == EPI_GLU ? BN / 2 : BN);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int li = lane & 31, hf = lane >> 5;

    const int srow = tid >> 3, sch = tid & 7;                 // A: fp32, 8 lanes x 16 B = the 128 B of one tile row
    const float* a_ptr[NA];
#pragma unroll
    for (int p = 0; p < NA; ++p) a_ptr[p] = a_row_ptr<CONV>(g, m0 + srow + 32 * p);

    // this wave's weight row blocks (32 rows each) in the packed buffer
    const int ksteps = g.K / 16;                              // K % 16 == 0
    const int nblocks = (g.N + 31) / 32;
    const __bf16* w_blk[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t) {
        int row;
        if (EPI == EPI_GLU) row = n0 + wc * 32 + (t == 0 ? 0 : g.n_out);       // tile 0 = value rows, tile 1 = gate rows
        else row = n0 + wc * (BN / 2) + 32 * t;
        const int blk = min(row / 32, nblocks - 1);                            // beyond N: clamped, never stored
        w_blk[t] = reinterpret_cast<const __bf16*>(g.W) + ((int64_t)blk * ksteps * NPL * 64 + lane) * 8;
    }

    // Global loads run far ahead of their use (an iteration of 8*TM*TN*.. MFMAs is shorter than the memory latency):
    // weights: a ring of four K-step slots -- the slot a K-step has just consumed is refilled with the K-step four ahead
    // (two K-tiles); activations: two register sets, loaded two K-tiles ahead, split into LDS one K-tile ahead.
    struct WFrag { bf16x8 f[2][TN][NPL]; };                   // the two K-steps of one K-tile
    auto load_w_step = [&](int kt, int s, WFrag& w) {
        const int ks = min(kt * 2 + s, ksteps - 1);           // a K-step beyond K meets zeroed activations
#pragma unroll
        for (int t = 0; t < TN; ++t)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl)
                w.f[s][t][pl] = *reinterpret_cast<const bf16x8*>(w_blk[t] + ((int64_t)ks * NPL + pl) * 512);
    };
    struct ARegs { f32x4 v[NA]; bool valid; };
    auto load_a = [&](int kt, ARegs& r) {
        const int k = kt * BK + sch * 4;
        r.valid = k < g.K;
        const int kc = max(min(k, g.K - 4), 0);
        const int64_t aoff = a_k_offset<CONV>(g, kc - sch * 4) + sch * 4;
#pragma unroll
        for (int p = 0; p < NA; ++p) r.v[p] = *reinterpret_cast<const f32x4*>(a_ptr[p] + aoff);
    };
    auto store_a = [&](const ARegs& r, int p, __bf16* stage) {
        f32x4 v = r.valid ? r.v[p] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
            const bf16x4 h = Lowp<__bf16>::cvt4(v);                                       // RNE: |v - h| <= 2^-9 |v|
            *reinterpret_cast<bf16x4*>(stage + (pl * BM + srow + 32 * p) * ROWB + sch * 4) = h;
            if (pl + 1 < NPL) v = v - widen(h);                                           // exact
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int a_row = wr * (BM / 2) + li;
    // Every load and LDS store below is UNCONDITIONAL (addresses clamped, activations beyond K zeroed by a select): a load
    // inside a conditional block makes the compiler's s_waitcnt bookkeeping fall back to vmcnt(0) at the join, which
    // drains the whole prefetch ring every K-tile.  The K-tile count is rounded up to even (an extra tile multiplies zeros).
    const int nkt = (((g.K + BK - 1) / BK) + 1) & ~1;

    // K-tile kt: activations in LDS `cur`, weights in `w` (refilled with tile kt+2 as its K-steps retire); splits tile kt+1
    // (held in `a_next`) into `other`; loads the activations of tile kt+2 into `a_free`
    auto iteration = [&](int kt, const __bf16* cur, __bf16* other, WFrag& w, const ARegs& a_next, ARegs& a_free) {
        load_a(kt + 2, a_free);
        __builtin_amdgcn_sched_barrier(0);
        int unit = 0;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 fa[TM][NPL];
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
                for (int t = 0; t < TM; ++t)
                    fa[t][pl] = *reinterpret_cast<const bf16x8*>(cur + (pl * BM + a_row + 32 * t) * ROWB + 16 * s + 8 * hf);
            // term-major order: consecutive MFMAs hit different accumulators; smallest terms first; every product is exact
#pragma unroll
            for (int order = NPL - 1; order >= 0; --order)
#pragma unroll
                for (int i = 0; i <= order; ++i) {
#pragma unroll
                    for (int mt = 0; mt < TM; ++mt)
#pragma unroll
                        for (int nt = 0; nt < TN; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.f[s][nt][order - i], fa[mt][i], acc[mt][nt],
                                                                                  0, 0, 0);
                    // slices of the next tile's staging in the shadow of these MFMAs (NA slices over the term groups)
                    constexpr int SLOTS = 2 * (NPL * (NPL + 1) / 2);
#pragma unroll
                    for (int p = 0; p < NA; ++p)
                        if (p * SLOTS / NA == unit) store_a(a_next, p, other);
                    ++unit;
                    __builtin_amdgcn_sched_barrier(0);
                }
            load_w_step(kt + 2, s, w);                         // this K-step's slot is free again
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    };

    WFrag w0, w1;
    ARegs a0, a1;
    load_a(0, a0);
    load_w_step(0, 0, w0); load_w_step(0, 1, w0);
    load_a(1, a1); load_w_step(1, 0, w1); load_w_step(1, 1, w1);
#pragma unroll
    for (int p = 0; p < NA; ++p) store_a(a0, p, lds);
    __syncthreads();
    for (int kt = 0; kt < nkt; kt += 2) {
        iteration(kt, lds, lds + STAGE, w0, a1, a0);
        iteration(kt + 1, lds + STAGE, lds, w1, a0, a1);
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

// packed[block][kstep][plane][lane][8]: the exact bf16 expansion of W (N,K) in MFMA fragment order (rows >= N: zeros)
__global__ __launch_bounds__(256) void split_pack_kernel(const float* __restrict__ w, __bf16* __restrict__ dst, int N, int K,
                                                         int planes) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // one 8-element fragment slot per thread
    const int ksteps = K / 16;
    const int64_t slots = (int64_t)((N + 31) / 32) * ksteps * 64;
    if (i >= slots) return;
    const int lane = (int)(i & 63);
    const int64_t bk = i >> 6;
    const int ks = (int)(bk % ksteps), blk = (int)(bk / ksteps);
    const int row = blk * 32 + (lane & 31), k = ks * 16 + (lane >> 5) * 8;
    f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = lo;
    if (row < N) {
        lo = *reinterpret_cast<const f32x4*>(w + (int64_t)row * K + k);
        hi = *reinterpret_cast<const f32x4*>(w + (int64_t)row * K + k + 4);
    }
    for (int pl = 0; pl < planes; ++pl) {
        const bf16x4 a = Lowp<__bf16>::cvt4(lo), b = Lowp<__bf16>::cvt4(hi);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[e] = a[e]; o[4 + e] = b[e]; }
        *reinterpret_cast<bf16x8*>(dst + ((bk * planes + pl) * 64 + lane) * 8) = o;
        lo = lo - widen(a); hi = hi - widen(b);
    }
}

}  // namespace

// planes: 3 (six products, fp32-level error) or 2 (three products, 2^-15).  W_split: the packed expansion made by
// cfm_split_pack_bf16_f32 from the (N,K) weight (for epi 3 / GLU: all 2*n_out rows, n_out % 32 == 0).  K % 16 == 0.  Everything else as cfm_gemm_mfma16_f32 with fp32 A and C.
extern "C" int cfm_gemm_split_bf16_f32(int planes, int epi, const float* A, const void* W_split, const float* bias,
                                       const float* R_or_null, float alpha, float* C, int64_t M, int N, int K, int64_t lda,
                                       int64_t ldr, int64_t ldc, cfm_stream_t stream) {
    CFM_REQUIRE(A && W_split && bias && C, CFM_ERR_NULL);
    CFM_REQUIRE(planes == 2 || planes == 3, CFM_ERR_UNSUPPORTED);
    CFM_REQUIRE(M > 0 && N > 0 && K > 0 && (K & 15) == 0 && (lda & 3) == 0 && lda >= K && ldc >= N, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(CFM_ALIGNED16(A) && CFM_ALIGNED16(W_split), CFM_ERR_ALIGN);
    GemmArgs g{};
    g.A = A; g.W = static_cast<const float*>(W_split); g.bias = bias; g.R = R_or_null; g.C = C; g.M = M; g.K = K;
    g.lda = lda; g.ldr = ldr; g.ldc = ldc; g.alpha = alpha;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (epi == EPI_GLU) {
        CFM_REQUIRE((N & 31) == 0, CFM_ERR_BAD_SHAPE);
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

// split-operand form of cfm_subsample_conv2_relu_f32 (C % 64 == 0); w2p_split: cfm_split_pack_bf16_f32 of the (C, 9C) packed conv weight
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

// dst <- the exact bf16 expansion of the (N,K) fp32 weight in MFMA fragment order: cfm_split_pack_elems(planes,N,K) bf16
// elements.  K % 16 == 0.
extern "C" int64_t cfm_split_pack_elems(int planes, int N, int K) {
    if ((planes != 2 && planes != 3) || N <= 0 || K <= 0 || (K & 15)) return -1;
    return (int64_t)planes * ((N + 31) / 32) * 32 * K;
}

extern "C" int cfm_split_pack_bf16_f32(int planes, const float* w, void* dst, int N, int K, cfm_stream_t stream) {
    CFM_REQUIRE(w && dst, CFM_ERR_NULL);
    CFM_REQUIRE(planes == 2 || planes == 3, CFM_ERR_UNSUPPORTED);
    CFM_REQUIRE(N > 0 && K > 0 && (K & 15) == 0, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(CFM_ALIGNED16(w) && CFM_ALIGNED16(dst), CFM_ERR_ALIGN);
    const int64_t slots = (int64_t)((N + 31) / 32) * (K / 16) * 64;
    hipLaunchKernelGGL(split_pack_kernel, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       w, static_cast<__bf16*>(dst), N, K, planes);
    return cfm_launch_status();
}
