// fp32 GEMM family on the CDNA4 matrix pipe: C = epilogue(A . W^T + bias).
//
// v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain, 64 FLOP/clk/SIMD = 157 TFLOP/s chip peak).
// Block tile 128x128x16, 4 waves (2x2), each wave a 64x64 sub-tile = 2x2 MFMA tiles of 32x32
// (four independent accumulator chains per wave keep the 64-cycle MFMA issue rate).
// Global -> register -> LDS staging with a two-buffer LDS ring and one barrier per K step;
// the loads of K-tile t+1 are issued before the 32 MFMAs of tile t and written to LDS after them.
// LDS rows are padded to 20 floats (80 B) so the 16-byte MFMA-operand reads (ds_read_b128: one lane
// fetches 4 consecutive k of its row) are bank-conflict free; the k order inside a K-tile is
// permuted identically for A and W (lane half h of read c supplies k = 8c+4h+e at MFMA step e),
// which is legal because the contraction is a sum.
//
// Two A-addressing modes share the kernel: plain row-major (A + m*lda) and the implicit-GEMM
// gather of the stem's second 3x3/stride-2 convolution over a channel-last activation
// (row m = (b,t2,f2); K index = (kf,kt,ci)), see subsample.hip.
#include "cfm_common.h"

namespace {

constexpr int BK = 16, LDSR = BK + 4;  // K-tile depth; LDS row stride in floats (80 B, conflict-free b128 reads)

enum Epi { EPI_BIAS = 0, EPI_SWISH = 1, EPI_RELU = 2, EPI_GLU = 3, EPI_RESID = 4 };

struct GemmArgs {
    const float* A; const float* W; const float* bias; const float* R; float* C;
    int64_t M; int N; int K; int64_t lda, ldr, ldc; float alpha;
    int n_out;                      // GLU: output columns (N = 2*n_out)
    int cT2, cF2, cT1, cF1, cC;     // conv mode geometry
    unsigned tiles_m, tiles_n;
};

// Block tile BM x BN (each 64 or 128), always 4 waves as 2x2; a wave owns (BM/2) x (BN/2) = TM x TN MFMA tiles.
// ABL (diagnostics only, tools/gemm_tune.py): 0 = real kernel; 1 = no global->LDS refills after the prologue;
// 4 = global loads but no LDS writes; 5 = LDS writes but no global loads.  ABL > 0 computes garbage on purpose --
// it prices the loop's ingredients, nothing else.
template <int BM, int BN, int EPI, bool CONV, int ABL = 0>
__global__ __launch_bounds__(256, (BM == 128 && BN == 128) ? 3 : 4) void gemm_f32_kernel(const GemmArgs g) {
    constexpr int TM = BM / 64, TN = BN / 64;
    static_assert(EPI != EPI_GLU || TN == 2, "GLU keeps value and gate tiles in one wave");
    __shared__ __attribute__((aligned(16))) float lds[2 * (BM + BN) * LDSR];
    float* As = lds;                    // [2][BM][LDSR]
    float* Bs = lds + 2 * BM * LDSR;    // [2][BN][LDSR]

    const unsigned nwg = g.tiles_m * g.tiles_n;
    const unsigned tile = xcd_remap(blockIdx.x, nwg);
    const unsigned tm = tile / g.tiles_n, tn = tile % g.tiles_n;
    const int64_t m0 = (int64_t)tm * BM;
    const int n0 = (int)tn * (EPI == EPI_GLU ? BN / 2 : BN);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int li = lane & 31, hf = lane >> 5;

    // ---- staging assignment: thread -> (row, 16-byte chunk) for TM rows of A and TN rows of W
    const int chunk = tid & 3, srow = tid >> 2;
    const float* a_ptr[TM];
    const float* w_ptr[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        int64_t m = m0 + srow + 64 * i;
        if (m >= g.M) m = g.M - 1;                       // clamp: row is loaded but never stored
        if (CONV) {
            const int f2 = (int)(m % g.cF2);
            const int64_t bt = m / g.cF2;
            const int t2 = (int)(bt % g.cT2);
            const int64_t b = bt / g.cT2;
            a_ptr[i] = g.A + (((b * g.cT1 + 2 * t2) * g.cF1 + 2 * f2) * (int64_t)g.cC);
        } else {
            a_ptr[i] = g.A + m * g.lda;
        }
    }
#pragma unroll
    for (int i = 0; i < TN; ++i) {
        const int r = srow + 64 * i;                     // LDS row of the W tile
        int n;
        if (EPI == EPI_GLU) {                            // wave wc: n-tile 0 = value cols, n-tile 1 = gate cols
            const int j = r & 31, nt = (r >> 5) & 1, w = r >> 6;
            n = n0 + w * 32 + j;
            if (n >= g.n_out) n = g.n_out - 1;
            n += nt * g.n_out;
        } else {
            n = n0 + r;
            if (n >= g.N) n = g.N - 1;
        }
        w_ptr[i] = g.W + (int64_t)n * g.K;
    }

    const int nkt = (g.K + BK - 1) / BK;
    f32x4 ra0[TM], rb0[TN], ra1[TM], rb1[TN];          // two staging sets: tiles t+1 and t+2 in flight
    auto load_tile = [&](f32x4 (&ra)[TM], f32x4 (&rb)[TN], int kt) {
        const int k = kt * BK + chunk * 4;
        int64_t aoff = k;
        if (CONV) {
            const int kk = kt * BK;
            const int tap = kk / g.cC, ci = kk - tap * g.cC;
            const int kf = tap / 3, ktp = tap - 3 * kf;
            aoff = ((int64_t)ktp * g.cF1 + kf) * g.cC + ci + chunk * 4;
        }
        const bool ok = k < g.K;
        if (ABL == 6) {   // same bytes and instruction count, but a wave-instruction covers 8 rows x one full 128-B line
            const int64_t off = (int64_t)((kt * 32) % g.K) + (tid & 7) * 4;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                int64_t m = m0 + (tid >> 3) + 32 * i;
                if (m >= g.M) m = g.M - 1;
                ra[i] = *reinterpret_cast<const f32x4*>(g.A + m * g.lda + off);
            }
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                int n = n0 + (tid >> 3) + 32 * i;
                if (n >= g.N) n = g.N - 1;
                rb[i] = *reinterpret_cast<const f32x4*>(g.W + (int64_t)n * g.K + off);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
            ra[i] = ok ? *reinterpret_cast<const f32x4*>(a_ptr[i] + aoff) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < TN; ++i)
            rb[i] = ok ? *reinterpret_cast<const f32x4*>(w_ptr[i] + k) : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto store_tile = [&](const f32x4 (&ra)[TM], const f32x4 (&rb)[TN], int buf) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
            *reinterpret_cast<f32x4*>(As + (buf * BM + srow + 64 * i) * LDSR + chunk * 4) = ra[i];
#pragma unroll
        for (int i = 0; i < TN; ++i)
            *reinterpret_cast<f32x4*>(Bs + (buf * BN + srow + 64 * i) * LDSR + chunk * 4) = rb[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int a_row = wr * (BM / 2) + li, b_row = wc * (BN / 2) + li;
    f32x4 fa0[TM], fb0[TN], fa1[TM], fb1[TN];
    auto read_frags = [&](f32x4 (&fa)[TM], f32x4 (&fb)[TN], int buf, int c) {
#pragma unroll
        for (int t = 0; t < TM; ++t)
            fa[t] = *reinterpret_cast<const f32x4*>(As + (buf * BM + a_row + 32 * t) * LDSR + 8 * c + 4 * hf);
#pragma unroll
        for (int t = 0; t < TN; ++t)
            fb[t] = *reinterpret_cast<const f32x4*>(Bs + (buf * BN + b_row + 32 * t) * LDSR + 8 * c + 4 * hf);
    };
    auto mfma_half = [&](const f32x4 (&fa)[TM], const f32x4 (&fb)[TN]) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int mt = 0; mt < TM; ++mt)
#pragma unroll
                for (int nt = 0; nt < TN; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mt][e], fb[nt][e], acc[mt][nt], 0, 0, 0);
    };

    // Software pipeline (per wave): while the MFMAs of one half K-tile run, the operand fragments of the next half
    // are already in flight from LDS, tile t+1 is being written to the other LDS buffer and tiles t+2, t+3 are in
    // flight from L2/HBM in two alternating register sets (a load gets two full K-steps to land).
    // One barrier per K-tile, placed between the two MFMA groups.
    load_tile(ra0, rb0, 0);
    store_tile(ra0, rb0, 0);
    if (nkt > 1) load_tile(ra1, rb1, 1);
    if (nkt > 2) load_tile(ra0, rb0, 2);
    __syncthreads();
    read_frags(fa0, fb0, 0, 0);
    auto k_step = [&](int kt, f32x4 (&ra)[TM], f32x4 (&rb)[TN]) {   // (ra, rb) holds tile kt+1 on entry
        const int cur = kt & 1;
        const bool more = kt + 1 < nkt;
        read_frags(fa1, fb1, cur, 1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_half(fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        if (more && ABL != 1) {
            if (ABL != 4 && ABL != 6) store_tile(ra, rb, cur ^ 1);   // ABL 4/6: loads only (no LDS write)
            else {
#pragma unroll
                for (int i = 0; i < TM; ++i) asm volatile("" :: "v"(ra[i]));
#pragma unroll
                for (int i = 0; i < TN; ++i) asm volatile("" :: "v"(rb[i]));
            }
            if (kt + 3 < nkt && ABL != 5) load_tile(ra, rb, kt + 3);  // ABL 5: LDS writes only (no loads)
        }
        __syncthreads();
        if (more) read_frags(fa0, fb0, cur ^ 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_half(fa1, fb1);
        __builtin_amdgcn_sched_barrier(0);
    };
    int kt = 0;
    for (; kt + 1 < nkt; kt += 2) {
        k_step(kt, ra1, rb1);
        k_step(kt + 1, ra0, rb0);
    }
    if (kt < nkt) k_step(kt, ra1, rb1);

    // ---- epilogue.  C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    if (EPI == EPI_GLU) {
        const int col = n0 + wc * 32 + li;
        if (col < g.n_out) {
            const float bv = g.bias[col], bg = g.bias[g.n_out + col];
#pragma unroll
            for (int mt = 0; mt < TM; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t row = m0 + wr * (BM / 2) + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf;
                    if (row < g.M)
                        g.C[row * g.ldc + col] = (acc[mt][0][r] + bv) * sigmoidf_acc(acc[mt][TN - 1][r] + bg);
                }
        }
    } else {
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) {
            const int col = n0 + wc * (BN / 2) + nt * 32 + li;
            if (col >= g.N) continue;
            const float bb = g.bias[col];
#pragma unroll
            for (int mt = 0; mt < TM; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t row = m0 + wr * (BM / 2) + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf;
                    if (row >= g.M) continue;
                    float v = acc[mt][nt][r] + bb;
                    if (EPI == EPI_SWISH) v = swishf_acc(v);
                    if (EPI == EPI_RELU) v = fmaxf(v, 0.f);
                    if (EPI == EPI_RESID) v = g.alpha * v + g.R[row * g.ldr + col];
                    g.C[row * g.ldc + col] = v;
                }
        }
    }
}

// ---- tile-shape selection.  fp32 MFMA is so slow relative to LDS/L2 that the only thing that matters is keeping
// every SIMD's matrix pipe fed: >= 2-4 resident waves per SIMD and no nearly-empty trailing round of blocks.
// cfg: 0 = 128x128, 1 = 128x64, 2 = 64x128, 3 = 64x64.
constexpr int CFG_BM[4] = {128, 128, 64, 64};
constexpr int CFG_BN[4] = {128, 64, 128, 64};

// Measured on MI355X (tools/gemm_tune.py, profiles/r01_gemm_tune.txt): the 128x128 tile only wins when every CU gets
// a long queue of tiles (>= ~12 per CU: conv2, big square GEMMs) so that prologue/epilogue phases of different
// blocks overlap; the K=512 layer GEMMs (4 tiles per CU or fewer, all resident at once) run 7-12 % faster on
// 128x64, and the N=512 GEMMs (1 tile per CU at 128x128) 15-40 % faster on 64x64.
inline int choose_cfg(int64_t M, int ncols, bool glu) {
    const int bn = glu ? 64 : 128;
    const int64_t n128 = ((M + 127) / 128) * ((ncols + bn - 1) / bn);
    if (n128 >= 12 * 256) return 0;
    if (glu) return 2;
    if (n128 >= 3 * 256) return 1;
    return 3;
}

template <int BM, int BN, int EPI, bool CONV, int ABL = 0>
int launch_cfg(GemmArgs g, hipStream_t s) {
    const int ncols = EPI == EPI_GLU ? g.n_out : g.N;
    const int bn = EPI == EPI_GLU ? BN / 2 : BN;
    g.tiles_m = (unsigned)((g.M + BM - 1) / BM);
    g.tiles_n = (unsigned)((ncols + bn - 1) / bn);
    hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, EPI, CONV, ABL>), dim3(g.tiles_m * g.tiles_n), dim3(256), 0, s, g);
    return cfm_launch_status();
}

template <int EPI, bool CONV>
int launch(const GemmArgs& g, hipStream_t s, int force_cfg = -1) {
    const int ncols = EPI == EPI_GLU ? g.n_out : g.N;
    int cfg = force_cfg >= 0 ? force_cfg : choose_cfg(g.M, ncols, EPI == EPI_GLU);
    if constexpr (EPI == EPI_GLU) {
        if (cfg == 1) cfg = 0;
        if (cfg == 3) cfg = 2;
        return cfg == 0 ? launch_cfg<128, 128, EPI, CONV>(g, s) : launch_cfg<64, 128, EPI, CONV>(g, s);
    } else {
        switch (cfg) {
            case 0: return launch_cfg<128, 128, EPI, CONV>(g, s);
            case 1: return launch_cfg<128, 64, EPI, CONV>(g, s);
            case 2: return launch_cfg<64, 128, EPI, CONV>(g, s);
            default: return launch_cfg<64, 64, EPI, CONV>(g, s);
        }
    }
}

int check(const GemmArgs& g) {
    CFM_REQUIRE(g.A && g.W && g.bias && g.C, CFM_ERR_NULL);
    CFM_REQUIRE(g.M > 0 && g.N > 0 && g.K > 0 && (g.K & 3) == 0, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE((g.lda & 3) == 0 && g.lda >= g.K, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE((g.M + 63) / 64 * (int64_t)((g.N + 31) / 32) < ((int64_t)1 << 31), CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(CFM_ALIGNED16(g.A) && CFM_ALIGNED16(g.W), CFM_ERR_ALIGN);
    return CFM_OK;
}

}  // namespace

#define GEMM_ARGS_PLAIN(nn) GemmArgs g{}; g.A = A; g.W = W; g.bias = bias; g.C = C; g.M = M; g.N = (nn); g.K = K; \
    g.lda = lda; g.ldc = ldc; g.alpha = 1.f

extern "C" int cfm_gemm_bias_f32(const float* A, const float* W, const float* bias, float* C, int64_t M, int N,
                                 int K, int64_t lda, int64_t ldc, cfm_stream_t stream) {
    GEMM_ARGS_PLAIN(N);
    int st = check(g); if (st) return st;
    CFM_REQUIRE(ldc >= N, CFM_ERR_BAD_SHAPE);
    return launch<EPI_BIAS, false>(g, static_cast<hipStream_t>(stream));
}

extern "C" int cfm_gemm_bias_swish_f32(const float* A, const float* W, const float* bias, float* C, int64_t M,
                                       int N, int K, int64_t lda, int64_t ldc, cfm_stream_t stream) {
    GEMM_ARGS_PLAIN(N);
    int st = check(g); if (st) return st;
    CFM_REQUIRE(ldc >= N, CFM_ERR_BAD_SHAPE);
    return launch<EPI_SWISH, false>(g, static_cast<hipStream_t>(stream));
}

extern "C" int cfm_gemm_bias_relu_f32(const float* A, const float* W, const float* bias, float* C, int64_t M,
                                      int N, int K, int64_t lda, int64_t ldc, cfm_stream_t stream) {
    GEMM_ARGS_PLAIN(N);
    int st = check(g); if (st) return st;
    CFM_REQUIRE(ldc >= N, CFM_ERR_BAD_SHAPE);
    return launch<EPI_RELU, false>(g, static_cast<hipStream_t>(stream));
}

extern "C" int cfm_gemm_bias_glu_f32(const float* A, const float* W, const float* bias, float* C, int64_t M,
                                     int n_out, int K, int64_t lda, int64_t ldc, cfm_stream_t stream) {
    GEMM_ARGS_PLAIN(2 * n_out);
    g.n_out = n_out;
    int st = check(g); if (st) return st;
    CFM_REQUIRE(n_out > 0 && ldc >= n_out, CFM_ERR_BAD_SHAPE);
    return launch<EPI_GLU, false>(g, static_cast<hipStream_t>(stream));
}

extern "C" int cfm_gemm_bias_residual_f32(const float* A, const float* W, const float* bias, const float* R,
                                          float alpha, float* C, int64_t M, int N, int K, int64_t lda,
                                          int64_t ldr, int64_t ldc, cfm_stream_t stream) {
    GEMM_ARGS_PLAIN(N);
    g.R = R; g.ldr = ldr; g.alpha = alpha;
    int st = check(g); if (st) return st;
    CFM_REQUIRE(R != nullptr, CFM_ERR_NULL);
    CFM_REQUIRE(ldc >= N && ldr >= N, CFM_ERR_BAD_SHAPE);
    return launch<EPI_RESID, false>(g, static_cast<hipStream_t>(stream));
}

// Implicit-GEMM second stem convolution (3x3, stride 2, channel-last input, packed weight).  Declared in
// subsample.hip's section of the ABI; lives here to share the kernel template.
extern "C" int cfm_subsample_conv2_relu_f32(const float* h1, const float* w2p, const float* b2, float* h2, int B,
                                            int F1, int T1, int C, cfm_stream_t stream) {
    CFM_REQUIRE(h1 && w2p && b2 && h2, CFM_ERR_NULL);
    CFM_REQUIRE(B > 0 && F1 >= 3 && T1 >= 3 && C > 0, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(C % BK == 0, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(CFM_ALIGNED16(h1) && CFM_ALIGNED16(w2p) && CFM_ALIGNED16(h2), CFM_ERR_ALIGN);
    GemmArgs g{};
    g.cT1 = T1; g.cF1 = F1; g.cC = C; g.cT2 = (T1 - 1) / 2; g.cF2 = (F1 - 1) / 2;
    g.A = h1; g.W = w2p; g.bias = b2; g.C = h2;
    g.M = (int64_t)B * g.cT2 * g.cF2; g.N = C; g.K = 9 * C; g.lda = 0; g.ldc = C; g.alpha = 1.f;
    return launch<EPI_RELU, true>(g, static_cast<hipStream_t>(stream));
}

// Tuning / diagnostics: run the residual-epilogue GEMM with a forced tile shape (cfg 0..3 = 128x128, 128x64,
// 64x128, 64x64; -1 = the heuristic).  Used by tools/gemm_tune.py; results are identical for every cfg.
extern "C" int cfm_debug_gemm_cfg_f32(int cfg, const float* A, const float* W, const float* bias, const float* R,
                                      float alpha, float* C, int64_t M, int N, int K, cfm_stream_t stream) {
    GemmArgs g{}; g.A = A; g.W = W; g.bias = bias; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = K; g.ldc = N;
    g.R = R; g.ldr = N; g.alpha = alpha;
    int st = check(g); if (st) return st;
    CFM_REQUIRE(R != nullptr, CFM_ERR_NULL);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (cfg >= 16) {                                   // ablation builds: cfg = 16*ABL + tile (0 or 3)
        const int abl = cfg >> 4, tile = cfg & 15;
        if (tile == 0) {
            if (abl == 1) return launch_cfg<128, 128, EPI_RESID, false, 1>(g, s);
            if (abl == 4) return launch_cfg<128, 128, EPI_RESID, false, 4>(g, s);
            if (abl == 5) return launch_cfg<128, 128, EPI_RESID, false, 5>(g, s);
            if (abl == 6) return launch_cfg<128, 128, EPI_RESID, false, 6>(g, s);
        } else if (tile == 3) {
            if (abl == 1) return launch_cfg<64, 64, EPI_RESID, false, 1>(g, s);
            if (abl == 4) return launch_cfg<64, 64, EPI_RESID, false, 4>(g, s);
            if (abl == 5) return launch_cfg<64, 64, EPI_RESID, false, 5>(g, s);
            if (abl == 6) return launch_cfg<64, 64, EPI_RESID, false, 6>(g, s);
        }
        return CFM_ERR_BAD_SHAPE;
    }
    CFM_REQUIRE(cfg >= -1 && cfg <= 3, CFM_ERR_BAD_SHAPE);
    return launch<EPI_RESID, false>(g, s, cfg);
}
