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

constexpr int BM = 128, BN = 128, BK = 16, LDSR = BK + 4;  // LDS row stride in floats

enum Epi { EPI_BIAS = 0, EPI_SWISH = 1, EPI_RELU = 2, EPI_GLU = 3, EPI_RESID = 4 };

struct GemmArgs {
    const float* A; const float* W; const float* bias; const float* R; float* C;
    int64_t M; int N; int K; int64_t lda, ldr, ldc; float alpha;
    int n_out;                      // GLU: output columns (N = 2*n_out)
    int cT2, cF2, cT1, cF1, cC;     // conv mode geometry
    unsigned tiles_m, tiles_n;
};

template <int EPI, bool CONV>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 2 * BM * LDSR];
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

    // ---- staging assignment: thread -> (row, 16-byte chunk) for two rows of A and two rows of W
    const int chunk = tid & 3, srow = tid >> 2;
    const float* a_ptr[2];
    const float* w_ptr[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
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
    f32x4 ra[2], rb[2];
    auto load_tile = [&](int kt) {
        const int k = kt * BK + chunk * 4;
        int64_t aoff = k;
        if (CONV) {
            const int kk = kt * BK;
            const int tap = kk / g.cC, ci = kk - tap * g.cC;
            const int kf = tap / 3, ktp = tap - 3 * kf;
            aoff = ((int64_t)ktp * g.cF1 + kf) * g.cC + ci + chunk * 4;
        }
        const bool ok = k < g.K;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            ra[i] = ok ? *reinterpret_cast<const f32x4*>(a_ptr[i] + aoff) : f32x4{0.f, 0.f, 0.f, 0.f};
            rb[i] = ok ? *reinterpret_cast<const f32x4*>(w_ptr[i] + k) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = srow + 64 * i;
            *reinterpret_cast<f32x4*>(As + (buf * BM + r) * LDSR + chunk * 4) = ra[i];
            *reinterpret_cast<f32x4*>(Bs + (buf * BN + r) * LDSR + chunk * 4) = rb[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    load_tile(0);
    store_tile(0);
    __syncthreads();

    const int a_row = wr * 64 + li, b_row = wc * 64 + li;
    int cur = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        const bool more = kt + 1 < nkt;
        if (more) load_tile(kt + 1);
        f32x4 a[2][2], b[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                a[t][c] = *reinterpret_cast<const f32x4*>(As + (cur * BM + a_row + 32 * t) * LDSR + 8 * c + 4 * hf);
                b[t][c] = *reinterpret_cast<const f32x4*>(Bs + (cur * BN + b_row + 32 * t) * LDSR + 8 * c + 4 * hf);
            }
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][c][e], b[nt][c][e], acc[mt][nt], 0, 0, 0);
        if (more) store_tile(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue.  C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    if (EPI == EPI_GLU) {
        const int col = n0 + wc * 32 + li;
        if (col < g.n_out) {
            const float bv = g.bias[col], bg = g.bias[g.n_out + col];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t row = m0 + wr * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf;
                    if (row < g.M)
                        g.C[row * g.ldc + col] = (acc[mt][0][r] + bv) * sigmoidf_acc(acc[mt][1][r] + bg);
                }
        }
    } else {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int col = n0 + wc * 64 + nt * 32 + li;
            if (col >= g.N) continue;
            const float bb = g.bias[col];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t row = m0 + wr * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf;
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

template <int EPI, bool CONV>
int launch(GemmArgs g, hipStream_t s) {
    const int ncols = EPI == EPI_GLU ? g.n_out : g.N;
    const int bn = EPI == EPI_GLU ? BN / 2 : BN;
    g.tiles_m = (unsigned)((g.M + BM - 1) / BM);
    g.tiles_n = (unsigned)((ncols + bn - 1) / bn);
    hipLaunchKernelGGL((gemm_f32_kernel<EPI, CONV>), dim3(g.tiles_m * g.tiles_n), dim3(256), 0, s, g);
    return cfm_launch_status();
}

int check(const GemmArgs& g) {
    CFM_REQUIRE(g.A && g.W && g.bias && g.C, CFM_ERR_NULL);
    CFM_REQUIRE(g.M > 0 && g.N > 0 && g.K > 0 && (g.K & 3) == 0, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE((g.lda & 3) == 0 && g.lda >= g.K, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE((g.M + BM - 1) / BM * (int64_t)((g.N + BN / 2 - 1) / (BN / 2)) < ((int64_t)1 << 31), CFM_ERR_BAD_SHAPE);
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
