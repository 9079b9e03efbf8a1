// General fp32 MFMA GEMM for the backward pass:  C[i][j] (+)= alpha * sum_k A(i,k) * B(j,k)  [* swish'(Z[i][j])]
//
// Each operand is either ROW ("index-major": X[idx*ld + k], the forward layout) or COL ("contraction-major":
// X[k*ld + idx]).  That covers the three products of a Linear layer without ever transposing a tensor in HBM:
//     dX = dY . W          A = dY ROW (contraction = N),  B = W  COL            (I = M, J = K)
//     dW = dY^T . X        A = dY COL (contraction = M),  B = X  COL            (I = N, J = K)
//     (forward y = X . W^T A = X ROW, B = W ROW -- gemm_f32.hip)
// Same machinery as the forward kernel: v_mfma_f32_32x32x2_f32, 4 waves (2x2) on a BM x BN tile, K-step 16, LDS double
// buffer, fragments of the next k-slice in flight during the MFMAs, global loads two K-steps ahead, accumulators held
// transposed so the epilogue moves 16 bytes per lane.  COL operands are staged k-major in LDS ([16][BM+4] floats:
// 16-byte coalesced global loads and ds_write_b128 along the index dimension) and their fragments are read with four
// conflict-free ds_read_b32 per 8-deep k-slice instead of one ds_read_b128.
// Weight-gradient products have a short output (N x K) and a long contraction (M = B*T'): they are split along the
// contraction over gridDim.y slices that accumulate with fp32 atomics into a zero-initialised C (caller zeroes it).
#include "gemm_bwd_args.h"

namespace {

template <int BT, bool ROW>
struct OperandTile {                                    // LDS image of one operand's BT x 16 tile, one stage
    static constexpr int FLOATS = ROW ? BT * 20 : 16 * (BT + 4);
};

// GATHER (stem backward, implicit GEMMs -- nothing is im2col'ed in memory):
//   1: B operand = im2col(h1) read contraction-major: B(kidx, m) = h1[row(m) + tap(kidx)]   (dW2 = dz2^T . im2col(h1))
//   2: A operand = rows of dz2 selected per (class row, tap), zero when the tap falls outside h2; C rows are scattered
//      to the class's positions of dh1                                             (dh1 = conv-transpose(dz2, W2))
// SPLITK: the contraction is split over gridDim.y and partial tiles are summed with fp32 atomics.  Atomics only run at
// full rate when one wave instruction covers whole 128-byte row segments (MI355X_MICROARCH.md, global float atomics:
// 64 lanes in 64 different rows are ~17x slower), so this variant keeps the accumulators in the NATURAL MFMA
// orientation (lane = output column) instead of the transposed one used for 16-byte stores.
template <int BM, int BN, bool AROW, bool BROW, int EPI, int GATHER = 0, bool SPLITK = false>
__global__ __launch_bounds__(256, (BM == 128 && BN == 128) ? 3 : 4) void gemm_bwd_kernel(const BwdArgs g) {
    constexpr int TM = BM / 64, TN = BN / 64, BK = 16;
    constexpr int AF = OperandTile<BM, AROW>::FLOATS, BF = OperandTile<BN, BROW>::FLOATS;
    __shared__ __attribute__((aligned(16))) float lds[2 * (AF + BF)];
    float* As = lds;                 // [2][AF]
    float* Bs = lds + 2 * AF;        // [2][BF]

    const unsigned nwg = g.tiles_i * g.tiles_j;
    const unsigned tile = xcd_remap(blockIdx.x, nwg);
    const unsigned ti = tile / g.tiles_j, tj = tile % g.tiles_j;
    const int i0 = (int)ti * BM, j0 = (int)tj * BN;
    const int64_t kbeg = (int64_t)blockIdx.y * g.k_per_split;
    const int64_t kend = min(g.Kc, kbeg + g.k_per_split);
    if (kbeg >= kend) return;
    const int zb0 = blockIdx.z / g.nb1, zb1 = blockIdx.z % g.nb1;
    const float* Ab = g.A + zb0 * g.sa0 + zb1 * g.sa1;
    const float* Bb = g.B + zb0 * g.sb0 + zb1 * g.sb1;
    float* Cb = g.C + zb0 * g.sc0 + zb1 * g.sc1;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int li = lane & 31, hf = lane >> 5;

    // ---- staging.  ROW: thread -> (row tid>>2 (+64 p), k-chunk tid&3).  COL: thread -> (k row, 16-byte index chunk).
    f32x4 ra0[TM], rb0[TN], ra1[TM], rb1[TN];
    // GATHER 2: the TM class rows this thread stages, decoded once: (b, a, c) -> dz2 row base of the (dt,df)=(0,0) tap
    int g2_b[TM], g2_a[TM], g2_c[TM];
    if (GATHER == 2) {
#pragma unroll
        for (int p = 0; p < TM; ++p) {
            int idx = i0 + (tid >> 2) + 64 * p;
            if (idx >= g.I) idx = g.I - 1;
            const int per = g.pA * g.pC;
            g2_b[p] = idx / per;
            const int r = idx - g2_b[p] * per;
            g2_a[p] = r / g.pC;
            g2_c[p] = r - g2_a[p] * g.pC;
        }
    }
    auto load_operand = [&](auto& regs, const float* X, int64_t ld, int idx0, int IDX, bool row, int bt, int64_t k0,
                            int gather) {
        constexpr int NV = sizeof(regs) / sizeof(f32x4);
#pragma unroll
        for (int p = 0; p < NV; ++p) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (gather == 2) {                                         // A rows of dz2 per (class row, tap); C % 16 == 0
                const int tap = (int)(k0 / g.cC);
                const int co = (int)(k0 - (int64_t)tap * g.cC) + (tid & 3) * 4;
                const int t2 = g2_a[p] + g.tap_dt[tap], f2 = g2_c[p] + g.tap_df[tap];
                if (t2 >= 0 && t2 < g.cT2 && f2 >= 0 && f2 < g.cF2)
                    v = *reinterpret_cast<const f32x4*>(X + (((int64_t)g2_b[p] * g.cT2 + t2) * g.cF2 + f2) * g.cC + co);
            } else if (gather == 1) {                                  // B = im2col(h1), contraction-major
                const int cpr = bt >> 2;
                const int slot = tid + 256 * p;
                const int kk = slot / cpr, ch = slot - kk * cpr;
                const int64_t m = k0 + kk;
                const int idx = idx0 + 4 * ch;                         // K index (tap, ci); a chunk never straddles taps
                if (m < kend && idx < IDX) {
                    const int f2 = (int)(m % g.cF2);
                    const int64_t bt2 = m / g.cF2;
                    const int t2 = (int)(bt2 % g.cT2);
                    const int64_t b = bt2 / g.cT2;
                    const int tap = idx / g.cC, ci = idx - tap * g.cC;
                    const int kf = tap / 3, ktp = tap - 3 * kf;
                    v = *reinterpret_cast<const f32x4*>(
                        X + (((b * g.cT1 + 2 * t2 + ktp) * g.cF1 + 2 * f2 + kf) * (int64_t)g.cC) + ci);
                }
            } else if (row) {
                int idx = idx0 + (tid >> 2) + 64 * p;
                const int64_t k = k0 + (tid & 3) * 4;
                if (idx >= IDX) idx = IDX - 1;
                if (k + 3 < kend) v = *reinterpret_cast<const f32x4*>(X + (int64_t)idx * ld + k);
                else if (k < kend) {                                   // ragged end of the contraction
                    const float* s = X + (int64_t)idx * ld + k;
                    v.x = s[0];
                    if (k + 1 < kend) v.y = s[1];
                    if (k + 2 < kend) v.z = s[2];
                }
            } else {
                const int cpr = bt >> 2;                               // 16-byte chunks per k row (32 or 16)
                const int slot = tid + 256 * p;
                const int kk = slot / cpr, ch = slot - kk * cpr;
                const int64_t k = k0 + kk;
                const int idx = idx0 + 4 * ch;
                if (k < kend) {
                    const float* s = X + k * ld + idx;
                    if (idx + 3 < IDX) v = *reinterpret_cast<const f32x4*>(s);
                    else {                                             // ragged index edge: zero-filled
                        if (idx < IDX) v.x = s[0];
                        if (idx + 1 < IDX) v.y = s[1];
                        if (idx + 2 < IDX) v.z = s[2];
                    }
                }
            }
            regs[p] = v;
        }
    };
    auto store_operand = [&](const auto& regs, float* S, bool row, int bt) {
        constexpr int NV = sizeof(regs) / sizeof(f32x4);
#pragma unroll
        for (int p = 0; p < NV; ++p) {
            if (row) {
                *reinterpret_cast<f32x4*>(S + ((tid >> 2) + 64 * p) * 20 + (tid & 3) * 4) = regs[p];
            } else {
                const int cpr = bt >> 2;
                const int slot = tid + 256 * p;
                const int kk = slot / cpr, ch = slot - kk * cpr;
                *reinterpret_cast<f32x4*>(S + kk * (bt + 4) + 4 * ch) = regs[p];
            }
        }
    };
    auto load_tile = [&](f32x4 (&ra)[TM], f32x4 (&rb)[TN], int kt) {
        const int64_t k0 = kbeg + (int64_t)kt * BK;
        load_operand(ra, Ab, g.lda, i0, g.I, AROW, BM, k0, GATHER == 2 ? 2 : 0);
        load_operand(rb, Bb, g.ldb, j0, g.J, BROW, BN, k0, GATHER == 1 ? 1 : 0);
    };
    auto store_tile = [&](const f32x4 (&ra)[TM], const f32x4 (&rb)[TN], int buf) {
        store_operand(ra, As + buf * AF, AROW, BM);
        store_operand(rb, Bs + buf * BF, BROW, BN);
    };

    // ---- fragments: lane (index li within a 32-wide MFMA tile, k-half hf) needs k = 8c + 4hf + {0..3}
    const int a_idx = wr * (BM / 2) + li, b_idx = wc * (BN / 2) + li;
    f32x4 fa0[TM], fb0[TN], fa1[TM], fb1[TN];
    auto read_one = [&](const float* S, bool row, int bt, int idx, int c) -> f32x4 {
        if (row) return *reinterpret_cast<const f32x4*>(S + idx * 20 + 8 * c + 4 * hf);
        const float* s = S + (8 * c + 4 * hf) * (bt + 4) + idx;
        return f32x4{s[0], s[bt + 4], s[2 * (bt + 4)], s[3 * (bt + 4)]};
    };
    auto read_frags = [&](f32x4 (&fa)[TM], f32x4 (&fb)[TN], int buf, int c) {
#pragma unroll
        for (int t = 0; t < TM; ++t) fa[t] = read_one(As + buf * AF, AROW, BM, a_idx + 32 * t, c);
#pragma unroll
        for (int t = 0; t < TN; ++t) fb[t] = read_one(Bs + buf * BF, BROW, BN, b_idx + 32 * t, c);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#define BWD_MFMA_SLICE(FA, FB)                                                                              \
    _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                           \
    _Pragma("unroll") for (int mt = 0; mt < TM; ++mt)                                                       \
    _Pragma("unroll") for (int nt = 0; nt < TN; ++nt)                                                       \
        acc[mt][nt] = SPLITK ? __builtin_amdgcn_mfma_f32_32x32x2f32(FA[mt][e], FB[nt][e], acc[mt][nt], 0, 0, 0)    \
                             : __builtin_amdgcn_mfma_f32_32x32x2f32(FB[nt][e], FA[mt][e], acc[mt][nt], 0, 0, 0)

    const int nkt = (int)((kend - kbeg + BK - 1) / BK);
    load_tile(ra0, rb0, 0);
    store_tile(ra0, rb0, 0);
    if (nkt > 1) load_tile(ra1, rb1, 1);
    if (nkt > 2) load_tile(ra0, rb0, 2);
    __syncthreads();
    read_frags(fa0, fb0, 0, 0);
    auto k_step = [&](int kt, f32x4 (&ra)[TM], f32x4 (&rb)[TN]) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nkt;
        read_frags(fa1, fb1, cur, 1);
        __builtin_amdgcn_sched_barrier(0);
        BWD_MFMA_SLICE(fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
            store_tile(ra, rb, cur ^ 1);
            if (kt + 3 < nkt) load_tile(ra, rb, kt + 3);
        }
        __syncthreads();
        if (more) read_frags(fa0, fb0, cur ^ 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        BWD_MFMA_SLICE(fa1, fb1);
        __builtin_amdgcn_sched_barrier(0);
    };
    int kt = 0;
    for (; kt + 1 < nkt; kt += 2) {
        k_step(kt, ra1, rb1);
        k_step(kt + 1, ra0, rb0);
    }
    if (kt < nkt) k_step(kt, ra1, rb1);
#undef BWD_MFMA_SLICE

    if (SPLITK) bwd_epilogue_atomic<BM, BN, TM, TN>(g, Cb, acc, i0, j0, wr, wc, li, hf);
    else bwd_epilogue_rows<BM, BN, EPI, GATHER, TM, TN>(g, Cb, acc, i0, j0, wr, wc, li, hf);
}

template <int BM, int BN, bool AROW, bool BROW, int EPI, int GATHER = 0>
int launch_one(BwdArgs g, hipStream_t s) {
    constexpr bool kCanSplit = EPI == BEPI_SCALE && GATHER != 2;
    g.tiles_i = (unsigned)((g.I + BM - 1) / BM);
    g.tiles_j = (unsigned)((g.J + BN - 1) / BN);
    // split the contraction when the output alone cannot fill the chip (weight gradients)
    const unsigned tiles = g.tiles_i * g.tiles_j;
    int splits = 1;
    if (g.splits == 0 && kCanSplit) {                  // 0 = auto, 1 = forbid
        while ((int64_t)tiles * g.nbatch * splits < 768 && g.Kc / (splits * 2) >= 512 && splits < 32) splits *= 2;
    }
    g.splits = splits;
    const int64_t per = (g.Kc + splits - 1) / splits;
    g.k_per_split = (per + 15) / 16 * 16;
    const dim3 grid(tiles, (unsigned)splits, (unsigned)g.nbatch);
    if constexpr (kCanSplit) {
        if (splits > 1) {
            hipLaunchKernelGGL((gemm_bwd_kernel<BM, BN, AROW, BROW, EPI, GATHER, true>), grid, dim3(256), 0, s, g);
            return cfm_launch_status();
        }
    }
    hipLaunchKernelGGL((gemm_bwd_kernel<BM, BN, AROW, BROW, EPI, GATHER, false>), grid, dim3(256), 0, s, g);
    return cfm_launch_status();
}

int g_debug_tile = -1;   // diagnostics only (tools/gemm_tune.py bwd): -1 = heuristic, 0 = 128x128, 1 = 128x64, 3 = 64x64

template <bool AROW, bool BROW, int EPI>
int launch_layout(const BwdArgs& g, hipStream_t s) {
    const int64_t t128 = (int64_t)((g.I + 127) / 128) * ((g.J + 127) / 128) * g.nbatch;
    int tile = g_debug_tile;
    if (tile < 0) {
        if (t128 >= 12 * 256 && g.I >= 96 && g.J >= 96) tile = 0;
        else if (t128 >= 3 * 256 && g.I >= 96 && g.J >= 48) tile = 1;
        else tile = 3;
    }
    if (tile == 0) return launch_one<128, 128, AROW, BROW, EPI>(g, s);
    if (tile == 1) return launch_one<128, 64, AROW, BROW, EPI>(g, s);
    return launch_one<64, 64, AROW, BROW, EPI>(g, s);
}

}  // namespace

int cfm_bwd_debug_tile() { return g_debug_tile; }

// C (I x J, leading dim ldc) (+)= alpha * op(A) . op(B)^T, optionally times swish'(Z).  a_col / b_col select the
// contraction-major layout of each operand (see the header of this file).  Requirements: lda, ldb, ldc multiples of
// 4; A, B, C, Z 16-byte aligned (also per batch: strides multiples of 4).  `allow_split` != 0 lets the contraction be
// split with fp32 atomics: C MUST then be zero-filled by the caller and the summation order is not deterministic.
// `accumulate` != 0: C += (plain read-modify-write; splitting is disabled).  Batched form: nbatch = nb0*nb1 problems,
// problem (b0,b1) uses A + b0*sa0 + b1*sa1 etc. (pass nbatch = nb1 = 1 and zero strides for a single GEMM).
// drop_p/drop_seed (with Z): the forward applied dropout to swish(Z) -- the same mask is applied to the result.
extern "C" int cfm_gemm_bwd_batched_f32(const float* A, int a_col, int64_t lda, const float* B, int b_col, int64_t ldb,
                                        const float* Z_or_null, int64_t ldz, float alpha, float* C, int64_t ldc,
                                        int I, int J, int64_t Kc, int allow_split, int accumulate, int nbatch, int nb1,
                                        int64_t sa0, int64_t sa1, int64_t sb0, int64_t sb1, int64_t sc0, int64_t sc1,
                                        float drop_p, uint64_t drop_seed, cfm_stream_t stream) {
    CFM_REQUIRE(A && B && C, CFM_ERR_NULL);
    CFM_REQUIRE(I > 0 && J > 0 && Kc > 0 && nbatch > 0 && nb1 > 0 && nbatch % nb1 == 0 && nbatch <= 65535, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE((ldc & 3) == 0 && (lda & 3) == 0 && (ldb & 3) == 0, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(((sa0 | sa1 | sb0 | sb1 | sc0 | sc1) & 3) == 0, CFM_ERR_BAD_SHAPE);
    // (lda < Kc is legal for an index-major operand: overlapping rows, e.g. STFT frames with lda = hop)
    CFM_REQUIRE(CFM_ALIGNED16(A) && CFM_ALIGNED16(B) && CFM_ALIGNED16(C), CFM_ERR_ALIGN);
    CFM_REQUIRE(!Z_or_null || (CFM_ALIGNED16(Z_or_null) && (ldz & 3) == 0), CFM_ERR_ALIGN);
    BwdArgs g{};
    g.A = A; g.B = B; g.Z = Z_or_null; g.C = C; g.I = I; g.J = J; g.Kc = Kc;
    g.lda = lda; g.ldb = ldb; g.ldz = ldz; g.ldc = ldc; g.alpha = alpha;
    g.splits = (allow_split && !accumulate) ? 0 : 1;
    g.accumulate = accumulate; g.nbatch = nbatch; g.nb1 = nb1; g.drop_p = drop_p; g.drop_seed = drop_seed;
    g.sa0 = sa0; g.sa1 = sa1; g.sb0 = sb0; g.sb1 = sb1; g.sc0 = sc0; g.sc1 = sc1;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (Z_or_null) {
        CFM_REQUIRE(!a_col && b_col && nbatch == 1, CFM_ERR_UNSUPPORTED);   // only dX = dY.W carries the swish' epilogue
        g.splits = 1;
        return launch_layout<true, false, BEPI_DSWISH>(g, s);
    }
    if (!a_col && !b_col) return launch_layout<true, true, BEPI_SCALE>(g, s);
    if (!a_col && b_col) return launch_layout<true, false, BEPI_SCALE>(g, s);
    if (a_col && b_col) return launch_layout<false, false, BEPI_SCALE>(g, s);
    return launch_layout<false, true, BEPI_SCALE>(g, s);
}

// ---- backward of the stem's second convolution (convolution.py:46, 3x3 stride 2, channel-last) ------------------------
// dw2p (C, 9C) [packed (co, kf, kt, ci) layout of cfm_pack_conv2_weight_f32] += dz2^T . im2col(h1); caller zero-fills.
extern "C" int cfm_subsample_conv2_bwd_weight_f32(const float* dz2, const float* h1, float* dw2p, int B, int F1, int T1,
                                                  int C, cfm_stream_t stream) {
    CFM_REQUIRE(dz2 && h1 && dw2p, CFM_ERR_NULL);
    CFM_REQUIRE(B > 0 && F1 >= 3 && T1 >= 3 && C > 0 && (C & 3) == 0, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(CFM_ALIGNED16(dz2) && CFM_ALIGNED16(h1) && CFM_ALIGNED16(dw2p), CFM_ERR_ALIGN);
    BwdArgs g{};
    g.cT1 = T1; g.cF1 = F1; g.cT2 = (T1 - 1) / 2; g.cF2 = (F1 - 1) / 2; g.cC = C;
    g.A = dz2; g.B = h1; g.C = dw2p; g.I = C; g.J = 9 * C; g.Kc = (int64_t)B * g.cT2 * g.cF2;
    g.lda = C; g.ldb = 0; g.ldc = 9 * C; g.alpha = 1.f; g.splits = 0; g.nbatch = 1; g.nb1 = 1;
    return launch_one<128, 128, false, false, BEPI_SCALE, 1>(g, static_cast<hipStream_t>(stream));
}

// dh1 (B,T1,F1,C) = conv-transpose(dz2 (B,T2,F2,C), w2): four parity classes of (t1,f1), each a regular implicit GEMM
// over the 1, 2, 2 or 4 taps that reach it.  w2c: class-packed weights from cfm_pack_conv2_weight_t_f32: for class
// q = 2*pt+pf a (C_in, ntap_q*C_out) block at offset C*C*{0,4,6,8}[q] floats (9*C*C floats in total).
extern "C" int cfm_subsample_conv2_bwd_input_f32(const float* dz2, const float* w2c, float* dh1, int B, int F1, int T1,
                                                 int C, cfm_stream_t stream) {
    CFM_REQUIRE(dz2 && w2c && dh1, CFM_ERR_NULL);
    CFM_REQUIRE(B > 0 && F1 >= 3 && T1 >= 3 && C > 0 && (C % 16) == 0, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(CFM_ALIGNED16(dz2) && CFM_ALIGNED16(w2c) && CFM_ALIGNED16(dh1), CFM_ERR_ALIGN);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int T2 = (T1 - 1) / 2, F2 = (F1 - 1) / 2;
    const int64_t woff[4] = {0, 4, 6, 8};
    for (int pt = 0; pt < 2; ++pt)
        for (int pf = 0; pf < 2; ++pf) {
            BwdArgs g{};
            g.cT1 = T1; g.cF1 = F1; g.cT2 = T2; g.cF2 = F2; g.cC = C; g.pt = pt; g.pf = pf;
            g.pA = (T1 - pt + 1) / 2; g.pC = (F1 - pf + 1) / 2;
            if (g.pA <= 0 || g.pC <= 0) continue;
            int nt = 0;                                   // taps (kt,kf) with kt = pt (mod 2), kf = pf (mod 2); t2 = a - (kt-pt)/2
            for (int kt = pt; kt < 3; kt += 2)
                for (int kf = pf; kf < 3; kf += 2) { g.tap_dt[nt] = -(kt - pt) / 2; g.tap_df[nt] = -(kf - pf) / 2; ++nt; }
            const int q = 2 * pt + pf;
            g.A = dz2; g.B = w2c + woff[q] * C * C; g.C = dh1;
            g.I = B * g.pA * g.pC; g.J = C; g.Kc = (int64_t)nt * C;
            g.lda = 0; g.ldb = (int64_t)nt * C; g.ldc = C; g.alpha = 1.f; g.splits = 1; g.nbatch = 1; g.nb1 = 1;
            int st = launch_one<128, 128, true, true, BEPI_SCALE, 2>(g, s);
            if (st) return st;
        }
    return CFM_OK;
}

// diagnostics only: force the block tile of every later cfm_gemm_bwd* call in this process (-1 restores the heuristic)
extern "C" int cfm_debug_set_bwd_tile(int tile) { g_debug_tile = tile; return CFM_OK; }

extern "C" int cfm_gemm_bwd_f32(const float* A, int a_col, int64_t lda, const float* B, int b_col, int64_t ldb,
                                const float* Z_or_null, int64_t ldz, float alpha, float* C, int64_t ldc,
                                int I, int J, int64_t Kc, int allow_split, cfm_stream_t stream) {
    return cfm_gemm_bwd_batched_f32(A, a_col, lda, B, b_col, ldb, Z_or_null, ldz, alpha, C, ldc, I, J, Kc, allow_split,
                                    0, 1, 1, 0, 0, 0, 0, 0, 0, 0.f, 0, stream);
}
