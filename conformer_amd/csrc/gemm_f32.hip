// fp32 GEMM family on the CDNA4 matrix pipe: C = epilogue(A . W^T + bias).
//
// v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain, 64 FLOP/clk/SIMD = 157 TFLOP/s chip peak).  A workgroup is 4 waves
// (2x2) on a BM x BN block tile (each 64 or 128); a wave owns (BM/2) x (BN/2) = TM x TN MFMA tiles of 32x32, i.e.
// up to four independent accumulator chains.  The k order inside a K-tile is permuted identically for A and W
// (lane half h of fragment read c supplies k = 8c+4h+e at MFMA step e) -- legal because the contraction is a sum --
// so ONE 16-byte LDS read feeds FOUR MFMA steps.  Per-wave software pipeline: the fragments of the next 8-deep
// k-slice are in flight from LDS while the MFMAs of the current one issue; one barrier per K-tile.
//
// Staging: global -> registers -> LDS (K-tile 16, rows padded to 80 B: conflict-free ds_read_b128), loads issued two
// K-tiles ahead in two alternating register sets, LDS double buffer.  (An LDS-DMA engine -- global_load_lds_dwordx4,
// K-tile 32, XOR-swizzled unpadded image -- was built and measured in round 1: 2-10 % SLOWER at every hot-path
// shape; it was removed.)
// Two A-addressing modes: plain row-major (A + m*lda) and the implicit-GEMM gather of the stem's second 3x3/stride-2
// convolution over a channel-last activation (row m = (b,t2,f2); K index = (kf,kt,ci)).
#include "gemm_shared.h"

namespace {

#define GEMM_MFMA_SLICE(FA, FB)                                                                             \
    _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                           \
    _Pragma("unroll") for (int mt = 0; mt < TM; ++mt)                                                       \
    _Pragma("unroll") for (int nt = 0; nt < TN; ++nt)                                                       \
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(FB[nt][e], FA[mt][e], acc[mt][nt], 0, 0, 0)

// ---- the kernel: register staging, K-tile 16 (any K % 4 == 0) ---------------------------------------------------
// LN fold (inference; the four LayerNorms in front of a block's Linear layers leave the launch list):
//   LN == 2, producer: the epilogue also writes, per 32 stored columns of every C row, (sum, M2 about their own mean);
//   LN == 1, consumer: A is the UN-normalised row, W / bias are the folded W.diag(gamma) / b + W.beta; a prologue merges the
//            partials of the tile's rows into (mean, rstd) in LDS (Chan's formula, fixed order) while the first K-tiles are
//            in flight and the epilogue applies rstd * (acc - mean * colsum).  The K-loop is untouched.
// BK = 16: the round-1/2 loop (two register staging sets, loads two K-tiles ahead).  BK = 32 (round 3): a K-tile row is a whole
// 128-byte cache line -- with BK = 16 every wave-level staging load touches 16 half lines and the two halves of a line are
// fetched by different instructions a K-tile apart (L1 is 32 KB per CU: the line is usually gone, so L2 serves it twice) --
// one register staging set, loads one (twice as deep) K-tile ahead, one barrier per 32 contraction steps; 144-byte LDS rows.
// F: epilogue flags (gemm_shared.h EPF_*): the LN-fold role, and what the epilogue does NOT need to carry -- the inference entry
// points compile the dropout mask generation and the 16-bit-output paths out (a bias / Swish / residual kernel was 46-58 KB of
// code, 2.4x its ReLU sibling, almost all of it epilogue that runs once per tile and is fetched cold).
template <int BM, int BN, int EPI, bool CONV, int F = 0, int BK = 16>
__global__ __launch_bounds__(256, (BM == 128 && BN == 128) ? (BK == 32 ? 2 : 3) : (BK == 32 && BM + BN > 128 ? 2 : 4))
void gemm_f32_kernel(const GemmArgs g) {
    static_assert(BK == 16 || BK == 32, "K-tile 16 or 32");
    constexpr int LN = F & EPF_LN_MASK;
    constexpr int TM = BM / 64, TN = BN / 64, LDSR = BK + 4;   // 80- / 144-byte rows: conflict-free ds_read_b128
    static_assert(EPI != EPI_GLU || TN == 2, "GLU keeps value and gate tiles in one wave");
    // staging double buffer; the row-major epilogue re-uses it as 4 per-wave transposition tiles (a little larger for the 64x128 tile)
    constexpr int LDS_STAGE = 2 * (BM + BN) * LDSR, LDS_EPI = 4 * 32 * (32 * TN + 4);
    constexpr bool ROWS = LDS_EPI <= LDS_STAGE + 1024;
    constexpr int LDS_MAIN = ROWS && LDS_EPI > LDS_STAGE ? LDS_EPI : LDS_STAGE;
    static_assert(LN == 0 || ROWS, "the LN fold lives in the row-major epilogue");
    __shared__ __attribute__((aligned(16))) float lds[LDS_MAIN + (LN == 1 ? 2 * BM : 0)];
    float* rowstats = lds + LDS_MAIN;   // LN == 1: [BM][2] = (mean, rstd) of the tile's A rows
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
    // split-K stage (small M, long K: a wave that owns a whole-K output tile runs K / 2 dependent MFMAs however few tiles there
    // are): slice blockIdx.y contracts [koff, koff + Kloc) and stores its raw partial tile into slab y
    const int koff = g.ksplit_len > 0 ? (int)blockIdx.y * g.ksplit_len : 0;
    const int Kloc = g.ksplit_len > 0 ? min(g.ksplit_len, g.K - koff) : g.K;
    if (g.trace && tid == 0) {                             // diagnostics only (tools/gemm_tune.py trace)
        g.trace[8 * blockIdx.x + 0] = __builtin_amdgcn_s_memrealtime();
        g.trace[8 * blockIdx.x + 2] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_ID
        g.trace[8 * blockIdx.x + 3] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));  // XCC_ID
    }

    // ---- staging assignment: thread -> (row, 16-byte chunk) for TM rows of A and TN rows of W
    constexpr int CPR = BK / 4, RPP = 256 / CPR, PA = BM / RPP, PB = BN / RPP;   // chunks per row, rows per pass, passes per operand
    const int chunk = tid & (CPR - 1), srow = tid / CPR;
    const float* a_ptr[PA];
    const float* w_ptr[PB];
#pragma unroll
    for (int i = 0; i < PA; ++i) a_ptr[i] = a_row_ptr<CONV>(g, m0 + srow + RPP * i) + (CONV ? 0 : koff);
#pragma unroll
    for (int i = 0; i < PB; ++i) w_ptr[i] = w_row_ptr<EPI, BN>(g, n0, srow + RPP * i) + koff;

    const int nkt = (Kloc + BK - 1) / BK;
    f32x4 ra0[PA], rb0[PB], ra1[BK == 16 ? PA : 1], rb1[BK == 16 ? PB : 1];   // BK = 16: two staging sets (tiles t+1 and t+2 in flight)
    auto load_tile = [&](auto& ra, auto& rb, int kt) {
        int kb = kt * BK;
        if constexpr (CONV == 1 && BK == 16) {
            // The stem's implicit GEMM walks K = (tap, channel) CHANNEL-CHUNK-major: all nine taps of 32 channels, then the next 32
            // (two K-tiles = one whole 128-byte line per tap).  In storage order (tap-major) an h1 line is needed again by another
            // output position of the same row tile 1-6 x 512 K-steps later (stride 2: each input pixel feeds 2.25 outputs) --
            // long after the 4 MB L2 has dropped it: 4.5 GB fetched per launch for 1.28 GB of h1 (PMC FETCH_SIZE).  Walked this way
            // the 2.25 uses of a line fall inside 18 consecutive K-tiles.  The sum is the same, its order differs.
            if (g.conv_kperm) {
                const int c32 = kt / 18, r = kt - 18 * c32;
                kb = (r >> 1) * g.cC + 32 * c32 + 16 * (r & 1);
            }
        }
        const int k = kb + chunk * 4;
        const int64_t aoff = a_k_offset<CONV>(g, kb) + chunk * 4;
        const bool ok = kt * BK + chunk * 4 < Kloc;
#pragma unroll
        for (int i = 0; i < PA; ++i)
            ra[i] = ok ? *reinterpret_cast<const f32x4*>(a_ptr[i] + aoff) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < PB; ++i)
            rb[i] = ok ? *reinterpret_cast<const f32x4*>(w_ptr[i] + k) : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto store_tile = [&](const auto& ra, const auto& rb, int buf) {
#pragma unroll
        for (int i = 0; i < PA; ++i)
            *reinterpret_cast<f32x4*>(As + (buf * BM + srow + RPP * i) * LDSR + chunk * 4) = ra[i];
#pragma unroll
        for (int i = 0; i < PB; ++i)
            *reinterpret_cast<f32x4*>(Bs + (buf * BN + srow + RPP * i) * LDSR + chunk * 4) = rb[i];
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

    // LN == 1: the statistics partials of this tile's BM rows are ONE contiguous block of stats (BM * parts * 8 bytes): it is
    // requested FIRST with coalesced 16-byte loads (two partials each; at most 4 per thread) so that its wait leaves the K-tile
    // loads in flight.  (First version: thread r read the 16 partials of row r with 16 eight-byte loads -- 64 cache lines per
    // wave instruction: +5.5 us on every consumer GEMM, as much as the LayerNorm launch it replaced.)
    constexpr int LN_NL = (BM * 16 / 2 + 255) / 256;                   // float4 loads per thread at 16 partials per row
    [[maybe_unused]] f32x4 lnq[LN == 1 ? LN_NL : 1];
    [[maybe_unused]] float2 ln1 = {0.f, 0.f};
    if constexpr (LN == 1) {
        const int64_t rows_here = min((int64_t)BM, g.M - m0);          // (the last row tile is ragged: clamp inside the block)
        if (g.ln_parts >= 2) {
            const int nf4 = (int)rows_here * (g.ln_parts >> 1);
            const f32x4* sp = reinterpret_cast<const f32x4*>(g.ln_stats + m0 * g.ln_parts * 2);
#pragma unroll
            for (int i = 0; i < LN_NL; ++i) lnq[i] = sp[min(tid + 256 * i, nf4 - 1)];
        } else {
            ln1 = reinterpret_cast<const float2*>(g.ln_stats)[m0 + min((int64_t)tid, rows_here - 1)];
        }
    }
    load_tile(ra0, rb0, 0);
    store_tile(ra0, rb0, 0);
    if constexpr (BK == 16) {
        if (nkt > 1) load_tile(ra1, rb1, 1);
        if (nkt > 2) load_tile(ra0, rb0, 2);
    } else {
        if (nkt > 1) load_tile(ra0, rb0, 1);
    }
    if constexpr (LN == 1) {
        // Merge (Chan, equal counts at every level, fixed tree order): a float4 holds two partials of ni = K / parts values; the
        // parts / 2 consecutive lanes that hold one row's float4s combine with xor-shuffles; the group's first lane writes
        // (mean, rstd).  parts is a power of two <= 16 (checked by the entry point).
        const float ni = (float)(g.K / g.ln_parts), inv_ni = 1.0f / ni, inv_k = 1.0f / (float)g.K;
        if (g.ln_parts >= 2) {
            const int lpr = g.ln_parts >> 1;                             // lanes per row: 1, 2, 4 or 8
#pragma unroll
            for (int i = 0; i < LN_NL; ++i) {
                const float ma = lnq[i][0] * inv_ni, mb = lnq[i][2] * inv_ni, d0 = mb - ma;
                float mean = 0.5f * (ma + mb), m2 = lnq[i][1] + lnq[i][3] + d0 * d0 * (0.5f * ni), cnt = 2.0f * ni;
                for (int st = 1; st < lpr; st <<= 1) {                    // (kernel-uniform trip count)
                    const float mo = __shfl_xor(mean, st, 64), m2o = __shfl_xor(m2, st, 64), dl = mo - mean;
                    mean = 0.5f * (mean + mo);
                    m2 = m2 + m2o + dl * dl * (0.5f * cnt);
                    cnt *= 2.0f;
                }
                const int f = tid + 256 * i, row = f / lpr;
                if ((f & (lpr - 1)) == 0 && row < BM) {
                    rowstats[2 * row] = mean;
                    rowstats[2 * row + 1] = 1.0f / sqrtf(m2 * inv_k + g.ln_eps);
                }
            }
        } else if (tid < BM) {
            rowstats[2 * tid] = ln1.x * inv_k;
            rowstats[2 * tid + 1] = 1.0f / sqrtf(ln1.y * inv_k + g.ln_eps);
        }
    }
    __syncthreads();
    read_frags(fa0, fb0, 0, 0);
    if constexpr (BK == 32) {
        // four 8-deep slices per K-tile, fragments of slice s+1 in flight from LDS under the MFMAs of slice s; the staged registers
        // (tile kt+1) go to the other LDS buffer behind slice 2 and are re-loaded with tile kt+2 at once
        for (int kt = 0; kt < nkt; ++kt) {
            const int cur = kt & 1;
            const bool more = kt + 1 < nkt;
            read_frags(fa1, fb1, cur, 1);
            __builtin_amdgcn_sched_barrier(0);
            GEMM_MFMA_SLICE(fa0, fb0);
            __builtin_amdgcn_sched_barrier(0);
            read_frags(fa0, fb0, cur, 2);
            __builtin_amdgcn_sched_barrier(0);
            GEMM_MFMA_SLICE(fa1, fb1);
            __builtin_amdgcn_sched_barrier(0);
            read_frags(fa1, fb1, cur, 3);
            __builtin_amdgcn_sched_barrier(0);
            GEMM_MFMA_SLICE(fa0, fb0);
            __builtin_amdgcn_sched_barrier(0);
            if (more) {
                store_tile(ra0, rb0, cur ^ 1);
                if (kt + 2 < nkt) load_tile(ra0, rb0, kt + 2);
            }
            __syncthreads();
            if (more) read_frags(fa0, fb0, cur ^ 1, 0);
            __builtin_amdgcn_sched_barrier(0);
            GEMM_MFMA_SLICE(fa1, fb1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    auto k_step = [&](int kt, auto& ra, auto& rb) {   // (ra, rb) holds tile kt+1 on entry
        const int cur = kt & 1;
        const bool more = kt + 1 < nkt;
        read_frags(fa1, fb1, cur, 1);
        __builtin_amdgcn_sched_barrier(0);
        GEMM_MFMA_SLICE(fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
            store_tile(ra, rb, cur ^ 1);
            if (kt + 3 < nkt) load_tile(ra, rb, kt + 3);
        }
        __syncthreads();
        if (more) read_frags(fa0, fb0, cur ^ 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        GEMM_MFMA_SLICE(fa1, fb1);
        __builtin_amdgcn_sched_barrier(0);
    };
    if constexpr (BK == 16) {
        int kt = 0;
        for (; kt + 1 < nkt; kt += 2) {
            k_step(kt, ra1, rb1);
            k_step(kt + 1, ra0, rb0);
        }
        if (kt < nkt) k_step(kt, ra1, rb1);
    }
    if (g.trace && tid == 0) g.trace[8 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();   // main loop done

    // row-major write-out through a per-wave LDS tile (whole 256-byte row segments per store instruction) where the dead
    // staging buffers can hold it; the GLU tile pairs value and gate columns in one wave and stores straight from the accumulators
    if constexpr ((F & EPF_NO_BIAS) != 0) {                // split-K stage: this slice's slab
        GemmArgs gl = g;
        gl.C = g.C + (int64_t)blockIdx.y * g.M * g.ldc;
        __syncthreads();
        gemm_epilogue_rows<BM, BN, EPI, TM, TN, 2, false, F>(gl, acc, m0, n0, wr, wc, lane, lds + wave * 32 * (32 * TN + 4), rowstats);
    } else if constexpr (ROWS) {
        __syncthreads();                                   // every wave is done reading the staging buffers
        gemm_epilogue_rows<BM, BN, EPI, TM, TN, 2, false, F>(g, acc, m0, n0, wr, wc, lane, lds + wave * 32 * (32 * TN + 4), rowstats);
    } else {
        gemm_epilogue<BM, BN, EPI, TM, TN, F>(g, acc, m0, n0, wr, wc, li, hf);
    }
    if (g.trace && tid == 0) {
        g.trace[8 * blockIdx.x + 4] = __builtin_amdgcn_s_memrealtime();                        // epilogue issued
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        g.trace[8 * blockIdx.x + 5] = __builtin_amdgcn_s_memrealtime();                        // ... and drained
    }
}

// ---- tile-shape selection ---------------------------------------------------------------------------------------
// cfg: 0 = 128x128, 1 = 128x64, 2 = 64x128, 3 = 64x64.
// Measured on MI355X (tools/gemm_tune.py, profiles/r01_gemm_tune*.txt): the 128x128 tile only wins when every CU
// gets a long queue of tiles (>= ~12 per CU: conv2, big square GEMMs) so that prologue/epilogue phases of different
// blocks overlap; the K=512 layer GEMMs (4 tiles per CU or fewer, all resident at once) prefer 128x64, and the
// N=512 GEMMs (1 tile per CU at 128x128) prefer 64x64.
inline int choose_tile(int64_t M, int ncols, bool glu, int K) {
    const int bn = glu ? 64 : 128;
    const int64_t n128 = ((M + 127) / 128) * ((ncols + bn - 1) / bn);
    if (n128 >= 12 * 256) return 0;
    if (glu) return 2;
    if (n128 > 690 && n128 <= 768) return 0;      // exactly one full round at 3 blocks/CU (fused QKV: 756 tiles)
    if (n128 >= 3 * 256) return 1;
    // long contractions (the input Linear, K = 9728; FFN out, K = 2048) at N = 512: 64x128 once it gives ~2 workgroups per CU
    // (round 3, tools/gemm_tune.py bk, medians of 7 interleaved rounds: input Linear 637 us vs 653 on 128x64 / 680 on 64x64; FFN out
    // 139.8 vs 142.3 on 64x64)
    if (K >= 2048 && ((M + 63) / 64) * ((ncols + 127) / 128) >= 448) return 2;
    if (K >= 4096 && ((M + 127) / 128) * ((ncols + 63) / 64) >= 448) return 1;
    return 3;
}

template <int BM, int BN, int EPI, bool CONV, int LN = 0, int BK = 16>
int launch_cfg(GemmArgs g, hipStream_t s) {
    const int ncols = EPI == EPI_GLU ? g.n_out : g.N;
    const int bn = EPI == EPI_GLU ? BN / 2 : BN;
    g.tiles_m = (unsigned)((g.M + BM - 1) / BM);
    g.tiles_n = (unsigned)((ncols + bn - 1) / bn);
    // Occupancy control: a CU has 160 KiB of LDS; asking for extra (unused) dynamic LDS caps the resident blocks per
    // CU so that tiles / (256 * blocks_per_CU) lands just under an integer number of rounds (no half-empty last round).
    constexpr int kStatic = 2 * (BM + BN) * (BK + 4) * 4;
    size_t pad = 0;
    if (g.occ_cap > 0) {
        const int per = (160 * 1024) / g.occ_cap;
        pad = per > kStatic ? (size_t)((per - kStatic) & ~255) : 0;
    }
    const unsigned gy = g.ksplit_len > 0 ? (unsigned)((g.K + g.ksplit_len - 1) / g.ksplit_len) : 1u;
    hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, EPI, CONV, LN, BK>), dim3(g.tiles_m * g.tiles_n, gy), dim3(256), pad, s, g);
    return cfm_launch_status();
}

// (BK = 32 is only reachable through cfm_debug_gemm_cfg_f32: measured slower at every hot-path shape but one)
template <int EPI, bool CONV, int F = 0, int BK = 16>
int launch(const GemmArgs& g, hipStream_t s, int force_cfg = -1) {
    const int ncols = EPI == EPI_GLU ? g.n_out : g.N;
    const int tile = force_cfg >= 0 ? force_cfg : choose_tile(g.M, ncols, EPI == EPI_GLU, g.K);
    if constexpr (EPI == EPI_GLU) {
        return (tile == 0 || tile == 1) ? launch_cfg<128, 128, EPI, CONV, F, BK>(g, s) : launch_cfg<64, 128, EPI, CONV, F, BK>(g, s);
    } else {
        switch (tile) {
            case 0: return launch_cfg<128, 128, EPI, CONV, F, BK>(g, s);
            case 1: return launch_cfg<128, 64, EPI, CONV, F, BK>(g, s);
            case 2: return launch_cfg<64, 128, EPI, CONV, F, BK>(g, s);
            default: return launch_cfg<64, 64, EPI, CONV, F, BK>(g, s);
        }
    }
}

// the LN-fold forms live in the row-major vectorised epilogue only
int check_ln(const GemmArgs& g, int ncols) {
    CFM_REQUIRE((g.ldc & 3) == 0 && (ncols & 3) == 0 && CFM_ALIGNED16(g.C) && CFM_ALIGNED16(g.bias), CFM_ERR_ALIGN);
    if (g.R) CFM_REQUIRE((g.ldr & 3) == 0 && CFM_ALIGNED16(g.R), CFM_ERR_ALIGN);
    if (g.stats_out) {
        CFM_REQUIRE((g.N & 31) == 0, CFM_ERR_BAD_SHAPE);
        CFM_REQUIRE((reinterpret_cast<uintptr_t>(g.stats_out) & 7u) == 0, CFM_ERR_ALIGN);
    }
    if (g.ln_stats) {
        CFM_REQUIRE(g.ln_colsum != nullptr, CFM_ERR_NULL);
        CFM_REQUIRE(g.ln_parts >= 1 && g.K % g.ln_parts == 0, CFM_ERR_BAD_SHAPE);
        CFM_REQUIRE(g.ln_parts <= 16 && (g.ln_parts & (g.ln_parts - 1)) == 0, CFM_ERR_UNSUPPORTED);     // 1, 2, 4, 8, 16
        CFM_REQUIRE((reinterpret_cast<uintptr_t>(g.ln_stats) & (g.ln_parts >= 2 ? 15u : 7u)) == 0 && CFM_ALIGNED16(g.ln_colsum),
                    CFM_ERR_ALIGN);
    }
    return CFM_OK;
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
    return launch<EPI_BIAS, false, EPF_INFER>(g, static_cast<hipStream_t>(stream));
}

extern "C" int cfm_gemm_bias_swish_f32(const float* A, const float* W, const float* bias, float* C, int64_t M,
                                       int N, int K, int64_t lda, int64_t ldc, cfm_stream_t stream) {
    GEMM_ARGS_PLAIN(N);
    int st = check(g); if (st) return st;
    CFM_REQUIRE(ldc >= N, CFM_ERR_BAD_SHAPE);
    return launch<EPI_SWISH, false, EPF_INFER>(g, static_cast<hipStream_t>(stream));
}

// training forward of ffn.py:17-18: C = swish(Z), Z = A.W^T + bias is stored too (needed by swish' in the backward)
extern "C" int cfm_gemm_bias_swish_save_f32(const float* A, const float* W, const float* bias, float* C, float* Z,
                                            int64_t M, int N, int K, int64_t lda, int64_t ldc, cfm_stream_t stream) {
    GEMM_ARGS_PLAIN(N);
    g.Zsave = Z;
    int st = check(g); if (st) return st;
    CFM_REQUIRE(Z != nullptr, CFM_ERR_NULL);
    CFM_REQUIRE(ldc >= N, CFM_ERR_BAD_SHAPE);
    return launch<EPI_SWISH, false, EPF_INFER>(g, static_cast<hipStream_t>(stream));
}

// Training forward with dropout fused into the epilogue (element index = row*N + col, regenerated in the backward):
//   epi 0: C = drop(A.W^T + b)                       (encoder.py:23-25)
//   epi 1: C = drop(swish(Z)), Z = A.W^T + b saved   (ffn.py:17-19)
//   epi 4: C = alpha * drop(A.W^T + b) + R           (ffn.py:20-21 / attention.py:17 / convolution.py:29-30 + block.py)
extern "C" int cfm_gemm_train_f32(int epi, const float* A, const float* W, const float* bias, const float* R_or_null,
                                  float alpha, float* C, float* Z_or_null, int64_t M, int N, int K, int64_t lda,
                                  int64_t ldr, int64_t ldc, float drop_p, uint64_t drop_seed, cfm_stream_t stream) {
    GEMM_ARGS_PLAIN(N);
    g.R = R_or_null; g.ldr = ldr; g.alpha = alpha; g.Zsave = Z_or_null; g.drop_p = drop_p; g.drop_seed = drop_seed;
    int st = check(g); if (st) return st;
    CFM_REQUIRE(ldc >= N && drop_p >= 0.f && drop_p < 1.f, CFM_ERR_BAD_SHAPE);
    hipStream_t s = static_cast<hipStream_t>(stream);
    // (the dropout-free instantiations are the inference kernels: a training step with p = 0 runs the same code as eval)
    if (epi == EPI_BIAS) return drop_p > 0.f ? launch<EPI_BIAS, false, EPF_F32_OUT>(g, s) : launch<EPI_BIAS, false, EPF_INFER>(g, s);
    if (epi == EPI_SWISH) return drop_p > 0.f ? launch<EPI_SWISH, false, EPF_F32_OUT>(g, s) : launch<EPI_SWISH, false, EPF_INFER>(g, s);
    if (epi == EPI_RESID) {
        CFM_REQUIRE(R_or_null != nullptr, CFM_ERR_NULL);
        CFM_REQUIRE(ldr >= N, CFM_ERR_BAD_SHAPE);
        return drop_p > 0.f ? launch<EPI_RESID, false, EPF_F32_OUT>(g, s) : launch<EPI_RESID, false, EPF_INFER>(g, s);
    }
    return CFM_ERR_UNSUPPORTED;
}

extern "C" int cfm_gemm_bias_relu_f32(const float* A, const float* W, const float* bias, float* C, int64_t M,
                                      int N, int K, int64_t lda, int64_t ldc, cfm_stream_t stream) {
    GEMM_ARGS_PLAIN(N);
    int st = check(g); if (st) return st;
    CFM_REQUIRE(ldc >= N, CFM_ERR_BAD_SHAPE);
    return launch<EPI_RELU, false, EPF_INFER>(g, static_cast<hipStream_t>(stream));
}

extern "C" int cfm_gemm_bias_glu_f32(const float* A, const float* W, const float* bias, float* C, int64_t M,
                                     int n_out, int K, int64_t lda, int64_t ldc, cfm_stream_t stream) {
    GEMM_ARGS_PLAIN(2 * n_out);
    g.n_out = n_out;
    int st = check(g); if (st) return st;
    CFM_REQUIRE(n_out > 0 && ldc >= n_out, CFM_ERR_BAD_SHAPE);
    return launch<EPI_GLU, false, EPF_INFER>(g, static_cast<hipStream_t>(stream));
}

extern "C" int cfm_gemm_bias_residual_f32(const float* A, const float* W, const float* bias, const float* R,
                                          float alpha, float* C, int64_t M, int N, int K, int64_t lda,
                                          int64_t ldr, int64_t ldc, cfm_stream_t stream) {
    GEMM_ARGS_PLAIN(N);
    g.R = R; g.ldr = ldr; g.alpha = alpha;
    int st = check(g); if (st) return st;
    CFM_REQUIRE(R != nullptr, CFM_ERR_NULL);
    CFM_REQUIRE(ldc >= N && ldr >= N, CFM_ERR_BAD_SHAPE);
    return launch<EPI_RESID, false, EPF_INFER>(g, static_cast<hipStream_t>(stream));
}

// ---- split-K for small M (streaming chunks, short utterances) --------------------------------------------------------
// out = epi(sum_s slab_s + bias): the slabs are summed in a fixed order (bit-reproducible), four columns per thread
namespace {
template <int EPI>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slabs, int splits, const float* __restrict__ bias,
                                                            const float* __restrict__ R, float alpha, float* __restrict__ C,
                                                            int64_t M, int N, int64_t ldr, int64_t ldc) {
    const int n4 = N >> 2;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= M * n4) return;
    const int64_t row = idx / n4;
    const int col = (int)(idx - row * n4) * 4;
    f32x4 acc = *reinterpret_cast<const f32x4*>(slabs + row * N + col);
    for (int s = 1; s < splits; ++s) acc = acc + *reinterpret_cast<const f32x4*>(slabs + ((int64_t)s * M + row) * N + col);
    acc = acc + *reinterpret_cast<const f32x4*>(bias + col);
    if (EPI == EPI_SWISH) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = swishf_acc(acc[e]);
    }
    if (EPI == EPI_RESID) acc = alpha * acc + *reinterpret_cast<const f32x4*>(R + row * ldr + col);
    *reinterpret_cast<f32x4*>(C + row * ldc + col) = acc;
}
}  // namespace

// C = epi(A.W^T + bias) with the contraction split into `splits` slices over gridDim.y (each a workgroup per output tile, raw partial
// tiles into workspace (splits, M, N) fp32) and one reduce + epilogue pass.  epi: 0 bias | 1 +swish | 4 alpha*y + R.
// N % 4 == 0, ldc / ldr % 4 == 0, 16-byte aligned C / R / bias / workspace.  For products whose M gives fewer 64x64 tiles than
// the chip has CUs while K is long (FFN out at M = 1280: 63 -> ~25 us).
extern "C" int cfm_gemm_splitk_f32(int epi, const float* A, const float* W, const float* bias, const float* R_or_null,
                                   float alpha, float* C, float* workspace, int splits, int64_t M, int N, int K,
                                   int64_t lda, int64_t ldr, int64_t ldc, cfm_stream_t stream) {
    GEMM_ARGS_PLAIN(N);
    int st = check(g); if (st) return st;
    CFM_REQUIRE(workspace != nullptr, CFM_ERR_NULL);
    CFM_REQUIRE(splits >= 2 && splits <= 16 && ldc >= N && (N & 3) == 0 && (ldc & 3) == 0, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(CFM_ALIGNED16(C) && CFM_ALIGNED16(bias) && CFM_ALIGNED16(workspace), CFM_ERR_ALIGN);
    CFM_REQUIRE(epi == EPI_BIAS || epi == EPI_SWISH || epi == EPI_RESID, CFM_ERR_UNSUPPORTED);
    if (epi == EPI_RESID) {
        CFM_REQUIRE(R_or_null != nullptr, CFM_ERR_NULL);
        CFM_REQUIRE(ldr >= N && (ldr & 3) == 0 && CFM_ALIGNED16(R_or_null), CFM_ERR_BAD_SHAPE);
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    g.ksplit_len = ((K + splits - 1) / splits + 15) / 16 * 16;
    g.C = workspace; g.ldc = N;
    st = launch<EPI_BIAS, false, EPF_INFER | EPF_NO_BIAS>(g, s, 3);            // 64x64 tiles: the most workgroups
    if (st) return st;
    const int nsl = (K + g.ksplit_len - 1) / g.ksplit_len;
    const int64_t total = M * (N >> 2);
    const dim3 grid((unsigned)((total + 255) / 256)), block(256);
    if (epi == EPI_BIAS) hipLaunchKernelGGL(splitk_reduce_kernel<EPI_BIAS>, grid, block, 0, s, workspace, nsl, bias, R_or_null, alpha, C, M, N, ldr, ldc);
    else if (epi == EPI_SWISH) hipLaunchKernelGGL(splitk_reduce_kernel<EPI_SWISH>, grid, block, 0, s, workspace, nsl, bias, R_or_null, alpha, C, M, N, ldr, ldc);
    else hipLaunchKernelGGL(splitk_reduce_kernel<EPI_RESID>, grid, block, 0, s, workspace, nsl, bias, R_or_null, alpha, C, M, N, ldr, ldc);
    return cfm_launch_status();
}

// ---- the front end's DFT: frames are OVERLAPPING rows of the padded waveform ------------------------------------------
// spec (rows, n_cols) = frames . basis^T with frame r = wave[r * hop : r * hop + n_fft] (lda = hop < K: rows overlap) on the tuned
// forward kernel (no bias).  The caller lays the utterances out with a pitch that is a multiple of hop, so that row r = b * rpu + t
// addresses frame t of utterance b (the last rpu - T rows of an utterance are junk frames, computed and ignored), and leaves
// n_fft floats of slack behind the last utterance.  processor.py:155-158 (torch.stft inside MelSpectrogram).
extern "C" int cfm_dft_frames_f32(const float* wave, const float* basis, float* spec, int64_t rows, int n_cols, int n_fft,
                                  int hop, cfm_stream_t stream) {
    CFM_REQUIRE(wave && basis && spec, CFM_ERR_NULL);
    CFM_REQUIRE(rows > 0 && n_cols > 0 && n_fft > 0 && hop > 0 && (n_fft & 3) == 0 && (hop & 3) == 0 && (n_cols & 3) == 0, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(CFM_ALIGNED16(wave) && CFM_ALIGNED16(basis) && CFM_ALIGNED16(spec), CFM_ERR_ALIGN);
    GemmArgs g{};
    g.A = wave; g.W = basis; g.bias = basis; g.C = spec; g.M = rows; g.N = n_cols; g.K = n_fft; g.lda = hop; g.ldc = n_cols; g.alpha = 1.f;
    return launch<EPI_BIAS, false, EPF_INFER | EPF_NO_BIAS>(g, static_cast<hipStream_t>(stream));
}

// ---- LayerNorm folded into its neighbours (inference) ---------------------------------------------------------------
// producers: the plain / residual GEMM that also emits the LayerNorm statistics partials of the rows it stores
extern "C" int cfm_gemm_bias_stats_f32(const float* A, const float* W, const float* bias, float* C, float* stats_out,
                                       int64_t M, int N, int K, int64_t lda, int64_t ldc, cfm_stream_t stream) {
    GEMM_ARGS_PLAIN(N);
    g.stats_out = stats_out;
    int st = check(g); if (st) return st;
    CFM_REQUIRE(stats_out != nullptr, CFM_ERR_NULL);
    CFM_REQUIRE(ldc >= N, CFM_ERR_BAD_SHAPE);
    st = check_ln(g, N); if (st) return st;
    return launch<EPI_BIAS, false, EPF_INFER | EPF_LN_PRODUCE>(g, static_cast<hipStream_t>(stream));
}

extern "C" int cfm_gemm_bias_residual_stats_f32(const float* A, const float* W, const float* bias, const float* R,
                                                float alpha, float* C, float* stats_out, int64_t M, int N, int K,
                                                int64_t lda, int64_t ldr, int64_t ldc, cfm_stream_t stream) {
    GEMM_ARGS_PLAIN(N);
    g.R = R; g.ldr = ldr; g.alpha = alpha; g.stats_out = stats_out;
    int st = check(g); if (st) return st;
    CFM_REQUIRE(R != nullptr && stats_out != nullptr, CFM_ERR_NULL);
    CFM_REQUIRE(ldc >= N && ldr >= N, CFM_ERR_BAD_SHAPE);
    st = check_ln(g, N); if (st) return st;
    return launch<EPI_RESID, false, EPF_INFER | EPF_LN_PRODUCE>(g, static_cast<hipStream_t>(stream));
}

// consumer: C = epi(LN(A).W^T + b) computed from the UN-normalised A, its statistics partials and the folded parameters
// Wf = W.diag(gamma), bias_f = b + W.beta, colsum[n] = sum_k Wf[n,k].  epi: 0 bias | 1 +swish | 3 +GLU (N = n_out, Wf has 2N rows).
extern "C" int cfm_gemm_lnfold_f32(int epi, const float* A, const float* ln_stats, int ln_parts, float ln_eps,
                                   const float* Wf, const float* bias_f, const float* colsum, float* C, int64_t M, int N,
                                   int K, int64_t lda, int64_t ldc, cfm_stream_t stream) {
    const float* W = Wf; const float* bias = bias_f;
    GEMM_ARGS_PLAIN(epi == EPI_GLU ? 2 * N : N);
    if (epi == EPI_GLU) g.n_out = N;
    g.ln_stats = ln_stats; g.ln_parts = ln_parts; g.ln_eps = ln_eps; g.ln_colsum = colsum;
    int st = check(g); if (st) return st;
    CFM_REQUIRE(ln_stats != nullptr, CFM_ERR_NULL);
    CFM_REQUIRE(N > 0 && ldc >= N && ln_eps >= 0.f, CFM_ERR_BAD_SHAPE);
    st = check_ln(g, N); if (st) return st;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (epi == EPI_BIAS) return launch<EPI_BIAS, false, EPF_INFER | EPF_LN_CONSUME>(g, s);
    if (epi == EPI_SWISH) return launch<EPI_SWISH, false, EPF_INFER | EPF_LN_CONSUME>(g, s);
    if (epi == EPI_GLU) return launch<EPI_GLU, false, EPF_INFER | EPF_LN_CONSUME>(g, s);
    return CFM_ERR_UNSUPPORTED;
}

static int g_conv2_bk = 16;
static int g_conv2_kperm = 1;
// Implicit-GEMM second stem convolution (3x3, stride 2, channel-last input, packed weight).  Declared in the stem
// section of the ABI; lives here to share the kernel templates.
extern "C" int cfm_subsample_conv2_relu_f32(const float* h1, const float* w2p, const float* b2, float* h2, int B,
                                            int F1, int T1, int C, cfm_stream_t stream) {
    CFM_REQUIRE(h1 && w2p && b2 && h2, CFM_ERR_NULL);
    CFM_REQUIRE(B > 0 && F1 >= 3 && T1 >= 3 && C > 0, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(C % 16 == 0, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(CFM_ALIGNED16(h1) && CFM_ALIGNED16(w2p) && CFM_ALIGNED16(h2), CFM_ERR_ALIGN);
    GemmArgs g{};
    g.cT1 = T1; g.cF1 = F1; g.cC = C; g.cT2 = (T1 - 1) / 2; g.cF2 = (F1 - 1) / 2;
    g.A = h1; g.W = w2p; g.bias = b2; g.C = h2;
    g.M = (int64_t)B * g.cT2 * g.cF2; g.N = C; g.K = 9 * C; g.lda = 0; g.ldc = C; g.alpha = 1.f;
    g.conv_kperm = g_conv2_kperm && C % 32 == 0;
    if (g_conv2_bk == 32 && C % 32 == 0) return launch<EPI_RELU, true, EPF_INFER, 32>(g, static_cast<hipStream_t>(stream));
    return launch<EPI_RELU, true, EPF_INFER>(g, static_cast<hipStream_t>(stream));
}

// diagnostics (tools/conv2_bk_ab.py): K-tile of the stem's implicit GEMM: 16 (default) | 32; returns the previous setting
extern "C" int cfm_debug_set_conv2_bk(int bk) {
    const int prev = g_conv2_bk;
    if (bk == 16 || bk == 32) g_conv2_bk = bk;
    if (bk == 0 || bk == 1) g_conv2_kperm = bk;            // 0 / 1: K walked in storage order / channel-chunk-major (default)
    return prev;
}

// Tuning / diagnostics: the residual-epilogue GEMM with a forced block-tile shape
// (cfg 0..3 = 128x128, 128x64, 64x128, 64x64; -1 = the built-in choice).  Same results for every cfg.
// trace_or_null: device buffer of 8 x uint64 per block: {start, main-loop end, HW_ID, XCC_ID, epilogue issued, epilogue
// drained, -, -}; times in 100 MHz s_memrealtime ticks.
extern "C" int cfm_debug_gemm_cfg_f32(int cfg, const float* A, const float* W, const float* bias, const float* R,
                                      float alpha, float* C, int64_t M, int N, int K, void* trace_or_null,
                                      cfm_stream_t stream) {
    GemmArgs g{}; g.A = A; g.W = W; g.bias = bias; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = K; g.ldc = N;
    g.R = R; g.ldr = N; g.alpha = alpha; g.trace = static_cast<unsigned long long*>(trace_or_null);
    int st = check(g); if (st) return st;
    CFM_REQUIRE(R != nullptr, CFM_ERR_NULL);
    g.occ_cap = cfg >= 0 ? (cfg >> 8) : 0;            // cfg + 256*cap: blocks/CU cap
    if (cfg >= 0) cfg &= 255;
    CFM_REQUIRE(cfg >= -1 && (cfg < 0 || (cfg & 15) <= 3) && cfg < 128, CFM_ERR_BAD_SHAPE);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (cfg >= 64) {                                                     // cfg + 64: K-tile 32 (whole cache lines per staged row)
        cfg &= 63;
        if (cfg >= 32) return launch<EPI_SWISH, false, EPF_INFER, 32>(g, s, cfg & 15);
        if (cfg >= 16) return launch<EPI_BIAS, false, EPF_INFER, 32>(g, s, cfg & 15);
        return launch<EPI_RESID, false, EPF_INFER, 32>(g, s, cfg);
    }
    if (cfg >= 32) return launch<EPI_SWISH, false, EPF_INFER>(g, s, cfg & 15);     // cfg + 32: swish epilogue (no residual read)
    if (cfg >= 16) return launch<EPI_BIAS, false, EPF_INFER>(g, s, cfg & 15);      // cfg + 16: bias epilogue
    return launch<EPI_RESID, false, EPF_INFER>(g, s, cfg);
}
