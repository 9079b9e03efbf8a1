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
#include <stdlib.h>

#include "gemm_shared.h"

namespace {

// W16: the weight operand is already stored in the 16-bit type (a cached cast of the fp32 master weights, made once per
// optimizer step): half the L2->LDS bytes of the operand that every row tile re-reads, and no conversion.
// A16 (with W16): the activation operand is a 16-bit tensor too (LayerNorm / Swish / stem conv1 outputs written in the
// 16-bit type by their producers), lda in elements; the stem's implicit-GEMM gather uses the same element offsets.
// WM = waves along M (2: 256 threads, block tiles up to 128x128; 4: 512 threads, 256x128 / 256x256 with one workgroup per CU).
// The K-loop is bound by the operand stream L2 -> L1 -> LDS: at 128x128 two co-resident workgroups pull 64 KB per ~1 us K-tile
// step into a CU (16 TB/s chip-wide, about half of the L2's peak) for 0.23 us of MFMA work each -- deeper prefetch and
// fragment pipelining changed nothing -- so the big tiles exist to halve the bytes per MFMA.
// CONV: 0 plain rows; 1 the stem's 3x3 / stride-2 im2col gather (forward conv2); 2 a parity class of its transposed conv (the
// input gradient of conv2): row = class position (b, a, c), K-tile = a tap (dt, df) of dz2 -- rows whose tap falls outside dz2
// are zero (per-row validity, re-derived when the tap changes), output rows are scattered to the class's dh1 positions.
// F: epilogue flags (gemm_shared.h EPF_*).  EPF_NO_DROPOUT: the launcher picks the mask-free instantiation whenever drop_p == 0
// (inference, and training without dropout) -- the mask generation was most of a 30-55 KB kernel whose epilogue is fetched cold.
template <typename T16, int BM, int BN, int EPI, int CONV, bool W16, bool A16 = false, int WM = 2, int F = 0>
__global__ __launch_bounds__(WM * 128, 2) void gemm_mfma16_kernel(const GemmArgs g) {
    static_assert(!A16 || W16, "16-bit A operand comes together with 16-bit weights");
    using x8 = typename Lowp<T16>::x8;
    using x4 = typename Lowp<T16>::x4;
    constexpr int NT = WM * 128;                              // threads
    constexpr int TM = BM / (32 * WM), TN = BN / 64, BK = 64;
    constexpr int ROWB = 72;                                  // LDS row in bf16 elements: 64 + 8 pad = 144 B
    constexpr int RP4 = NT / 16, RP8 = NT / 8;                // tile rows staged per pass: 16 lanes (fp32) / 8 lanes (16-bit) per row
    constexpr int NA = BM / RP4, NB = BN / RP4;               // float4 loads per thread per K-tile
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
    const int wr = wave >> 1, wc = wave & 1;                 // wr in [0, WM)
    const int li = lane & 31, hf = lane >> 5;

    // ---- staging: 16 lanes cover the 256 B of one tile row; pass p handles rows srow + 16 p
    const int srow = tid >> 4, sch = tid & 15;
    constexpr int NBH = BN / RP8;                             // W16: 16-byte loads per thread per K-tile (8 lanes per row)
    const int hrow = tid >> 3, hch = tid & 7;
    constexpr int NAH = BM / RP8;
    const float* a_ptr[A16 ? 1 : NA];
    const T16* ah_ptr[A16 ? NAH : 1];
    const float* w_ptr[W16 ? 1 : NB];
    const T16* wh_ptr[W16 ? NBH : 1];
    if (A16) {
#pragma unroll
        for (int p = 0; p < NAH; ++p) {
            ah_ptr[p] = reinterpret_cast<const T16*>(g.A) + (a_row_ptr<CONV>(g, m0 + hrow + RP8 * p) - g.A);   // element offset
        }
    } else {
#pragma unroll
        for (int p = 0; p < NA; ++p) a_ptr[p] = a_row_ptr<CONV>(g, m0 + srow + RP4 * p);
    }
    constexpr int NCLS = CONV == 2 ? (A16 ? NAH : NA) : 1;
    int cls_a[NCLS], cls_c[NCLS];                                     // class coordinates of the rows this thread stages
    if constexpr (CONV == 2) {
#pragma unroll
        for (int p = 0; p < NCLS; ++p) {
            const int64_t m = min(m0 + (A16 ? hrow + RP8 * p : srow + RP4 * p), g.M - 1);
            const int per = g.pA * g.pC;
            const int r = (int)(m - (m / per) * per);
            cls_a[p] = r / g.pC;
            cls_c[p] = r - cls_a[p] * g.pC;
        }
    }
    if (W16) {
#pragma unroll
        for (int p = 0; p < NBH; ++p)
            wh_ptr[p] = reinterpret_cast<const T16*>(g.W) + (int64_t)w_row_index<EPI, BN>(g, n0, hrow + RP8 * p) * g.K;
    } else {
#pragma unroll
        for (int p = 0; p < NB; ++p) w_ptr[p] = w_row_ptr<EPI, BN>(g, n0, srow + RP4 * p);
    }
    // All loads are unconditional: a K-tile chunk beyond K (only possible in the last tile when K % 64 != 0) reads a clamped
    // address and is zeroed by a select when it is staged into LDS (`kvalid*`), after the MFMAs of the current tile.
    // Two staging register sets: the loads of K-tiles kt+1 and kt+2 are both in flight while tile kt is multiplied.  The
    // per-K-tile timeline (tools/gemm16_sites.py trace) showed 1.2-1.6 us per K-tile against 0.23 us of MFMA time with one
    // tile in flight: a round trip to L2 under load per 64-deep step.
    struct Stage {
        f32x4 ra[A16 ? 1 : NA], rb[W16 ? 1 : NB];
        x8 rah[A16 ? NAH : 1], rbh[W16 ? NBH : 1];
        bool kvalid, kvalid_h;
        unsigned amask;                                       // CONV == 2: bit p = row p of this thread has the K-tile's tap inside dz2
    };
    constexpr bool DEEP = WM == 2;                            // 8-wave tiles: one set (the second one spills at 256x256)
    Stage st0, st1;
    auto load_tile = [&](Stage& t, int kt) {
        int kb = kt * BK;
        if constexpr (CONV == 1) {
            // stem conv2: K = (tap, channel) walked channel-chunk-major -- the nine taps of 64 channels, then the next 64 -- so that the
            // 2.25 uses of an h1 line (stride 2) fall within nine consecutive K-tiles instead of up to 6 x C K-steps apart (gemm_f32.hip)
            if (g.conv_kperm) { const int c64 = kt / 9; kb = (kt - 9 * c64) * g.cC + 64 * c64; }
        }
        const int k = kb + sch * 4, kh = kb + hch * 8;                 // K % 4 == 0 (K % 8 == 0 for 16-bit operands)
        t.kvalid = k < g.K; t.kvalid_h = kh < g.K;
        const int kc = max(0, min(k, g.K - 4)), khc = max(0, min(kh, g.K - 8));
        if (A16 && CONV == 2) {
            const int tap = (khc - hch * 8) / g.cC;                   // (uniform: a K-tile lies inside one tap, C % 64 == 0)
            const int dt = g.tap_dt[tap], df = g.tap_df[tap];
            const int64_t ahoff = a_k_offset<CONV>(g, khc - hch * 8) + hch * 8;
            t.amask = 0;
#pragma unroll
            for (int p = 0; p < NAH; ++p) {
                const bool ok = (unsigned)(cls_a[p] + dt) < (unsigned)g.cT2 && (unsigned)(cls_c[p] + df) < (unsigned)g.cF2;
                t.rah[p] = *reinterpret_cast<const x8*>(ok ? ah_ptr[p] + ahoff : reinterpret_cast<const T16*>(g.A) + hch * 8);
                t.amask |= ok ? (1u << p) : 0u;
            }
        } else if (A16) {
            const int64_t ahoff = a_k_offset<CONV>(g, khc - hch * 8) + hch * 8;
#pragma unroll
            for (int p = 0; p < NAH; ++p) t.rah[p] = *reinterpret_cast<const x8*>(ah_ptr[p] + ahoff);
        } else if constexpr (CONV == 2) {
            const int tap = (kc - sch * 4) / g.cC;                    // (uniform: a K-tile lies inside one tap, C % 64 == 0)
            const int dt = g.tap_dt[tap], df = g.tap_df[tap];
            const int64_t aoff = a_k_offset<CONV>(g, kc - sch * 4) + sch * 4;
            t.amask = 0;
#pragma unroll
            for (int p = 0; p < NA; ++p) {                            // unconditional load: invalid rows read the tensor's first row
                const bool ok = (unsigned)(cls_a[p] + dt) < (unsigned)g.cT2 && (unsigned)(cls_c[p] + df) < (unsigned)g.cF2;
                t.ra[p] = *reinterpret_cast<const f32x4*>(ok ? a_ptr[p] + aoff : g.A + sch * 4);
                t.amask |= ok ? (1u << p) : 0u;
            }
        } else {
            const int64_t aoff = a_k_offset<CONV>(g, kc - sch * 4) + sch * 4;
#pragma unroll
            for (int p = 0; p < NA; ++p) t.ra[p] = *reinterpret_cast<const f32x4*>(a_ptr[p] + aoff);
        }
        if (W16) {
#pragma unroll
            for (int p = 0; p < NBH; ++p) t.rbh[p] = *reinterpret_cast<const x8*>(wh_ptr[p] + khc);
        } else {
#pragma unroll
            for (int p = 0; p < NB; ++p) t.rb[p] = *reinterpret_cast<const f32x4*>(w_ptr[p] + kc);
        }
    };
    auto store_tile = [&](const Stage& t, int buf) {
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        x8 z8;
#pragma unroll
        for (int e = 0; e < 8; ++e) z8[e] = (T16)0.f;
        if (A16) {
#pragma unroll
            for (int p = 0; p < NAH; ++p)
                *reinterpret_cast<x8*>(As + (buf * BM + hrow + RP8 * p) * ROWB + hch * 8) =
                    (t.kvalid_h && (CONV != 2 || ((t.amask >> p) & 1u))) ? t.rah[p] : z8;
        } else {
#pragma unroll
            for (int p = 0; p < NA; ++p)
                *reinterpret_cast<x4*>(As + (buf * BM + srow + RP4 * p) * ROWB + sch * 4) =
                    Lowp<T16>::cvt4((t.kvalid && (CONV != 2 || ((t.amask >> p) & 1u))) ? t.ra[p] : z4);
        }
        if (W16) {
#pragma unroll
            for (int p = 0; p < NBH; ++p)
                *reinterpret_cast<x8*>(Bs + (buf * BN + hrow + RP8 * p) * ROWB + hch * 8) = t.kvalid_h ? t.rbh[p] : z8;
        } else {
#pragma unroll
            for (int p = 0; p < NB; ++p)
                *reinterpret_cast<x4*>(Bs + (buf * BN + srow + RP4 * p) * ROWB + sch * 4) = Lowp<T16>::cvt4(t.kvalid ? t.rb[p] : z4);
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
    const int a_row = wr * (BM / WM) + li, b_row = wc * (BN / 2) + li;
    const int nkt = (g.K + BK - 1) / BK;
    const bool tracer = g.trace && tid == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2);   // diagnostics only
    unsigned long long* tr = g.trace + (blockIdx.x == 0 ? 0 : 64);
    if (tracer) tr[0] = __builtin_amdgcn_s_memrealtime();
    load_tile(st0, 0);
    store_tile(st0, 0);
    if (DEEP) {
        load_tile(st1, 1);                                    // (tiles beyond K read a clamped address and are never staged)
        load_tile(st0, 2);
    } else {
        load_tile(st0, 1);
    }
    __syncthreads();
    if (tracer) tr[1] = __builtin_amdgcn_s_memrealtime();
    auto k_step = [&](int kt, Stage& nxt) {                   // `nxt` holds tile kt+1 on entry and is refilled with tile kt+3
        const int cur = kt & 1;
        const bool more = kt + 1 < nkt;
        // per-wave software pipeline: the fragments of MFMA step s+1 are read while step s multiplies (one exposed LDS
        // latency per K-tile instead of four: the trace showed ~1.2 us per K-tile against 0.23 us of MFMA time)
        constexpr int NF = (WM == 4 && BN == 256 && !A16) ? 1 : 2;   // (256x256 with an fp32 A operand: no registers left for a second set)
        x8 fa[NF][TM], fb[NF][TN];
        auto read_frags = [&](int set, int s) {
#pragma unroll
            for (int t = 0; t < TM; ++t)
                fa[set][t] = *reinterpret_cast<const x8*>(As + (cur * BM + a_row + 32 * t) * ROWB + 16 * s + 8 * hf);
#pragma unroll
            for (int t = 0; t < TN; ++t)
                fb[set][t] = *reinterpret_cast<const x8*>(Bs + (cur * BN + b_row + 32 * t) * ROWB + 16 * s + 8 * hf);
        };
        if (NF == 2) read_frags(0, 0);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (NF == 2) { if (s < 3) read_frags((s + 1) & 1, s + 1); }
            else read_frags(0, s);
#pragma unroll
            for (int mt = 0; mt < TM; ++mt)
#pragma unroll
                for (int nt = 0; nt < TN; ++nt)
                    acc[mt][nt] = Lowp<T16>::mfma(fb[s & (NF - 1)][nt], fa[s & (NF - 1)][mt], acc[mt][nt]);
            // staging rides in the shadow of the MFMAs: the next tile goes to LDS after the first MFMA step (its buffer has been
            // free since the last barrier), the refill is requested after the second.  At the end of the K-tile all 8 waves did
            // this at once with the matrix pipe idle: 2.3 us per K-tile against 1.5 us for the last one, which stages nothing.
            if (W16 && more && s == 0) store_tile(nxt, cur ^ 1);
            if (W16 && more && s == 1) load_tile(nxt, kt + (DEEP ? 3 : 2));
        }
        if (!W16) {                                           // (fp32 weights: the early form spills; staging after the MFMAs)
            __builtin_amdgcn_sched_barrier(0);
            if (more) {
                store_tile(nxt, cur ^ 1);
                load_tile(nxt, kt + (DEEP ? 3 : 2));
            }
        }
        __syncthreads();
        if (tracer && kt < 60) tr[2 + kt] = __builtin_amdgcn_s_memrealtime();
    };
    int kt = 0;
    if (DEEP) {
        for (; kt + 1 < nkt; kt += 2) {
            k_step(kt, st1);
            k_step(kt + 1, st0);
        }
        if (kt < nkt) k_step(kt, st1);
    } else {
        for (; kt < nkt; ++kt) k_step(kt, st0);
    }
    static_assert(EPI != EPI_GLU || WM == 2, "GLU tiles: 2x2 waves");
    {                                                        // (the K-loop ended on a __syncthreads: the staging buffers are free)
        static_assert(2 * WM * 32 * (32 * TN + 4) * 4 <= 2 * (BM + BN) * ROWB * 2, "row-major epilogue scratch must fit the staging LDS");
        gemm_epilogue_rows<BM, BN, EPI, TM, TN, WM, CONV == 2, F>(g, acc, m0, n0, wr, wc, lane, reinterpret_cast<float*>(lds) + wave * 32 * (32 * TN + 4));
    }
    if (tracer) {
        tr[62] = __builtin_amdgcn_s_memrealtime();                       // epilogue issued
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tr[63] = __builtin_amdgcn_s_memrealtime();                       // ... and drained
    }
}

template <typename T16, int BM, int BN, int EPI, int CONV>
int launch_cfg(GemmArgs g, int src16, hipStream_t s) {          // src16: 0 = fp32 operands, 1 = 16-bit W, 2 = 16-bit A and W
    const int ncols = EPI == EPI_GLU ? g.n_out : g.N;
    const int bn = EPI == EPI_GLU ? BN / 2 : BN;
    g.tiles_m = (unsigned)((g.M + BM - 1) / BM);
    g.tiles_n = (unsigned)((ncols + bn - 1) / bn);
    const dim3 grid(g.tiles_m * g.tiles_n);
    // dropout only exists for the bias / Swish / residual epilogues of the training forward (and, as a mask replay, in EPI_DSWISH)
    constexpr bool CAN_DROP = EPI == EPI_BIAS || EPI == EPI_SWISH || EPI == EPI_RESID;
    constexpr int FN = EPI == EPI_DSWISH ? 0 : EPF_NO_DROPOUT;
    if (CAN_DROP && g.drop_p > 0.f) {
        if constexpr (CAN_DROP) {
            if (src16 == 2) hipLaunchKernelGGL((gemm_mfma16_kernel<T16, BM, BN, EPI, CONV, true, true, 2, 0>), grid, dim3(256), 0, s, g);
            else if (src16 == 1) hipLaunchKernelGGL((gemm_mfma16_kernel<T16, BM, BN, EPI, CONV, true, false, 2, 0>), grid, dim3(256), 0, s, g);
            else if (src16 == 0) hipLaunchKernelGGL((gemm_mfma16_kernel<T16, BM, BN, EPI, CONV, false, false, 2, 0>), grid, dim3(256), 0, s, g);
            else return CFM_ERR_UNSUPPORTED;
        }
        return cfm_launch_status();
    }
    if (src16 == 2) hipLaunchKernelGGL((gemm_mfma16_kernel<T16, BM, BN, EPI, CONV, true, true, 2, FN>), grid, dim3(256), 0, s, g);
    else if (src16 == 1) hipLaunchKernelGGL((gemm_mfma16_kernel<T16, BM, BN, EPI, CONV, true, false, 2, FN>), grid, dim3(256), 0, s, g);
    else if (src16 == 0) hipLaunchKernelGGL((gemm_mfma16_kernel<T16, BM, BN, EPI, CONV, false, false, 2, FN>), grid, dim3(256), 0, s, g);
    else return CFM_ERR_UNSUPPORTED;
    return cfm_launch_status();
}

inline bool gemm_epilogue_vec_ok_host(const GemmArgs& g, int epi) {      // the row-major epilogue's alignment conditions
    return ((g.ldc & 3) == 0) && ((g.N & 3) == 0) && (epi != EPI_RESID || (g.ldr & 3) == 0) &&
           ((reinterpret_cast<uintptr_t>(g.C) & (g.c_prec ? 7 : 15)) == 0) &&
           (epi != EPI_RESID || (reinterpret_cast<uintptr_t>(g.R) & 15) == 0);
}

template <typename T16, int BM, int BN, int EPI, int CONV = 0>
int launch_big(GemmArgs g, int src16, hipStream_t s) {          // 512-thread workgroups, 16-bit weights, no im2col gather, no GLU
    g.tiles_m = (unsigned)((g.M + BM - 1) / BM);
    g.tiles_n = (unsigned)((g.N + BN - 1) / BN);
    const dim3 grid(g.tiles_m * g.tiles_n);
    constexpr bool CAN_DROP = EPI == EPI_BIAS || EPI == EPI_SWISH || EPI == EPI_RESID;
    constexpr int FN = EPI == EPI_DSWISH ? 0 : EPF_NO_DROPOUT;
    if constexpr (CONV != 0) {
        if (src16 == 2) hipLaunchKernelGGL((gemm_mfma16_kernel<T16, BM, BN, EPI, CONV, true, true, 4, FN>), grid, dim3(512), 0, s, g);
        else hipLaunchKernelGGL((gemm_mfma16_kernel<T16, BM, BN, EPI, CONV, true, false, 4, FN>), grid, dim3(512), 0, s, g);
    } else {
        if (CAN_DROP && g.drop_p > 0.f) {
            if constexpr (CAN_DROP) {
                if (src16 == 2) hipLaunchKernelGGL((gemm_mfma16_kernel<T16, BM, BN, EPI, 0, true, true, 4, 0>), grid, dim3(512), 0, s, g);
                else hipLaunchKernelGGL((gemm_mfma16_kernel<T16, BM, BN, EPI, 0, true, false, 4, 0>), grid, dim3(512), 0, s, g);
            }
            return cfm_launch_status();
        }
        if (src16 == 2) hipLaunchKernelGGL((gemm_mfma16_kernel<T16, BM, BN, EPI, 0, true, true, 4, FN>), grid, dim3(512), 0, s, g);
        else hipLaunchKernelGGL((gemm_mfma16_kernel<T16, BM, BN, EPI, 0, true, false, 4, FN>), grid, dim3(512), 0, s, g);
    }
    return cfm_launch_status();
}

template <typename T16, int EPI, int CONV>
int launch_t(const GemmArgs& g, int src16, hipStream_t s) {
    const int ncols = EPI == EPI_GLU ? g.n_out : g.N;
    const int bn = EPI == EPI_GLU ? 64 : 128;
    const int64_t t128 = ((g.M + 127) / 128) * ((ncols + bn - 1) / bn);
    if constexpr (EPI == EPI_GLU) {
        return t128 >= 512 ? launch_cfg<T16, 128, 128, EPI, CONV>(g, src16, s) : launch_cfg<T16, 64, 128, EPI, CONV>(g, src16, s);
    } else {
        if constexpr (CONV != 0) {                                // the stem's conv2 (forward im2col gather / transposed-conv classes):
            if (src16 >= 1 && gemm_epilogue_vec_ok_host(g, EPI) && (g.N % 256) == 0 && g.occ_cap != 1 &&      // 10^5 rows, N = C
                (g.M + 255) / 256 * (g.N / 256) >= 224)
                return launch_big<T16, 256, 256, EPI, CONV>(g, src16, s);
        }
        if constexpr (CONV == 0) {
            // big tiles (one 8-wave workgroup per CU) once they fill the chip about once over: half / three quarters of the
            // operand bytes per MFMA of the 128x128 tile
            const int force = g.occ_cap;                          // tuning hook: 1 = 128x128 family, 2 = 256x128, 3 = 256x256
            if (src16 >= 1 && gemm_epilogue_vec_ok_host(g, EPI)) {
                const int64_t rows256 = (g.M + 255) / 256;
                const int64_t t256 = rows256 * ((g.N + 255) / 256), t2128 = rows256 * ((g.N + 127) / 128);
                if (force == 3 || (force == 0 && t256 >= 190)) return launch_big<T16, 256, 256, EPI>(g, src16, s);   // (192 tiles: the B=32 QKV product, 25.7 vs 29.9 us on 256x128)
                if (force == 2 || (force == 0 && t2128 >= 224)) return launch_big<T16, 256, 128, EPI>(g, src16, s);
            }
        }
        if (t128 >= 512 && g.occ_cap < 4) return launch_cfg<T16, 128, 128, EPI, CONV>(g, src16, s);   // (tuning hook: 4 = 128x64, 5 = 64x64)
        if constexpr (CONV == 0 && (EPI == EPI_RESID || EPI == EPI_BIAS)) {
            // between the two: 128x64 tiles (three fragment reads per two MFMAs instead of two per one) once they give every CU
            // about two workgroups -- the N = 512 products of the B = 32 forward (FFN out, attention out, pointwise conv 2)
            const int64_t t12864 = ((g.M + 127) / 128) * ((g.N + 63) / 64);
            if ((t12864 >= 448 && g.K >= 1024 && g.occ_cap < 4) || g.occ_cap == 4) return      // (K = 512: no difference, measured)
                launch_cfg<T16, 128, 64, EPI, CONV>(g, src16, s);
        }
        return launch_cfg<T16, 64, 64, EPI, CONV>(g, src16, s);
    }
}

template <typename T16>
__global__ __launch_bounds__(256) void cast16_kernel(const float* __restrict__ src, T16* __restrict__ dst, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    *reinterpret_cast<typename Lowp<T16>::x4*>(dst + 4 * i) = Lowp<T16>::cvt4(reinterpret_cast<const f32x4*>(src)[i]);
}

// Multi-tensor form: one launch casts up to CFM_CAST_BATCH weights (the table rides in the kernel arguments, as in the fused
// Adam); transpose = 1 writes the (cols, rows) transpose (32 x 32 tiles through LDS: coalesced on both sides) -- the dX = dY.W
// products that run on the forward kernel want W^T.  grid = (blocks of the largest item, items).
struct CastBatch { cfm_cast_item it[CFM_CAST_BATCH]; int count; };

template <typename T16>
__global__ __launch_bounds__(256) void cast16_multi_kernel(const CastBatch b) {
    const cfm_cast_item it = b.it[blockIdx.y];
    T16* dst = static_cast<T16*>(it.dst);
    if (!it.transpose) {
        const int64_t n4 = it.rows * it.cols / 4;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256)
            *reinterpret_cast<typename Lowp<T16>::x4*>(dst + 4 * i) = Lowp<T16>::cvt4(reinterpret_cast<const f32x4*>(it.src)[i]);
        return;
    }
    __shared__ float tile[32][33];
    const int64_t tr = (it.rows + 31) / 32, tc = (it.cols + 31) / 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;                 // 32 x 8 threads
    for (int64_t t = blockIdx.x; t < tr * tc; t += gridDim.x) {
        const int64_t r0 = (t / tc) * 32, c0 = (t % tc) * 32;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t r = r0 + ty + 8 * j, c = c0 + tx;
            tile[ty + 8 * j][tx] = (r < it.rows && c < it.cols) ? it.src[r * it.cols + c] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t c = c0 + ty + 8 * j, r = r0 + tx;                 // dst is (cols, rows)
            if (c < it.cols && r < it.rows) dst[c * it.rows + r] = (T16)tile[tx][ty + 8 * j];
        }
        __syncthreads();
    }
}

template <int EPI, int CONV>
int launch(int prec, const GemmArgs& g, int src16, hipStream_t s) {
    if (prec == CFM_PREC_BF16) return launch_t<__bf16, EPI, CONV>(g, src16, s);
    if (prec == CFM_PREC_FP16) return launch_t<_Float16, EPI, CONV>(g, src16, s);
    return CFM_ERR_UNSUPPORTED;
}

}  // namespace

static unsigned long long* g_gemm16_trace = nullptr;     // diagnostics only (cfm_debug_gemm_mfma16_trace)
static int g_gemm16_force_tile = 0;                       // tuning only (cfm_debug_gemm_mfma16_force_tile)

// diagnostics only (tools/gemm16_sites.py trace): the next cfm_gemm_mfma16_f32 launches record s_memrealtime (100 MHz) stamps
// of thread 0 of the first and of the middle workgroup: [0] start, [1] first K-tile staged, [2+kt] K-tile kt done, [62] epilogue
// issued, [63] epilogue drained (2 x 64 uint64); NULL switches it off.
extern "C" int cfm_debug_gemm_mfma16_trace(void* trace_or_null) {
    g_gemm16_trace = static_cast<unsigned long long*>(trace_or_null);
    return CFM_OK;
}

// tuning only (tools/gemm16_sites.py sweep): 0 = built-in choice, 1 = 128x128 / 64x64 family, 2 = 256x128, 3 = 256x256
extern "C" int cfm_debug_gemm_mfma16_force_tile(int tile) {
    g_gemm16_force_tile = tile;
    return CFM_OK;
}

// prec: CFM_PREC_BF16 | CFM_PREC_FP16.  epi: 0 bias | 1 bias+swish | 2 bias+relu | 3 bias+GLU (N = n_out columns of C,
// W has 2*n_out rows) | 4 alpha*y + R.  Same layouts and argument rules as the fp32 entry points (cfm_gemm_train_f32 for
// Z_or_null / drop_p / drop_seed); results differ from them by the 16-bit rounding of A and W only.
extern "C" int cfm_gemm_mfma16_f32(int prec, int epi, const void* A, int a_is_16bit, const void* W, int w_is_16bit,
                                   const float* bias, const float* R_or_null, float alpha, void* C, int c_is_16bit,
                                   void* Z_or_null, int z_is_16bit, int64_t M, int N, int K, int64_t lda, int64_t ldr,
                                   int64_t ldc, float drop_p, uint64_t drop_seed, cfm_stream_t stream) {
    CFM_REQUIRE(A && W && C && (bias || epi == EPI_DSWISH), CFM_ERR_NULL);
    // a 16-bit Z takes the vectorised epilogue only: whole 8-byte groups of four columns
    CFM_REQUIRE(!z_is_16bit || (Z_or_null && (epi == EPI_SWISH || epi == EPI_DSWISH) && (N & 7) == 0 && (ldc & 7) == 0 && CFM_ALIGNED16(C) && CFM_ALIGNED16(Z_or_null)),
                CFM_ERR_UNSUPPORTED);
    CFM_REQUIRE(!w_is_16bit || (K & 7) == 0, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(!a_is_16bit || (w_is_16bit && (lda & 7) == 0), CFM_ERR_BAD_SHAPE);
    const int src16 = a_is_16bit ? 2 : (w_is_16bit ? 1 : 0);
    CFM_REQUIRE(M > 0 && N > 0 && K > 0 && (K & 3) == 0 && (lda & 3) == 0 && lda >= K && ldc >= N, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(drop_p >= 0.f && drop_p < 1.f, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(CFM_ALIGNED16(A) && CFM_ALIGNED16(W), CFM_ERR_ALIGN);
    GemmArgs g{};
    g.A = static_cast<const float*>(A); g.W = static_cast<const float*>(W); g.bias = bias; g.R = R_or_null;
    g.C = static_cast<float*>(C); g.c_prec = c_is_16bit ? prec : 0; g.M = M; g.K = K; g.lda = lda;
    g.ldr = ldr; g.ldc = ldc; g.alpha = alpha; g.Zsave = static_cast<float*>(Z_or_null); g.z_prec = z_is_16bit ? prec : 0; g.drop_p = drop_p; g.drop_seed = drop_seed;
    g.trace = g_gemm16_trace;
    g.occ_cap = g_gemm16_force_tile;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (epi == EPI_GLU) {
        g.n_out = N; g.N = 2 * N;
        return launch<EPI_GLU, 0>(prec, g, src16, s);
    }
    g.N = N;
    switch (epi) {
        case EPI_BIAS: return launch<EPI_BIAS, 0>(prec, g, src16, s);
        case EPI_SWISH: return launch<EPI_SWISH, 0>(prec, g, src16, s);
        case EPI_RELU: return launch<EPI_RELU, 0>(prec, g, src16, s);
        case EPI_RESID:
            CFM_REQUIRE(R_or_null != nullptr, CFM_ERR_NULL);
            CFM_REQUIRE(ldr >= N, CFM_ERR_BAD_SHAPE);
            return launch<EPI_RESID, 0>(prec, g, src16, s);
        case EPI_DSWISH:                                        // backward: C = alpha * (A.W^T) * swish'(Z); Z is READ, ldr = its leading dim
            CFM_REQUIRE(Z_or_null != nullptr, CFM_ERR_NULL);
            CFM_REQUIRE((N & 7) == 0 && (ldc & 7) == 0 && (ldr & 7) == 0 && ldr >= N && CFM_ALIGNED16(C) && CFM_ALIGNED16(Z_or_null),
                        CFM_ERR_UNSUPPORTED);
            return launch<EPI_DSWISH, 0>(prec, g, src16, s);
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
    g.occ_cap = g_gemm16_force_tile;
    static const bool storage_order = getenv("CONFORMER_AMD_CONV2_KORDER_STORAGE") != nullptr;   // (A/B only; read once)
    g.conv_kperm = C % 64 == 0 && !storage_order;
    return launch<EPI_RELU, 1>(prec, g, h1_is_16bit ? 2 : (w_is_16bit ? 1 : 0), static_cast<hipStream_t>(stream));
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

// Multi-tensor cfm_cast16_f32: items[i].dst (16-bit, prec) <- RNE(items[i].src), an (rows, cols) fp32 matrix, optionally
// transposed on the way (dst is then (cols, rows)).  rows * cols % 4 == 0 for the non-transposing items; any count (launched
// in batches of CFM_CAST_BATCH).  One launch per batch instead of one per weight: the ~130 per-step weight casts of a
// Conformer-L training step were 0.66 ms of 5 us launches for 0.13 ms worth of bytes.
extern "C" int cfm_cast16_multi_f32(int prec, const cfm_cast_item* items, int count, cfm_stream_t stream) {
    CFM_REQUIRE(items && count > 0, CFM_ERR_NULL);
    CFM_REQUIRE(prec == CFM_PREC_BF16 || prec == CFM_PREC_FP16, CFM_ERR_UNSUPPORTED);
    hipStream_t s = static_cast<hipStream_t>(stream);
    for (int base = 0; base < count; base += CFM_CAST_BATCH) {
        CastBatch b{};
        b.count = count - base < CFM_CAST_BATCH ? count - base : CFM_CAST_BATCH;
        int64_t most = 1;
        for (int i = 0; i < b.count; ++i) {
            const cfm_cast_item& it = items[base + i];
            CFM_REQUIRE(it.src && it.dst && it.rows > 0 && it.cols > 0, CFM_ERR_BAD_SHAPE);
            CFM_REQUIRE(it.transpose || ((it.rows * it.cols) & 3) == 0, CFM_ERR_BAD_SHAPE);
            CFM_REQUIRE(CFM_ALIGNED16(it.src) && (reinterpret_cast<uintptr_t>(it.dst) & 7) == 0, CFM_ERR_ALIGN);
            b.it[i] = it;
            const int64_t blocks = it.transpose ? ((it.rows + 31) / 32) * ((it.cols + 31) / 32) : (it.rows * it.cols / 4 + 255) / 256;
            most = blocks > most ? blocks : most;
        }
        const dim3 grid((unsigned)(most > 1024 ? 1024 : most), (unsigned)b.count);
        if (prec == CFM_PREC_BF16) hipLaunchKernelGGL(cast16_multi_kernel<__bf16>, grid, dim3(256), 0, s, b);
        else hipLaunchKernelGGL(cast16_multi_kernel<_Float16>, grid, dim3(256), 0, s, b);
    }
    return cfm_launch_status();
}

// Input gradient of the stem's conv2 (3x3, stride 2) on the forward kernel: dh1 (B,T1,F1,C) = conv_transpose(dz2 (B,T2,F2,C), w2)
// as four parity-class implicit GEMMs (t1 = 2a+pt, f1 = 2c+pf; the class's taps (kt,kf) = (pt,pf) mod 2), CONV == 2.
// w2c16: the transposed-pack of w2 (cfm_pack_conv2_weight_t_f32) cast to `prec` (cfm_cast16_f32); zero_bias: C zeros; dz2 fp32
// or stored in `prec` (cfm_relu_bwd_out16_f32).
// Every dh1 element is written exactly once.  C % 64 == 0.
static int conv2_bwd_input_classes(int prec, const void* dz2, int dz2_is_16bit, const void* w2c16, const float* zero_bias,
                                   void* dh1, int dh1_is_16bit, int B, int F1, int T1, int C, cfm_stream_t stream) {
    CFM_REQUIRE(dz2 && w2c16 && zero_bias && dh1, CFM_ERR_NULL);
    CFM_REQUIRE(B > 0 && F1 >= 3 && T1 >= 3 && C > 0, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(C % 64 == 0, CFM_ERR_UNSUPPORTED);
    CFM_REQUIRE(CFM_ALIGNED16(dz2) && CFM_ALIGNED16(w2c16) && CFM_ALIGNED16(dh1) && CFM_ALIGNED16(zero_bias), CFM_ERR_ALIGN);
    const int T2 = (T1 - 1) / 2, F2 = (F1 - 1) / 2;
    const int64_t woff[4] = {0, 4, 6, 8};                 // class q = 2*pt + pf: taps before it in the transposed pack
    hipStream_t s = static_cast<hipStream_t>(stream);
    for (int pt = 0; pt < 2; ++pt)
        for (int pf = 0; pf < 2; ++pf) {
            GemmArgs g{};
            g.cT1 = T1; g.cF1 = F1; g.cT2 = T2; g.cF2 = F2; g.cC = C; g.pt = pt; g.pf = pf;
            g.pA = (T1 - pt + 1) / 2; g.pC = (F1 - pf + 1) / 2;
            if (g.pA <= 0 || g.pC <= 0) continue;
            int nt = 0;                                   // taps (kt,kf) with kt = pt (mod 2), kf = pf (mod 2); t2 = a - (kt-pt)/2
            for (int kt = pt; kt < 3; kt += 2)
                for (int kf = pf; kf < 3; kf += 2) { g.tap_dt[nt] = -(kt - pt) / 2; g.tap_df[nt] = -(kf - pf) / 2; ++nt; }
            const int q = 2 * pt + pf;
            g.A = static_cast<const float*>(dz2); g.W = reinterpret_cast<const float*>(static_cast<const char*>(w2c16) + woff[q] * C * C * 2);
            g.bias = zero_bias; g.C = static_cast<float*>(dh1); g.c_prec = dh1_is_16bit ? prec : 0;
            g.M = (int64_t)B * g.pA * g.pC; g.N = C; g.K = nt * C; g.lda = 0; g.ldc = C; g.alpha = 1.f;
            const int st = launch<EPI_BIAS, 2>(prec, g, dz2_is_16bit ? 2 : 1, s);
            if (st) return st;
        }
    return CFM_OK;
}
extern "C" int cfm_subsample_conv2_bwd_input_fwdkernel_mfma16_f32(int prec, const void* dz2, int dz2_is_16bit, const void* w2c16,
                                                                  const float* zero_bias, float* dh1, int B, int F1, int T1, int C,
                                                                  cfm_stream_t stream) {
    return conv2_bwd_input_classes(prec, dz2, dz2_is_16bit, w2c16, zero_bias, dh1, 0, B, F1, T1, C, stream);
}
// ... with dh1 stored in the 16-bit type `prec`: its only consumer is the conv1 parameter-gradient reduction
// (cfm_subsample_conv1_bwd_d16_f32) -- under torch.autocast conv1's incoming gradient IS a 16-bit tensor.  Half the 2.5 GB.
extern "C" int cfm_subsample_conv2_bwd_input_fwdkernel_out16_mfma16_f32(int prec, const void* dz2, int dz2_is_16bit,
                                                                        const void* w2c16, const float* zero_bias, void* dh1_16,
                                                                        int B, int F1, int T1, int C, cfm_stream_t stream) {
    return conv2_bwd_input_classes(prec, dz2, dz2_is_16bit, w2c16, zero_bias, dh1_16, 1, B, F1, T1, C, stream);
}
