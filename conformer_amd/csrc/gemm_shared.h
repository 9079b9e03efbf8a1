// Pieces shared by the fp32-MFMA GEMM (gemm_f32.hip) and the bf16-MFMA GEMM (gemm_bf16.hip): argument block, operand
// row addressing (plain / GLU row interleave / implicit-GEMM conv gather) and the 16-byte epilogue.
#pragma once
#include "cfm_common.h"

namespace {

enum Epi { EPI_BIAS = 0, EPI_SWISH = 1, EPI_RELU = 2, EPI_GLU = 3, EPI_RESID = 4, EPI_DSWISH = 5 };
// Epilogue flags (template parameter F of the epilogue routines; 0 = everything, what the 16-bit kernels use):
//   EPF_LN_CONSUME / EPF_LN_PRODUCE  the folded-LayerNorm roles (see gemm_epilogue_apply);
//   EPF_NO_DROPOUT                   no dropout mask code (g.drop_p is ignored: the entry point guarantees it is 0);
//   EPF_F32_OUT                      C (and Z) are fp32: the 8-column 16-bit write-out path is not compiled.
//   EPF_NO_BIAS                      the accumulators are stored as they are (split-K partial slabs)
enum EpiFlags { EPF_LN_CONSUME = 1, EPF_LN_PRODUCE = 2, EPF_LN_MASK = 3, EPF_NO_DROPOUT = 4, EPF_F32_OUT = 8,
                EPF_INFER = EPF_NO_DROPOUT | EPF_F32_OUT, EPF_NO_BIAS = 16 };
// EPI_DSWISH (16-bit kernels, backward): C = alpha * acc * swish'(Z) [* the forward's dropout mask]; Z = g.Zsave (READ, leading
// dimension g.ldr, type g.z_prec), no bias.  Vectorised epilogue only (the entry point checks the alignment conditions).

struct GemmArgs {
    const float* A; const float* W; const float* bias; const float* R; float* C;
    float* Zsave;                   // swish epilogue, training: also store the pre-activation (same ldc), or NULL
    int z_prec;                     // 0: Zsave is fp32; CFM_PREC_BF16 / CFM_PREC_FP16: stored in that type (vectorised path only)
    float drop_p; unsigned long long drop_seed;   // training: dropout on the GEMM result (after Swish; before alpha*y+R)
    int64_t M; int N; int K; int64_t lda, ldr, ldc; float alpha;
    int n_out;                      // GLU: output columns (N = 2*n_out)
    int cT2, cF2, cT1, cF1, cC;     // conv mode geometry
    int conv_kperm;                 // fp32 kernel, CONV == 1: walk K channel-chunk-major (see load_tile)
    // CONV == 2 (16-bit kernel): transposed-conv parity class of the stem's conv2 backward (see gemm_bwd_args.h, GATHER 2): row
    // m = (b, a, c) of the class grid pA x pC, K-tile = tap (dt, df) of dz2 (B,T2,F2,C); output row = dh1 position (b, 2a+pt, 2c+pf)
    int pA, pC, pt, pf; int tap_dt[4], tap_df[4];
    unsigned tiles_m, tiles_n;
    int occ_cap;                    // 0 = natural; else blocks/CU cap enforced through a dynamic-LDS pad
    int c_prec;                     // 0: C is fp32; CFM_PREC_BF16 / CFM_PREC_FP16: C is stored in that 16-bit type (ldc in elements)
    unsigned long long* trace;      // diagnostics: per-block {start, end} s_memrealtime stamps + HW id, or NULL
    // LayerNorm folded into the GEMMs either side of it (fp32 kernel, inference; see gemm_f32.hip "LN fold"):
    float* stats_out;               // producer (LN == 2): [M][N/32][2] = per 32 stored columns of a C row (sum, M2 about their own mean)
    const float* ln_stats;          // consumer (LN == 1): [M][ln_parts][2] partials of the UN-normalised A rows (equal column counts)
    const float* ln_colsum;         // consumer: colsum[n] = sum_k W'[n,k], W' = W.diag(gamma) (GLU: value rows then gate rows)
    int ln_parts; float ln_eps;
    int ksplit_len;                 // fp32 kernel, split-K stage: > 0 = gridDim.y slices of this many contraction steps, slice y writes
                                    // its raw partial products to C + y * M * ldc (summed by splitk_reduce_kernel)
};

// ---- shared pieces --------------------------------------------------------------------------------------------
template <int CONV>
__device__ __forceinline__ const float* a_row_ptr(const GemmArgs& g, int64_t m) {
    if (m >= g.M) m = g.M - 1;                           // clamp: the row is loaded but never stored
    if (CONV == 2) {                                     // class row (b, a, c): formal base of dz2[b][a][c][:] (taps add (dt, df))
        const int per = g.pA * g.pC;
        const int b = (int)(m / per), r = (int)(m - (int64_t)b * per);
        const int a = r / g.pC, c = r - a * g.pC;
        return g.A + (((int64_t)b * g.cT2 + a) * g.cF2 + c) * (int64_t)g.cC;
    }
    if (CONV == 1) {
        const int f2 = (int)(m % g.cF2);
        const int64_t bt = m / g.cF2;
        const int t2 = (int)(bt % g.cT2);
        const int64_t b = bt / g.cT2;
        return g.A + (((b * g.cT1 + 2 * t2) * g.cF1 + 2 * f2) * (int64_t)g.cC);
    }
    return g.A + m * g.lda;
}

template <int EPI, int BN>
__device__ __forceinline__ int w_row_index(const GemmArgs& g, int n0, int r) {          // r = LDS row of the W tile
    int n;
    if (EPI == EPI_GLU) {                                // wave wc: n-tile 0 = value cols, n-tile 1 = gate cols
        const int j = r & 31, nt = (r >> 5) & 1, w = r >> 6;
        n = n0 + w * 32 + j;
        if (n >= g.n_out) n = g.n_out - 1;
        n += nt * g.n_out;
    } else {
        n = n0 + r;
        if (n >= g.N) n = g.N - 1;
    }
    return n;
}

template <int EPI, int BN>
__device__ __forceinline__ const float* w_row_ptr(const GemmArgs& g, int n0, int r) {
    return g.W + (int64_t)w_row_index<EPI, BN>(g, n0, r) * g.K;
}

template <int CONV>
__device__ __forceinline__ int64_t a_k_offset(const GemmArgs& g, int k) {   // k = first K index of an aligned slab
    if (CONV == 2) {                                     // k = (tap, co): the tap's (dt, df) neighbour of the class row
        const int tap = k / g.cC, co = k - tap * g.cC;
        return ((int64_t)g.tap_dt[tap] * g.cF2 + g.tap_df[tap]) * g.cC + co;
    }
    if (CONV == 1) {
        const int tap = k / g.cC, ci = k - tap * g.cC;
        const int kf = tap / 3, ktp = tap - 3 * kf;
        return ((int64_t)ktp * g.cF1 + kf) * g.cC + ci;
    }
    return k;
}

// The MFMAs are issued with W fragments as the A operand and activation fragments as the B operand, so an accumulator
// tile is C^T: lane (li, hf) holds output ROW m = tile_row + li and, in registers 4q..4q+3, the four CONSECUTIVE
// columns n = tile_col + 8q + 4hf + {0,1,2,3}.
//
// gemm_epilogue_at: bias / activation / dropout / residual / stores for FOUR consecutive columns of one output row
// (av = the accumulator values, gv = the matching gate values of a GLU tile).
// `crow`: the row of C the result is stored to (differs from `row` only for the scattered rows of CONV == 2)
// Operands of one 4-column group that come from memory (vectorised path): FETCHED by the caller ahead of the stores of earlier
// groups.  Loading them inside the per-group routine made every epilogue a chain of round trips -- the wait for group i's bias /
// residual load (s_waitcnt vmcnt(0)) also waits for group i-1's stores to be acknowledged: 64 serialised ~0.3 us trips, 19 us of
// a 40 us workgroup in the 256x256 FFN tile (per-K-tile trace, profiles/r02_gemm16_epilogue_trace.txt).
struct EpiOps {
    f32x4 bb, bg, rr;               // bias, GLU gate bias, residual (or fp32 Z of EPI_DSWISH)
    unsigned z16[2];                // EPI_DSWISH: four 16-bit Z values, converted when they are used
    f32x4 cs, cg;                   // LN fold (consumer): column sums of the folded weight (values, GLU gates)
};
template <int EPI, int F = 0>
__device__ __forceinline__ void gemm_epilogue_fetch_bias(const GemmArgs& g, int col, EpiOps& o) {
    constexpr int LN = F & EPF_LN_MASK;
    const int ncols = EPI == EPI_GLU ? g.n_out : g.N;
    const int cc = col < ncols ? col : 0;                              // (groups beyond the last column are never stored)
    if (F & EPF_NO_BIAS) o.bb = f32x4{0.f, 0.f, 0.f, 0.f};
    else if (EPI != EPI_DSWISH) o.bb = *reinterpret_cast<const f32x4*>(g.bias + cc);
    if (EPI == EPI_GLU) o.bg = *reinterpret_cast<const f32x4*>(g.bias + g.n_out + cc);
    if constexpr (LN == 1) {
        o.cs = *reinterpret_cast<const f32x4*>(g.ln_colsum + cc);
        if (EPI == EPI_GLU) o.cg = *reinterpret_cast<const f32x4*>(g.ln_colsum + g.n_out + cc);
    }
}
template <int EPI>
__device__ __forceinline__ void gemm_epilogue_fetch_row(const GemmArgs& g, int64_t row, int col, EpiOps& o) {
    if (EPI != EPI_RESID && EPI != EPI_DSWISH) return;
    const int64_t rc = row < g.M ? row : g.M - 1;
    const int cc = col < g.N ? col : 0;
    if (EPI == EPI_RESID) o.rr = *reinterpret_cast<const f32x4*>(g.R + rc * g.ldr + cc);
    if (EPI == EPI_DSWISH) {
        if (g.z_prec == 0) o.rr = *reinterpret_cast<const f32x4*>(g.Zsave + rc * g.ldr + cc);
        else {
            const uint2 t = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(g.Zsave) + rc * g.ldr + cc);
            o.z16[0] = t.x; o.z16[1] = t.y;
        }
    }
}

// gemm_epilogue_compute: activation / dropout / residual of one fetched 4-column group; returns the values to store,
// zpre = the pre-activation (what EPI_SWISH saves as Z)
template <int EPI, int F = 0>
__device__ __forceinline__ f32x4 gemm_epilogue_compute(const GemmArgs& g, const f32x4 av, const f32x4 gv, const EpiOps& o, int64_t row,
                                                       int col, f32x4& zpre) {
    if constexpr (EPI == EPI_DSWISH) {
        f32x4 z4;
        if (g.z_prec == 0) z4 = o.rr;
        else if (g.z_prec == CFM_PREC_BF16) {                          // bf16 -> fp32: the bits, shifted up
            z4 = f32x4{__uint_as_float(o.z16[0] << 16), __uint_as_float(o.z16[0] & 0xffff0000u), __uint_as_float(o.z16[1] << 16),
                       __uint_as_float(o.z16[1] & 0xffff0000u)};
        } else {
            Lowp<_Float16>::x4 t;
            __builtin_memcpy(&t, o.z16, 8);
            z4 = f32x4{(float)t[0], (float)t[1], (float)t[2], (float)t[3]};
        }
        f32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float sg = sigmoidf_acc(z4[e]);
            r[e] = g.alpha * av[e] * (sg * (1.0f + z4[e] * (1.0f - sg)));
        }
        if (!(F & EPF_NO_DROPOUT) && g.drop_p > 0.f) {                 // (vectorised path: N % 4 == 0 and col % 4 == 0 -- an aligned group)
            float keep[4];
            dropout_keep4(g.drop_seed, (unsigned long long)row * (unsigned long long)g.N + (unsigned)col, g.drop_p, 1.0f / (1.0f - g.drop_p), keep);
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] *= keep[e];
        }
        zpre = z4;
        return r;
    }
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = av[e] + o.bb[e];
    const bool drop = !(F & EPF_NO_DROPOUT) && g.drop_p > 0.f && EPI != EPI_GLU && EPI != EPI_RELU;
    const float inv_keep = drop ? 1.0f / (1.0f - g.drop_p) : 1.0f;
    const unsigned long long e0 = (unsigned long long)row * (unsigned long long)g.N + (unsigned)col;
    float keep[4] = {1.f, 1.f, 1.f, 1.f};
    if (drop) dropout_keep4(g.drop_seed, e0, g.drop_p, inv_keep, keep);   // (vectorised path: e0 % 4 == 0 -- one hash for the group)
    if (drop && EPI != EPI_SWISH) {                        // dropout(y) then alpha*y + R  (or plain y)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= keep[e];
    }
    if (EPI == EPI_GLU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= sigmoidf_acc(gv[e] + o.bg[e]);
    }
    if (EPI == EPI_RESID) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = g.alpha * v[e] + o.rr[e];
    }
    zpre = f32x4{v[0], v[1], v[2], v[3]};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (EPI == EPI_SWISH) {
            v[e] = swishf_acc(v[e]);
            if (drop) v[e] *= keep[e];
        }
        if (EPI == EPI_RELU) v[e] = fmaxf(v[e], 0.f);
    }
    return f32x4{v[0], v[1], v[2], v[3]};
}

// stores of 4 / 8 consecutive elements at element offset `off` of a tensor of type `prec` (0 fp32 | bf16 | fp16)
template <bool NT = false>   // NT: non-temporal (tensors saved for the backward pass: written now, read tens of milliseconds later)
__device__ __forceinline__ void epi_store4(void* base, int prec, int64_t off, const f32x4 v) {
    if (prec == 0) {
        if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + off));
        else *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + off) = v;
    } else if (prec == CFM_PREC_BF16) {
        const Lowp<__bf16>::x4 r = Lowp<__bf16>::cvt4(v);
        if (NT) __builtin_nontemporal_store(r, reinterpret_cast<Lowp<__bf16>::x4*>(reinterpret_cast<__bf16*>(base) + off));
        else *reinterpret_cast<Lowp<__bf16>::x4*>(reinterpret_cast<__bf16*>(base) + off) = r;
    } else {
        const Lowp<_Float16>::x4 r = Lowp<_Float16>::cvt4(v);
        if (NT) __builtin_nontemporal_store(r, reinterpret_cast<Lowp<_Float16>::x4*>(reinterpret_cast<_Float16*>(base) + off));
        else *reinterpret_cast<Lowp<_Float16>::x4*>(reinterpret_cast<_Float16*>(base) + off) = r;
    }
}
template <bool NT = false>
__device__ __forceinline__ void epi_store8(void* base, int prec, int64_t off, const f32x4 v0, const f32x4 v1) {
    if (prec == 0) {
        if (NT) {
            __builtin_nontemporal_store(v0, reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + off));
            __builtin_nontemporal_store(v1, reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + off + 4));
        } else {
            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + off) = v0;
            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + off + 4) = v1;
        }
    } else if (prec == CFM_PREC_BF16) {
        Lowp<__bf16>::x8 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) { r[e] = (__bf16)v0[e]; r[4 + e] = (__bf16)v1[e]; }
        if (NT) __builtin_nontemporal_store(r, reinterpret_cast<Lowp<__bf16>::x8*>(reinterpret_cast<__bf16*>(base) + off));
        else *reinterpret_cast<Lowp<__bf16>::x8*>(reinterpret_cast<__bf16*>(base) + off) = r;
    } else {
        Lowp<_Float16>::x8 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) { r[e] = (_Float16)v0[e]; r[4 + e] = (_Float16)v1[e]; }
        if (NT) __builtin_nontemporal_store(r, reinterpret_cast<Lowp<_Float16>::x8*>(reinterpret_cast<_Float16*>(base) + off));
        else *reinterpret_cast<Lowp<_Float16>::x8*>(reinterpret_cast<_Float16*>(base) + off) = r;
    }
}

// gemm_epilogue_apply: compute + stores of one fetched 4-column group (vectorised path)
// LN == 1 (consumer of a folded LayerNorm): the accumulator holds x.W'^T of the UN-normalised row x; with (mean, rstd) of the row
//   (rowstats: the tile's [BM][2] table in LDS, lrow = row - m0) LN(x).W^T = rstd * (acc - mean * colsum) (+ the folded bias).
// LN == 2 (producer): every lane takes part (no early exit); the 8 lanes that hold 32 consecutive columns of a row reduce
//   (sum, M2 about their own mean) of the STORED values and lane 0 of the group writes the partial (Chan-mergeable: no
//   sum-of-squares cancellation).  Needs N % 32 == 0 (a group is all in or all out).
template <int EPI, int F = 0>
__device__ __forceinline__ void gemm_epilogue_apply(const GemmArgs& g, f32x4 av, f32x4 gv, const EpiOps& o, int64_t row,
                                                    int col, int64_t crow, const float* rowstats = nullptr, int lrow = 0) {
    constexpr int LN = F & EPF_LN_MASK;
    const int ncols = EPI == EPI_GLU ? g.n_out : g.N;
    const bool ok = row < g.M && col < ncols;
    if constexpr (LN != 2) { if (!ok) return; }
    if constexpr (LN == 1) {
        const float mean = rowstats[2 * lrow], rstd = rowstats[2 * lrow + 1];
        av = (av - mean * o.cs) * rstd;
        if (EPI == EPI_GLU) gv = (gv - mean * o.cg) * rstd;
    }
    f32x4 zpre;
    const f32x4 v = gemm_epilogue_compute<EPI, F>(g, av, gv, o, row, col, zpre);
    if (ok) {
        if (EPI == EPI_SWISH && g.Zsave) epi_store4<true>(g.Zsave, (F & EPF_F32_OUT) ? 0 : g.z_prec, row * g.ldc + col, zpre);
        epi_store4(g.C, (F & EPF_F32_OUT) ? 0 : g.c_prec, crow * g.ldc + col, v);
    }
    if constexpr (LN == 2) {
        float s = (v[0] + v[1]) + (v[2] + v[3]);
        s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
        const float m = s * (1.0f / 32.0f);
        const f32x4 dv = v - m;
        float q = (dv[0] * dv[0] + dv[1] * dv[1]) + (dv[2] * dv[2] + dv[3] * dv[3]);
        q += __shfl_xor(q, 1, 64); q += __shfl_xor(q, 2, 64); q += __shfl_xor(q, 4, 64);
        if (ok && (col & 31) == 0)
            *reinterpret_cast<float2*>(g.stats_out + (row * (int64_t)(g.N >> 5) + (col >> 5)) * 2) = float2{s, q};
    }
}
// ... of two adjacent groups = 8 consecutive columns of one row: a 16-bit output is then one 16-byte store per lane.  The
// write-out of a tile is bound by the NUMBER of store instructions (~42 CU-cycles each at 8 or at 16 bytes per lane: the
// 256x256 Swish tile with C and Z in bf16 took 19 us, as long as its whole K-loop, at 8-byte stores)
template <int EPI>
__device__ __forceinline__ void gemm_epilogue_apply8(const GemmArgs& g, const f32x4 av0, const f32x4 av1, const EpiOps& o0,
                                                     const EpiOps& o1, int64_t row, int col, int64_t crow) {
    static_assert(EPI != EPI_GLU, "8-column groups: no GLU");
    if (row >= g.M || col >= g.N) return;                              // (N % 8 == 0: a group is all in or all out)
    f32x4 z0, z1;
    const f32x4 v0 = gemm_epilogue_compute<EPI>(g, av0, av0, o0, row, col, z0);
    const f32x4 v1 = gemm_epilogue_compute<EPI>(g, av1, av1, o1, row, col + 4, z1);
    if (EPI == EPI_SWISH && g.Zsave) epi_store8<true>(g.Zsave, g.z_prec, row * g.ldc + col, z0, z1);
    epi_store8(g.C, g.c_prec, crow * g.ldc + col, v0, v1);
}

// gemm_epilogue_at: fetch + apply of one group (scalar path for odd leading dimensions / widths)
template <int EPI, int F = 0>
__device__ __forceinline__ void gemm_epilogue_at(const GemmArgs& g, const f32x4 av, const f32x4 gv, int64_t row, int col,
                                                 bool vec_ok, int64_t crow = -1) {
    if (crow < 0) crow = row;
    const int ncols = EPI == EPI_GLU ? g.n_out : g.N;
    if (row >= g.M || col >= ncols) return;
    if (vec_ok) {
        EpiOps o;
        gemm_epilogue_fetch_bias<EPI>(g, col, o);
        gemm_epilogue_fetch_row<EPI>(g, row, col, o);
        gemm_epilogue_apply<EPI, F & ~EPF_LN_MASK>(g, av, gv, o, row, col, crow);
        return;
    }
    if constexpr (EPI == EPI_DSWISH) return;                           // (excluded by the entry point)
    {                                                              // odd leading dims / widths: scalar path
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (col + e >= ncols) continue;
            float x = av[e] + g.bias[col + e];
            const bool drop = !(F & EPF_NO_DROPOUT) && g.drop_p > 0.f && EPI != EPI_GLU && EPI != EPI_RELU;
            const float keep = drop ? dropout_keep(g.drop_seed, (unsigned long long)row * (unsigned long long)g.N +
                                                   (unsigned)(col + e), g.drop_p, 1.0f / (1.0f - g.drop_p)) : 1.0f;
            if (EPI != EPI_SWISH) x *= keep;
            if (EPI == EPI_GLU) x *= sigmoidf_acc(gv[e] + g.bias[g.n_out + col + e]);
            if (EPI == EPI_RESID) x = g.alpha * x + g.R[row * g.ldr + col + e];
            if (EPI == EPI_SWISH && g.Zsave) g.Zsave[row * g.ldc + col + e] = x;
            if (EPI == EPI_SWISH) x = swishf_acc(x) * keep;
            if (EPI == EPI_RELU) x = fmaxf(x, 0.f);
            if ((F & EPF_F32_OUT) || g.c_prec == 0) g.C[row * g.ldc + col + e] = x;
            else if (g.c_prec == CFM_PREC_BF16) reinterpret_cast<__bf16*>(g.C)[row * g.ldc + col + e] = (__bf16)x;
            else reinterpret_cast<_Float16*>(g.C)[row * g.ldc + col + e] = (_Float16)x;
        }
    }
}

__device__ __forceinline__ bool gemm_epilogue_vec_ok(const GemmArgs& g, int epi) {
    const int ncols = epi == EPI_GLU ? g.n_out : g.N;
    return ((g.ldc & 3) == 0) && ((ncols & 3) == 0) && (epi != EPI_RESID || (g.ldr & 3) == 0) &&
           ((reinterpret_cast<uintptr_t>(g.C) & (g.c_prec ? 7 : 15)) == 0) &&
           (epi != EPI_RESID || (reinterpret_cast<uintptr_t>(g.R) & 15) == 0);
}

// Straight from the accumulators: every memory instruction of a wave touches 32 rows x 32 bytes.  Kept for the GLU tile
// (value and gate tiles of one column range live in one wave) and as the fallback for odd leading dimensions.
template <int BM, int BN, int EPI, int TM, int TN, int F = 0>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, const f32x16 (&acc)[TM][TN], int64_t m0, int n0,
                                              int wr, int wc, int li, int hf) {
    static_assert((F & EPF_LN_MASK) == 0 || true, "the LN fold is only wired through gemm_epilogue_rows (entry points guarantee vec_ok)");
    const bool vec_ok = gemm_epilogue_vec_ok(g, EPI);
    constexpr int NTN = EPI == EPI_GLU ? 1 : TN;
    auto col_of = [&](int nt, int q) { return n0 + (EPI == EPI_GLU ? wc * 32 : wc * (BN / 2) + nt * 32) + 8 * q + 4 * hf; };
    auto row_of = [&](int mt) { return m0 + wr * (BM / 2) + mt * 32 + li; };
    auto av_of = [&](int mt, int nt, int q) {
        return f32x4{acc[mt][nt][4 * q], acc[mt][nt][4 * q + 1], acc[mt][nt][4 * q + 2], acc[mt][nt][4 * q + 3]};
    };
    if (vec_ok) {                                                      // (kernel-uniform)
        // every load of the epilogue is issued before its first store (see EpiOps)
        EpiOps ob[NTN][4];
#pragma unroll
        for (int nt = 0; nt < NTN; ++nt)
#pragma unroll
            for (int q = 0; q < 4; ++q) gemm_epilogue_fetch_bias<EPI>(g, col_of(nt, q), ob[nt][q]);
        constexpr bool ROWOPS = EPI == EPI_RESID || EPI == EPI_DSWISH;
        EpiOps orow[ROWOPS ? TM : 1][ROWOPS ? NTN : 1][ROWOPS ? 4 : 1];
        if constexpr (ROWOPS) {
#pragma unroll
            for (int mt = 0; mt < TM; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTN; ++nt)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        orow[mt][nt][q] = ob[nt][q];
                        gemm_epilogue_fetch_row<EPI>(g, row_of(mt), col_of(nt, q), orow[mt][nt][q]);
                    }
        }
#pragma unroll
        for (int mt = 0; mt < TM; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    gemm_epilogue_apply<EPI, F & ~EPF_LN_MASK>(g, av_of(mt, nt, q), av_of(mt, TN - 1, q), ROWOPS ? orow[ROWOPS ? mt : 0][ROWOPS ? nt : 0][ROWOPS ? q : 0] : ob[nt][q],
                                                                row_of(mt), col_of(nt, q), row_of(mt));
        return;
    }
#pragma unroll
    for (int mt = 0; mt < TM; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTN; ++nt)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                gemm_epilogue_at<EPI, F>(g, av_of(mt, nt, q), av_of(mt, TN - 1, q), row_of(mt), col_of(nt, q), false);
}

// ROW-MAJOR epilogue: each 32-row slab of the wave's accumulators goes through a per-wave LDS tile (the K-loop's staging
// buffers are dead by then) and comes back with 8*TN consecutive lanes on one output row, so every bias / residual load
// and every C / Z store of a wave covers whole 128- or 256-byte row segments.  Why: the per-K-tile timeline of the FFN-hidden
// GEMM of the training step (128x128 tiles, Z saved) showed 1.2-1.6 us per K-tile, 10 us for the whole K-loop -- and 17-30 us
// for the epilogue: all co-resident workgroups reach it together and their 32-byte write fragments (96 KB per tile) run at a
// fraction of the HBM write rate.  scratch: per-wave, 32 * (32*TN + 4) floats, 16-byte aligned.
// CLS: the rows are transposed-conv class rows (CONV == 2) and are scattered to their dh1 positions.  A TEMPLATE flag: as a run-time
// test of g.pA the (skipped) 64-bit divisions were still unrolled into every kernel's epilogue -- 60 % more code in the 256x256
// kernels, which ran 12-20 % slower from instruction-cache misses alone.
template <int BM, int BN, int EPI, int TM, int TN, int WM = 2, bool CLS = false, int F = 0>
__device__ __forceinline__ void gemm_epilogue_rows(const GemmArgs& g, const f32x16 (&acc)[TM][TN], int64_t m0, int n0,
                                                   int wr, int wc, int lane, float* scratch, const float* rowstats = nullptr) {
    static_assert(EPI != EPI_GLU || TN == 2, "GLU: n-tile 0 = values, n-tile 1 = gates of the same 32 output columns");
    // row pitch, lanes per row, rows per wave-instruction (GLU: a row of the LDS tile is 32 values | 32 gates -> 32 output columns)
    constexpr int P = 32 * TN + 4, LPR = EPI == EPI_GLU ? 8 : 8 * TN, RPI = 64 / LPR;
    const int li = lane & 31, hf = lane >> 5;
    if constexpr (WM == 2) {                                          // (8-wave tiles: the launcher guarantees vec_ok)
        if (!gemm_epilogue_vec_ok(g, EPI)) {                          // (kernel-uniform)
            gemm_epilogue<BM, BN, EPI, TM, TN, F>(g, acc, m0, n0, wr, wc, li, hf);
            return;
        }
    }
    constexpr bool ROWOPS = EPI == EPI_RESID || EPI == EPI_DSWISH;
    auto stage_slab = [&](int mt) {                                   // accumulators of 32 rows -> the wave's LDS tile
#pragma unroll
        for (int nt = 0; nt < TN; ++nt)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<f32x4*>(scratch + li * P + 32 * nt + 8 * q + 4 * hf) =
                    f32x4{acc[mt][nt][4 * q], acc[mt][nt][4 * q + 1], acc[mt][nt][4 * q + 2], acc[mt][nt][4 * q + 3]};
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto slab_done = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto class_row = [&](int64_t row) {                               // CLS: class row (b, a, c) -> dh1 position (b, 2a+pt, 2c+pf)
        const int64_t rr = min(row, g.M - 1);
        const int per = g.pA * g.pC;
        const int b = (int)(rr / per), r = (int)(rr - (int64_t)b * per);
        const int a = r / g.pC, c = r - a * g.pC;
        return ((int64_t)b * g.cT1 + 2 * a + g.pt) * g.cF1 + 2 * c + g.pf;
    };

    // ---- 16-bit C: 8 columns per lane, one 16-byte store per lane and output tensor (see gemm_epilogue_apply8)
    if constexpr (EPI != EPI_RESID && EPI != EPI_GLU && (F & (EPF_LN_MASK | EPF_F32_OUT)) == 0) {
        const bool wide = g.c_prec != 0 && ((g.N | (int)g.ldc) & 7) == 0 && (reinterpret_cast<uintptr_t>(g.C) & 15) == 0 &&
                          (EPI != EPI_SWISH || !g.Zsave || (reinterpret_cast<uintptr_t>(g.Zsave) & 15) == 0) &&
                          (EPI != EPI_DSWISH || (g.z_prec != 0 && (g.ldr & 7) == 0 && (reinterpret_cast<uintptr_t>(g.Zsave) & 15) == 0));
        if (wide) {                                                    // (kernel-uniform)
            constexpr int LPR8 = 4 * TN, RPI8 = 64 / LPR8, NIT8 = 32 / RPI8;
            const int rsub8 = lane / LPR8, c8 = (lane % LPR8) * 8;
            const int col8 = n0 + wc * (BN / 2) + c8;
            auto row8 = [&](int mt, int it) { return m0 + wr * (BM / WM) + mt * 32 + it * RPI8 + rsub8; };
            EpiOps b0, b1;
            gemm_epilogue_fetch_bias<EPI>(g, col8, b0);
            gemm_epilogue_fetch_bias<EPI>(g, col8 + 4, b1);
            uint4 zcur[EPI == EPI_DSWISH ? NIT8 : 1], znxt[EPI == EPI_DSWISH ? NIT8 : 1];
            auto fetch_z = [&](uint4* dst, int mt) {                   // eight 16-bit Z values per group
#pragma unroll
                for (int it = 0; it < NIT8; ++it) {
                    const int64_t rc = min(row8(mt, it), g.M - 1);
                    dst[it] = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(g.Zsave) + rc * g.ldr + (col8 < g.N ? col8 : 0));
                }
            };
            if constexpr (EPI == EPI_DSWISH) fetch_z(zcur, 0);
#pragma unroll
            for (int mt = 0; mt < TM; ++mt) {
                stage_slab(mt);
                if constexpr (EPI == EPI_DSWISH) {
                    if (mt + 1 < TM) fetch_z(znxt, mt + 1);
#pragma unroll
                    for (int it = 0; it < NIT8; ++it) {
                        const int rl = it * RPI8 + rsub8;
                        const f32x4 v0 = *reinterpret_cast<const f32x4*>(scratch + rl * P + c8);
                        const f32x4 v1 = *reinterpret_cast<const f32x4*>(scratch + rl * P + c8 + 4);
                        EpiOps o0 = b0, o1 = b1;
                        o0.z16[0] = zcur[it].x; o0.z16[1] = zcur[it].y; o1.z16[0] = zcur[it].z; o1.z16[1] = zcur[it].w;
                        const int64_t row = row8(mt, it);
                        gemm_epilogue_apply8<EPI>(g, v0, v1, o0, o1, row, col8, CLS ? class_row(row) : row);
                    }
#pragma unroll
                    for (int it = 0; it < NIT8; ++it) zcur[it] = znxt[it];
                } else {
#pragma unroll 2
                    for (int it = 0; it < NIT8; ++it) {
                        const int rl = it * RPI8 + rsub8;
                        const f32x4 v0 = *reinterpret_cast<const f32x4*>(scratch + rl * P + c8);
                        const f32x4 v1 = *reinterpret_cast<const f32x4*>(scratch + rl * P + c8 + 4);
                        const int64_t row = row8(mt, it);
                        gemm_epilogue_apply8<EPI>(g, v0, v1, b0, b1, row, col8, CLS ? class_row(row) : row);
                    }
                }
                slab_done();
            }
            return;
        }
    }

    // ---- 4 columns per lane
    const int rsub = lane / LPR, c4 = (lane % LPR) * 4;
    const int col = n0 + (EPI == EPI_GLU ? wc * 32 : wc * (BN / 2)) + c4;
    constexpr int NIT = 32 / RPI;
    auto row_of = [&](int mt, int it) { return m0 + wr * (BM / WM) + mt * 32 + it * RPI + rsub; };
    // Every load is issued ahead of the stores it would otherwise queue behind (see EpiOps): the bias once (a lane's columns
    // are the same for all its rows), the residual / Z rows of batch i+1 before batch i is written out.
    EpiOps ob;
    gemm_epilogue_fetch_bias<EPI, F>(g, col, ob);
    if constexpr (!ROWOPS) {
#pragma unroll
        for (int mt = 0; mt < TM; ++mt) {
            stage_slab(mt);
#pragma unroll 2
            for (int it = 0; it < NIT; ++it) {                         // (a real loop: the unrolled form was 12 000 instructions)
                const int rl = it * RPI + rsub;
                const f32x4 v = *reinterpret_cast<const f32x4*>(scratch + rl * P + c4);
                const f32x4 gate = EPI == EPI_GLU ? *reinterpret_cast<const f32x4*>(scratch + rl * P + 32 + c4) : v;
                const int64_t row = row_of(mt, it);
                gemm_epilogue_apply<EPI, F>(g, v, gate, ob, row, col, CLS ? class_row(row) : row, rowstats, (int)(row - m0));
            }
            slab_done();
        }
    } else {
        constexpr int HB = NIT < 8 ? NIT : 8, NH = NIT / HB, NBATCH = TM * NH;   // row operands travel in batches of <= 8 groups (32 registers)
        static_assert(NIT % HB == 0, "4, 8 or 16 row groups per 32-row slab");
        EpiOps ocur[HB], onxt[HB];
        auto fetch_batch = [&](EpiOps* dst, int bidx) {
            const int mt = bidx / NH, h = bidx % NH;
#pragma unroll
            for (int j = 0; j < HB; ++j) gemm_epilogue_fetch_row<EPI>(g, row_of(mt, h * HB + j), col, dst[j]);
        };
        fetch_batch(ocur, 0);
#pragma unroll
        for (int mt = 0; mt < TM; ++mt) {
            stage_slab(mt);
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                const int bidx = mt * NH + h;
                if (bidx + 1 < NBATCH) fetch_batch(onxt, bidx + 1);
#pragma unroll
                for (int j = 0; j < HB; ++j) {
                    const int it = h * HB + j;
                    const int rl = it * RPI + rsub;
                    const f32x4 v = *reinterpret_cast<const f32x4*>(scratch + rl * P + c4);
                    const int64_t row = row_of(mt, it);
                    EpiOps o = ob;
                    o.rr = ocur[j].rr; o.z16[0] = ocur[j].z16[0]; o.z16[1] = ocur[j].z16[1];
                    gemm_epilogue_apply<EPI, F>(g, v, v, o, row, col, CLS ? class_row(row) : row, rowstats, (int)(row - m0));
                }
#pragma unroll
                for (int j = 0; j < HB; ++j) ocur[j] = onxt[j];
            }
            slab_done();
        }
    }
}

}  // namespace
