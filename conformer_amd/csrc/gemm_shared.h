// Pieces shared by the fp32-MFMA GEMM (gemm_f32.hip) and the bf16-MFMA GEMM (gemm_bf16.hip): argument block, operand
// row addressing (plain / GLU row interleave / implicit-GEMM conv gather) and the 16-byte epilogue.
#pragma once
#include "cfm_common.h"

namespace {

enum Epi { EPI_BIAS = 0, EPI_SWISH = 1, EPI_RELU = 2, EPI_GLU = 3, EPI_RESID = 4 };

struct GemmArgs {
    const float* A; const float* W; const float* bias; const float* R; float* C;
    float* Zsave;                   // swish epilogue, training: also store the pre-activation (same ldc), or NULL
    float drop_p; unsigned long long drop_seed;   // training: dropout on the GEMM result (after Swish; before alpha*y+R)
    int64_t M; int N; int K; int64_t lda, ldr, ldc; float alpha;
    int n_out;                      // GLU: output columns (N = 2*n_out)
    int cT2, cF2, cT1, cF1, cC;     // conv mode geometry
    unsigned tiles_m, tiles_n;
    int occ_cap;                    // 0 = natural; else blocks/CU cap enforced through a dynamic-LDS pad
    int c_prec;                     // 0: C is fp32; CFM_PREC_BF16 / CFM_PREC_FP16: C is stored in that 16-bit type (ldc in elements)
    unsigned long long* trace;      // diagnostics: per-block {start, end} s_memrealtime stamps + HW id, or NULL
};

// ---- shared pieces --------------------------------------------------------------------------------------------
template <bool CONV>
__device__ __forceinline__ const float* a_row_ptr(const GemmArgs& g, int64_t m) {
    if (m >= g.M) m = g.M - 1;                           // clamp: the row is loaded but never stored
    if (CONV) {
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

template <bool CONV>
__device__ __forceinline__ int64_t a_k_offset(const GemmArgs& g, int k) {   // k = first K index of an aligned slab
    if (CONV) {
        const int tap = k / g.cC, ci = k - tap * g.cC;
        const int kf = tap / 3, ktp = tap - 3 * kf;
        return ((int64_t)ktp * g.cF1 + kf) * g.cC + ci;
    }
    return k;
}

// The MFMAs are issued with W fragments as the A operand and activation fragments as the B operand, so an accumulator
// tile is C^T: lane (li, hf) holds output ROW m = tile_row + li and, in registers 4q..4q+3, the four CONSECUTIVE
// columns n = tile_col + 8q + 4hf + {0,1,2,3}.  The epilogue therefore moves 16 bytes per lane per instruction
// (bias / residual loads and the C store): 4x fewer memory instructions than a dword-per-lane epilogue -- the
// epilogue of a short-K GEMM is store-ISSUE bound (17-22 us of a 160 us FFN GEMM before this change; per-block
// timeline in profiles/r01_gemm_timeline.txt).
template <int BM, int BN, int EPI, int TM, int TN>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, const f32x16 (&acc)[TM][TN], int64_t m0, int n0,
                                              int wr, int wc, int li, int hf) {
    const int ncols = EPI == EPI_GLU ? g.n_out : g.N;
    const bool vec_ok = ((g.ldc & 3) == 0) && ((ncols & 3) == 0) && (EPI != EPI_RESID || (g.ldr & 3) == 0) &&
                        ((reinterpret_cast<uintptr_t>(g.C) & (g.c_prec ? 7 : 15)) == 0) &&
                        (EPI != EPI_RESID || (reinterpret_cast<uintptr_t>(g.R) & 15) == 0);
#pragma unroll
    for (int mt = 0; mt < TM; ++mt) {
        const int64_t row = m0 + wr * (BM / 2) + mt * 32 + li;
        if (row >= g.M) continue;
#pragma unroll
        for (int nt = 0; nt < (EPI == EPI_GLU ? 1 : TN); ++nt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = n0 + (EPI == EPI_GLU ? wc * 32 : wc * (BN / 2) + nt * 32) + 8 * q + 4 * hf;
                if (col >= ncols) continue;
                float v[4];
                if (vec_ok) {                                                  // col + 3 < ncols because ncols % 4 == 0
                    const f32x4 bb = *reinterpret_cast<const f32x4*>(g.bias + col);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[mt][nt][4 * q + e] + bb[e];
                    const bool drop = g.drop_p > 0.f && EPI != EPI_GLU && EPI != EPI_RELU;
                    const float inv_keep = drop ? 1.0f / (1.0f - g.drop_p) : 1.0f;
                    const unsigned long long e0 = (unsigned long long)row * (unsigned long long)g.N + (unsigned)col;
                    if (drop && EPI != EPI_SWISH) {                        // dropout(y) then alpha*y + R  (or plain y)
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] *= dropout_keep(g.drop_seed, e0 + e, g.drop_p, inv_keep);
                    }
                    if (EPI == EPI_GLU) {
                        const f32x4 bg = *reinterpret_cast<const f32x4*>(g.bias + g.n_out + col);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] *= sigmoidf_acc(acc[mt][TN - 1][4 * q + e] + bg[e]);
                    }
                    if (EPI == EPI_RESID) {
                        const f32x4 rr = *reinterpret_cast<const f32x4*>(g.R + row * g.ldr + col);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = g.alpha * v[e] + rr[e];
                    }
                    if (EPI == EPI_SWISH && g.Zsave)
                        *reinterpret_cast<f32x4*>(g.Zsave + row * g.ldc + col) = f32x4{v[0], v[1], v[2], v[3]};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (EPI == EPI_SWISH) {
                            v[e] = swishf_acc(v[e]);
                            if (drop) v[e] *= dropout_keep(g.drop_seed, e0 + e, g.drop_p, inv_keep);
                        }
                        if (EPI == EPI_RELU) v[e] = fmaxf(v[e], 0.f);
                    }
                    if (g.c_prec == 0) *reinterpret_cast<f32x4*>(g.C + row * g.ldc + col) = f32x4{v[0], v[1], v[2], v[3]};
                    else if (g.c_prec == CFM_PREC_BF16)
                        *reinterpret_cast<Lowp<__bf16>::x4*>(reinterpret_cast<__bf16*>(g.C) + row * g.ldc + col) =
                            Lowp<__bf16>::cvt4(f32x4{v[0], v[1], v[2], v[3]});
                    else
                        *reinterpret_cast<Lowp<_Float16>::x4*>(reinterpret_cast<_Float16*>(g.C) + row * g.ldc + col) =
                            Lowp<_Float16>::cvt4(f32x4{v[0], v[1], v[2], v[3]});
                } else {                                                       // odd leading dims / widths: scalar path
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (col + e >= ncols) continue;
                        float x = acc[mt][nt][4 * q + e] + g.bias[col + e];
                        const bool drop = g.drop_p > 0.f && EPI != EPI_GLU && EPI != EPI_RELU;
                        const float keep = drop ? dropout_keep(g.drop_seed, (unsigned long long)row * (unsigned long long)g.N +
                                                               (unsigned)(col + e), g.drop_p, 1.0f / (1.0f - g.drop_p)) : 1.0f;
                        if (EPI != EPI_SWISH) x *= keep;
                        if (EPI == EPI_GLU) x *= sigmoidf_acc(acc[mt][TN - 1][4 * q + e] + g.bias[g.n_out + col + e]);
                        if (EPI == EPI_RESID) x = g.alpha * x + g.R[row * g.ldr + col + e];
                        if (EPI == EPI_SWISH && g.Zsave) g.Zsave[row * g.ldc + col + e] = x;
                        if (EPI == EPI_SWISH) x = swishf_acc(x) * keep;
                        if (EPI == EPI_RELU) x = fmaxf(x, 0.f);
                        if (g.c_prec == 0) g.C[row * g.ldc + col + e] = x;
                        else if (g.c_prec == CFM_PREC_BF16) reinterpret_cast<__bf16*>(g.C)[row * g.ldc + col + e] = (__bf16)x;
                        else reinterpret_cast<_Float16*>(g.C)[row * g.ldc + col + e] = (_Float16)x;
                    }
                }
            }
    }
}

}  // namespace
