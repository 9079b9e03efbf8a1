// Argument block shared by the fp32 (gemm_bwd_f32.hip) and 16-bit-MFMA (gemm_bwd_mfma16.hip) backward GEMM kernels.
#pragma once
#include "cfm_common.h"

enum BwdEpi { BEPI_SCALE = 0, BEPI_DSWISH = 1 };

int cfm_bwd_debug_tile();   // diagnostics (cfm_debug_set_bwd_tile): -1 = heuristic, 0 = 128x128, 1 = 128x64, 3 = 64x64

struct BwdArgs {
    const float* A; const float* B; const float* Z; float* C;
    int I, J; int64_t Kc; int64_t lda, ldb, ldz, ldc; float alpha;
    unsigned tiles_i, tiles_j; int splits; int64_t k_per_split;
    float drop_p; unsigned long long drop_seed;   // DSWISH: the forward dropped swish(Z): re-apply its mask to the incoming gradient
    int accumulate;                      // C += ... (non-atomic read-modify-write; not combined with splits)
    int b16;                             // 16-bit kernels: operand B is already stored in the 16-bit type (ldb/sb* in elements)
    int z16;                             // 16-bit kernels: Z is stored in the 16-bit type (1 bf16 | 2 fp16; ldz in elements)
    int c16;                             // 16-bit kernels, DSWISH only: C is stored in the 16-bit type (1 bf16 | 2 fp16; ldc in elements)
    int pad4;                            // 16-bit kernels: ragged Kc / I / J are physically padded to a multiple of 4 with zeros
    int nbatch, nb1;                     // batched: blockIdx.z = b0*nb1 + b1; operand offset = b0*s?0 + b1*s?1
    int64_t sa0, sa1, sb0, sb1, sc0, sc1;
    // stem-convolution gathers (GATHER != 0): geometry of the 3x3 / stride-2 conv over the channel-last h1
    int cT1, cF1, cT2, cF2, cC;          // h1: (B,T1,F1,C); h2: (B,T2,F2,C)
    int pA, pC;                          // GATHER 2: class grid (rows per utterance = pA*pC: a < pA, c < pC)
    int pt, pf;                          // GATHER 2: parity class: t1 = 2a+pt, f1 = 2c+pf
    int tap_dt[4], tap_df[4];            // GATHER 2: per class tap: t2 = a + dt, f2 = c + df
};

// Epilogue for accumulators held transposed (lane (li,hf) = row li of the 32-row tile, registers 4q..4q+3 = columns
// 8q + 4hf + {0..3}): scale, optional swish'(Z) (+ the forward's dropout mask), optional scatter of stem class rows,
// 16-byte stores (or read-modify-write when accumulating).
template <int BM, int BN, int EPI, int GATHER, int TM, int TN>
__device__ __forceinline__ void bwd_epilogue_rows(const BwdArgs& g, float* Cb, f32x16 (&acc)[TM][TN], int i0, int j0, int wr,
                                                  int wc, int li, int hf) {
#pragma unroll
    for (int mt = 0; mt < TM; ++mt) {
        const int row = i0 + wr * (BM / 2) + mt * 32 + li;
        if (row >= g.I) continue;
        int64_t crow = row;
        if (GATHER == 2) {                                                 // class row -> position (b, 2a+pt, 2c+pf) of dh1
            const int per = g.pA * g.pC;
            const int b = row / per, r = row - b * per;
            const int a = r / g.pC, c = r - a * g.pC;
            crow = ((int64_t)b * g.cT1 + 2 * a + g.pt) * g.cF1 + 2 * c + g.pf;
        }
        // every load of the slab (Z of swish', the accumulate operand) is issued before its first store: loaded group by group
        // they made the epilogue a chain of store-acknowledge -> load round trips (cf. EpiOps in gemm_shared.h)
        f32x4 zq[EPI == BEPI_DSWISH ? TN : 1][4], oq[TN][4];
#pragma unroll
        for (int nt = 0; nt < TN; ++nt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = j0 + wc * (BN / 2) + nt * 32 + 8 * q + 4 * hf;
                const bool full = col + 3 < g.J;
                if (EPI == BEPI_DSWISH) zq[EPI == BEPI_DSWISH ? nt : 0][q] = full ? *reinterpret_cast<const f32x4*>(g.Z + (int64_t)row * g.ldz + col) : f32x4{0.f, 0.f, 0.f, 0.f};
                oq[nt][q] = (full && g.accumulate) ? *reinterpret_cast<const f32x4*>(Cb + crow * g.ldc + col) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
        for (int nt = 0; nt < TN; ++nt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = j0 + wc * (BN / 2) + nt * 32 + 8 * q + 4 * hf;
                if (col >= g.J) continue;
                const bool full = col + 3 < g.J;                           // ragged last columns: element-wise
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = g.alpha * acc[mt][nt][4 * q + e];
                if (EPI == BEPI_DSWISH) {
                    const float* zp = g.Z + (int64_t)row * g.ldz + col;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (!full && col + e >= g.J) continue;
                        const float z = full ? zq[EPI == BEPI_DSWISH ? nt : 0][q][e] : zp[e];
                        const float sg = sigmoidf_acc(z);
                        v[e] *= sg * (1.0f + z * (1.0f - sg));
                        if (g.drop_p > 0.f)
                            v[e] *= dropout_keep(g.drop_seed, (unsigned long long)row * (unsigned long long)g.J + (unsigned)(col + e),
                                                 g.drop_p, 1.0f / (1.0f - g.drop_p));
                    }
                }
                float* dst = Cb + crow * g.ldc + col;
                if (full) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += oq[nt][q][e];      // (zeros unless accumulating)
                    *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (col + e >= g.J) continue;
                        dst[e] = g.accumulate ? dst[e] + v[e] : v[e];
                    }
                }
            }
    }
}

// The same through a per-wave LDS tile (cf. gemm_epilogue_rows in gemm_shared.h): each 32-row slab of the wave's accumulators
// is written to `scratch` (32 * (32*TN + 4) floats per wave; the K-loop's staging buffers, dead after its last barrier) and
// read back with 8*TN consecutive lanes on one output row, so the Z loads (swish'), the accumulate loads and the C stores of
// a wave cover whole 128 / 256-byte row segments instead of 32 rows x 32 bytes.  Requires J % 4 == 0, ldc % 4 == 0 (and
// ldz % 4 == 0) and 16-byte aligned bases -- bwd_rows_lds_ok(); otherwise the caller uses bwd_epilogue_rows.
__device__ __forceinline__ bool bwd_rows_lds_ok(const BwdArgs& g, const float* Cb, int epi) {
    return (g.J & 3) == 0 && (g.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(Cb) & 15) == 0 &&
           (epi != BEPI_DSWISH || ((g.ldz & 3) == 0 && (reinterpret_cast<uintptr_t>(g.Z) & (g.z16 ? 7 : 15)) == 0));
}

template <int BM, int BN, int EPI, int GATHER, int TM, int TN>
__device__ __forceinline__ void bwd_epilogue_rows_lds(const BwdArgs& g, float* Cb, f32x16 (&acc)[TM][TN], int i0, int j0, int wr,
                                                      int wc, int lane, float* scratch) {
    constexpr int P = 32 * TN + 4, LPR = 8 * TN, RPI = 64 / LPR;
    const int li = lane & 31, hf = lane >> 5;
    const int rsub = lane / LPR, c4 = (lane % LPR) * 4;
    const int col = j0 + wc * (BN / 2) + c4;
#pragma unroll
    for (int mt = 0; mt < TM; ++mt) {
#pragma unroll
        for (int nt = 0; nt < TN; ++nt)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<f32x4*>(scratch + li * P + 32 * nt + 8 * q + 4 * hf) =
                    f32x4{acc[mt][nt][4 * q], acc[mt][nt][4 * q + 1], acc[mt][nt][4 * q + 2], acc[mt][nt][4 * q + 3]};
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // the Z / accumulate loads of up to 8 row groups are issued together, ahead of the groups' stores (cf. EpiOps in gemm_shared.h)
        constexpr int NIT = 32 / RPI, HB = NIT < 8 ? NIT : 8;
#pragma unroll
        for (int h = 0; h < NIT / HB; ++h) {
            f32x4 zf[EPI == BEPI_DSWISH ? HB : 1], of[HB];
            uint2 zh[EPI == BEPI_DSWISH ? HB : 1];
#pragma unroll
            for (int j = 0; j < HB; ++j) {
                const int rl = (h * HB + j) * RPI + rsub;
                const int row = min(i0 + wr * (BM / 2) + mt * 32 + rl, g.I - 1);
                const int cc = col < g.J ? col : 0;
                if (EPI == BEPI_DSWISH) {
                    if (g.z16 == 0) zf[EPI == BEPI_DSWISH ? j : 0] = *reinterpret_cast<const f32x4*>(g.Z + (int64_t)row * g.ldz + cc);
                    else zh[EPI == BEPI_DSWISH ? j : 0] = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(g.Z) + ((int64_t)row * g.ldz + cc) * 2);
                }
                if (g.accumulate && !g.c16) {
                    int64_t crow = row;
                    if (GATHER == 2) {
                        const int per = g.pA * g.pC;
                        const int b = row / per, r = row - b * per;
                        const int a = r / g.pC, c = r - a * g.pC;
                        crow = ((int64_t)b * g.cT1 + 2 * a + g.pt) * g.cF1 + 2 * c + g.pf;
                    }
                    of[j] = *reinterpret_cast<const f32x4*>(Cb + crow * g.ldc + cc);
                } else of[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int j = 0; j < HB; ++j) {
                const int rl = (h * HB + j) * RPI + rsub;
                const int row = i0 + wr * (BM / 2) + mt * 32 + rl;
                f32x4 v = *reinterpret_cast<const f32x4*>(scratch + rl * P + c4);
                if (row >= g.I || col >= g.J) continue;
                v = v * g.alpha;
                if (EPI == BEPI_DSWISH) {
                    f32x4 z4;
                    if (g.z16 == 0) z4 = zf[EPI == BEPI_DSWISH ? j : 0];
                    else {
                        const uint2 t = zh[EPI == BEPI_DSWISH ? j : 0];
                        if (g.z16 == 1) z4 = f32x4{__uint_as_float(t.x << 16), __uint_as_float(t.x & 0xffff0000u), __uint_as_float(t.y << 16), __uint_as_float(t.y & 0xffff0000u)};
                        else {
                            typedef _Float16 hf4 __attribute__((ext_vector_type(4)));
                            hf4 th;
                            __builtin_memcpy(&th, &t, 8);
                            z4 = f32x4{(float)th[0], (float)th[1], (float)th[2], (float)th[3]};
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float sg = sigmoidf_acc(z4[e]);
                        v[e] *= sg * (1.0f + z4[e] * (1.0f - sg));
                    }
                    if (g.drop_p > 0.f) {                                  // (J % 4 == 0, col % 4 == 0: an aligned group, one hash)
                        float keep[4];
                        dropout_keep4(g.drop_seed, (unsigned long long)row * (unsigned long long)g.J + (unsigned)col, g.drop_p,
                                      1.0f / (1.0f - g.drop_p), keep);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] *= keep[e];
                    }
                }
                int64_t crow = row;
                if (GATHER == 2) {                                         // class row -> position (b, 2a+pt, 2c+pf) of dh1
                    const int per = g.pA * g.pC;
                    const int b = row / per, r = row - b * per;
                    const int a = r / g.pC, c = r - a * g.pC;
                    crow = ((int64_t)b * g.cT1 + 2 * a + g.pt) * g.cF1 + 2 * c + g.pf;
                }
                if (g.c16) {                                               // gradient stored for GEMM consumers only: 8-byte stores
                    typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
                    typedef _Float16 hf4 __attribute__((ext_vector_type(4)));
                    void* d16 = reinterpret_cast<char*>(Cb) + (crow * g.ldc + col) * 2;
                    if (g.c16 == 1) *reinterpret_cast<bf4*>(d16) = bf4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                    else *reinterpret_cast<hf4*>(d16) = hf4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                    continue;
                }
                v = v + of[j];                                             // (zeros unless accumulating)
                *reinterpret_cast<f32x4*>(Cb + crow * g.ldc + col) = v;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// Split-K epilogue for accumulators in the NATURAL MFMA orientation (lane li = column, register r = row
// (r&3) + 8*(r>>2) + 4*hf): one atomic instruction adds two contiguous 128-byte row segments.
template <int BM, int BN, int TM, int TN>
__device__ __forceinline__ void bwd_epilogue_atomic(const BwdArgs& g, float* Cb, f32x16 (&acc)[TM][TN], int i0, int j0, int wr,
                                                    int wc, int li, int hf) {
#pragma unroll
    for (int mt = 0; mt < TM; ++mt)
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) {
            const int col = j0 + wc * (BN / 2) + nt * 32 + li;
            if (col >= g.J) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i0 + wr * (BM / 2) + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf;
                if (row < g.I) atomicAdd(Cb + (int64_t)row * g.ldc + col, g.alpha * acc[mt][nt][r]);
            }
        }
}
