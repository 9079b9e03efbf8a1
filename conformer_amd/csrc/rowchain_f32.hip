// Row-local chains of the Conformer block, fp32 inference: everything between two operators that mix rows (attention, the
// depthwise convolution) acts on each (utterance, frame) row by itself -- Linear layers, LayerNorms, residual adds, Swish, GLU --
// so ONE workgroup can take 32 rows through the whole chain with the rows resident in LDS and nothing but weights streaming in:
//     K1 (block.py:19-21)  x -> FFN1 (ffn.py:15-23, + x/2 residual) -> y;   LN (attention.py:15) + q|k|v projection (78-80) -> qkv
//     K2 (block.py:21-23)  ctx -> out_proj (attention.py:90) + y -> y2;      LN (convolution.py:22) + pointwise_conv_1 + GLU (24-25) -> g
//     K3 (block.py:23-27)  c -> pointwise_conv_2 (convolution.py:29) + y2 -> y3;  FFN2 (+ y3/2) -> LayerNorm (block.py:27) -> out
// Stages (template flags): PRE = a d x d Linear + bias + residual whose result becomes the resident rows (statistics computed
// in the kernel); CORE = the fused feed-forward of ffn_fused_f32.hip (same loops: hidden activation in registers, the four
// waves split the hidden units and their partial output tiles are summed through LDS in a fixed order); the finish of the rows
// (plain | + statistics partials | + closing LayerNorm); POST = a d x N Linear with the LayerNorm in front of it folded
// (rstd (acc - mean colsum) + bias) and optionally GLU, stored row-major through a wave-private LDS staging tile.
// Why: a stand-alone row-block GEMM pays ~5 us to bring its 32 x d rows in (all 249 workgroups at once: a 16 MB burst) and ~5 us
// to finish / store them, whatever its size; the classic tiled kernels reach 0.62-0.72 of the matrix pipe at these K = 512
// shapes (pointwise_conv_2 / out_proj 43 us, q|k|v 110 us, pointwise_conv_1 77 us).  Inside a chain the same products run at the
// streaming rate of the feed-forward loop (~71 clocks per MFMA) and the bursts of the stages in between disappear.
// All weights are pre-packed in MFMA A-fragment order per wave (cfm_rowgemm_pack_f32 / cfm_ffn_pack_f32) and go L2 -> registers.
#include <type_traits>

#include "cfm_common.h"

namespace {

struct ChainArgs {
    const float* X; int64_t ldx;                 // PRE: the A rows of the pre-Linear (conv / attention output); else the residual-stream rows
    // PRE: Y1 = X.Wpre^T + bpre + R -> resident rows; stored to Y1 when not NULL
    const float* Wpre; const float* bpre; const float* R; int64_t ldr; float* Y1; int64_t ldy1;
    // statistics partials of the resident rows when PRE == 0 (cfm_gemm_lnfold_f32 format)
    const float* ln_stats; int ln_parts;
    float ln_eps;                                // eps of the LayerNorm folded into the first consumer of the resident rows (CORE, else POST)
    // CORE (cfm_ffn_fused_f32 arguments)
    const float* Wp; const float* b1f; const float* cs1; const float* b2; float alpha; int hidden; int64_t tile_stride; int rotate;
    // finish of the CORE rows
    float* Y; int64_t ldy; float* stats_out; const float* gamma2; const float* beta2; float eps2;
    // POST: Z = epi(LN(rows).Wpost^T + bpost), LN folded (cspost = column sums of the folded weight); rows = CORE result (or Y1)
    const float* Wpost; const float* bpost; const float* cspost; float post_eps; float* Z; int64_t ldz;
    int64_t M;
};

enum { CH_PLAIN = 0, CH_STATS = 1, CH_LN = 2 };          // finish of the CORE rows (as cfm_ffn_fused_f32's mode)
enum { POST_NONE = 0, POST_BIAS3 = 1, POST_GLU = 2 };     // POST_BIAS3: N = 3 d (fused q|k|v); POST_GLU: weight (2 d, d), output d

template <int ND, int PRE, int CORE, int POST, int MODE>
__global__ __launch_bounds__(256, 1) void rowchain_f32_kernel(const ChainArgs a) {
    constexpr int D = 32 * ND, XS = D + 4, NL = 4 * ND, RING = 16;
    constexpr int TPW = ND / 4, QW = 32 * TPW, EW = QW + 4, ERS = 32 * EW;   // tiles per wave of a d-wide product; its columns; LDS region
    static_assert(NL % RING == 0 && ND % 4 == 0, "d = 128, 256 or 512");
    static_assert(CORE || (PRE && POST), "a chain without the feed-forward is PRE + POST");
    __shared__ __attribute__((aligned(16))) float smem[32 * XS + 4 * ERS + 64];
    float* Xs = smem;                      // [32][XS]      the resident rows
    float* Ys = smem + 32 * XS;            // [4][32][EW]   residual staging (PRE) / tile-sum regions (CORE) / store staging (POST)
    float* St = Ys + 4 * ERS;              // [32][2]       (mean, rstd) of the resident rows, computed in the kernel

    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, hf = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t m0 = (int64_t)blockIdx.x * 32;
    constexpr int V = D / 4;                                            // float4 per row

    f32x16 y[ND];                                                       // accumulator tiles, shared by the stages
    const float* xrow = Xs + li * XS + 4 * hf;
    float mean = 0.f, rstd = 0.f;

    // ---- d-input product of the resident rows with a weight packed per wave as [c][t] (c = 8-dim chunk, t = the wave's tile):
    // one X fragment (LDS) per chunk serves all NT tiles; NT weight loads + 4 NT MFMAs per chunk; loads run RS / NT chunks ahead
    auto gemm_rows = [&](const f32x4* ws, auto nt_tag) {
        constexpr int NT = decltype(nt_tag)::value;
        constexpr int CB = NT > 12 ? 1 : NT > 4 ? 2 : NT > 2 ? 4 : NT > 1 ? 8 : 16, RS = CB * NT;   // chunks per loop body, ring size: 16 .. 24 loads in flight
        // (8 .. 12 were enough with the weights warm in L2, not inside a forward where every block's weights arrive from HBM: the
        //  out_proj + GLU chain took 140 us there against 121 us stand-alone)
        static_assert(NL % CB == 0, "");
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) y[t][r] = 0.f;
        f32x4 rg[RS];
#pragma unroll
        for (int i = 0; i < RS; ++i) rg[i] = ws[i * 64];
        constexpr int TOTAL = NL * NT;
        f32x4 xv = *reinterpret_cast<const f32x4*>(xrow);
#pragma unroll 1
        for (int c0 = 0; c0 < NL; c0 += CB) {
            const f32x4* wn = ws + (int64_t)min((c0 + CB) * NT, TOTAL - RS) * 64;   // the body's refills (last body: re-reads, unused)
            const float* xc = xrow + 8 * c0;
#pragma unroll
            for (int cc = 0; cc < CB; ++cc) {
                const f32x4 xn = *reinterpret_cast<const f32x4*>(xc + 8 * (cc + 1));   // (the last one reads 16 bytes past the row: unused)
#pragma unroll
                for (int t = 0; t < NT; t += (NT % 2 == 0 ? 2 : 1)) {
                    const int i = cc * NT + t;
                    if constexpr (NT % 2 == 0) {
                        const f32x4 w0 = rg[i], w1 = rg[i + 1];
                        rg[i] = wn[i * 64];
                        rg[i + 1] = wn[(i + 1) * 64];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            y[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0[e], xv[e], y[t], 0, 0, 0);
                            y[t + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[e], xv[e], y[t + 1], 0, 0, 0);
                        }
                    } else {
                        const f32x4 w0 = rg[i];
                        rg[i] = wn[i * 64];
#pragma unroll
                        for (int e = 0; e < 4; ++e) y[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0[e], xv[e], y[t], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                xv = xn;
            }
        }
    };
    // ---- (mean, rstd) of the resident rows from LDS: wave w takes rows 8 w .. 8 w + 7 (two-pass, row-major), all lanes then read
    // their row's pair from St.  store_to: also write the rows to global memory (PRE result that later kernels need), or NULL.
    auto row_stats = [&](float eps, float* store_to, int64_t ld) {
#pragma unroll 2
        for (int rr = 0; rr < 8; ++rr) {
            const int r = 8 * wave + rr;
            f32x4 v[(V + 63) / 64];
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < (V + 63) / 64; ++i) {
                const int c4 = lane + 64 * i;
                v[i] = c4 < V ? *reinterpret_cast<const f32x4*>(Xs + r * XS + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f};
                s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
                if (store_to && c4 < V && m0 + r < a.M) *reinterpret_cast<f32x4*>(store_to + (m0 + r) * ld + 4 * c4) = v[i];
            }
            const float mu = wave_sum(s) * (1.0f / (float)D);
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < (V + 63) / 64; ++i) {
                if (lane + 64 * i < V) {
                    const f32x4 dv = v[i] - mu;
                    q += (dv[0] * dv[0] + dv[1] * dv[1]) + (dv[2] * dv[2] + dv[3] * dv[3]);
                }
            }
            q = wave_sum(q);
            if (lane == 0) { St[2 * r] = mu; St[2 * r + 1] = 1.0f / sqrtf(q * (1.0f / (float)D) + eps); }
        }
        __syncthreads();
        mean = St[2 * li]; rstd = St[2 * li + 1];
    };

    // =============================================================== rows in
    {
#pragma unroll
        for (int p = 0; p < 32 * V / 256; ++p) {
            const int f = tid + 256 * p, r = f / V, c4 = f % V;
            const int64_t row = min(m0 + r, a.M - 1);                    // (rows beyond M: a clamped row, never stored)
            *reinterpret_cast<f32x4*>(Xs + r * XS + 4 * c4) = *reinterpret_cast<const f32x4*>(a.X + row * a.ldx + 4 * c4);
            if constexpr (PRE) *reinterpret_cast<f32x4*>(Ys + r * XS + 4 * c4) = *reinterpret_cast<const f32x4*>(a.R + row * a.ldr + 4 * c4);
        }
    }
    if constexpr (!PRE) {                                               // statistics partials written by the rows' producer: Chan merge
        const float* sp = a.ln_stats + min(m0 + li, a.M - 1) * a.ln_parts * 2;
        const float ni = (float)(D / a.ln_parts), inv_ni = 1.0f / ni;
        float cnt = 0.f, mu = 0.f, m2 = 0.f;
        for (int p = 0; p < a.ln_parts; ++p) {
            const float2 sq = *reinterpret_cast<const float2*>(sp + 2 * p);
            const float dl = sq.x * inv_ni - mu, tot = cnt + ni;
            mu += dl * (ni / tot);
            m2 += sq.y + dl * dl * (cnt * ni / tot);
            cnt = tot;
        }
        mean = mu;
        rstd = 1.0f / sqrtf(m2 * (1.0f / (float)D) + a.ln_eps);
    }
    __syncthreads();

    // =============================================================== PRE: rows <- X.Wpre^T + bpre + R   (wave w: columns QW w ..)
    if constexpr (PRE) {
        gemm_rows(reinterpret_cast<const f32x4*>(a.Wpre) + (int64_t)wave * NL * TPW * 64 + lane, std::integral_constant<int, TPW>{});
        __syncthreads();                                               // every wave has read the A rows: they are replaced now
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = QW * wave + 32 * t + 8 * q + 4 * hf;
                const f32x4 bb = *reinterpret_cast<const f32x4*>(a.bpre + col);
                const f32x4 rr = *reinterpret_cast<const f32x4*>(Ys + li * XS + col);
                *reinterpret_cast<f32x4*>(Xs + li * XS + col) =
                    f32x4{y[t][4 * q], y[t][4 * q + 1], y[t][4 * q + 2], y[t][4 * q + 3]} + bb + rr;
            }
        __syncthreads();
        row_stats(a.ln_eps, a.Y1, a.ldy1);
    }

    // =============================================================== CORE: the feed-forward sub-layer (ffn_fused_f32.hip)
    if constexpr (CORE) {
        const int NS = a.hidden >> 7;                                   // slices per wave
        const f32x4* wp = reinterpret_cast<const f32x4*>(a.Wp) + lane;
        const int64_t TILE = a.tile_stride;
        const int rot = a.rotate ? (int)((blockIdx.x >> 3) % (unsigned)NS) : 0;
        f32x4 ring[RING];
        {
            const f32x4* sb = wp + (4 * rot + wave) * TILE;
#pragma unroll
            for (int i = 0; i < RING; ++i) ring[i] = sb[i * 64];
        }
#pragma unroll
        for (int t = 0; t < ND; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) y[t][r] = 0.f;
        for (int s = 0; s < NS; ++s) {
            const int sr = s + rot < NS ? s + rot : s + rot - NS;
            const int g = 4 * sr + wave;
            const f32x4* sb = wp + g * TILE;
            const f32x4* nb = wp + (4 * (sr + 1 < NS ? sr + 1 : 0) + wave) * TILE;   // (after the last slice: loads that are never used)
            f32x4 csv[4], b1v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                csv[q] = *reinterpret_cast<const f32x4*>(a.cs1 + g * 32 + 8 * q + 4 * hf);
                b1v[q] = *reinterpret_cast<const f32x4*>(a.b1f + g * 32 + 8 * q + 4 * hf);
            }
            f32x16 ha;
#pragma unroll
            for (int r = 0; r < 16; ++r) ha[r] = 0.f;
            f32x4 xv = *reinterpret_cast<const f32x4*>(xrow);
#pragma unroll
            for (int c = 0; c < NL; ++c) {
                const f32x4 xn = *reinterpret_cast<const f32x4*>(xrow + 8 * (c + 1));
                const f32x4 wv = ring[c % RING];
                ring[c % RING] = sb[(c + RING) * 64];
                ha = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[0], xv[0], ha, 0, 0, 0);
                ha = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[1], xv[1], ha, 0, 0, 0);
                ha = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[2], xv[2], ha, 0, 0, 0);
                ha = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[3], xv[3], ha, 0, 0, 0);
                xv = xn;
                __builtin_amdgcn_sched_barrier(0);
            }
            float sw[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = rstd * (ha[r] - mean * csv[r >> 2][r & 3]) + b1v[r >> 2][r & 3];
                sw[r] = swishf_acc(v);
            }
#pragma unroll
            for (int j = 0; j < NL; j += 2) {
                const int q = j / ND, t = j % ND;
                const f32x4 w0 = ring[j % RING], w1 = ring[(j + 1) % RING];
                ring[j % RING] = j + RING < NL ? sb[(NL + j + RING) * 64] : nb[(j + RING - NL) * 64];
                ring[(j + 1) % RING] = j + 1 + RING < NL ? sb[(NL + j + 1 + RING) * 64] : nb[(j + 1 + RING - NL) * 64];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    y[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0[e], sw[4 * q + e], y[t], 0, 0, 0);
                    y[t + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[e], sw[4 * q + e], y[t + 1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        // ---- sum the four waves' partial tiles and finish the rows, one QUARTER of the columns at a time (see ffn_fused_f32.hip)
        float* Ex = Ys;                                                  // [4][32][EW]
        const bool cok = 4 * li < QW;
        const int cl = cok ? 4 * li : 0;
        f32x4 v[4][4];                                                   // [quarter][it]: row 8 w + 2 it + hf, columns QW c + 4 li ..
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int k = 0; k < TPW; ++k)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<f32x4*>(Ex + wave * ERS + li * EW + 32 * k + 8 * q + 4 * hf) =
                        f32x4{y[c * TPW + k][4 * q], y[c * TPW + k][4 * q + 1], y[c * TPW + k][4 * q + 2], y[c * TPW + k][4 * q + 3]};
            const f32x4 b2v = *reinterpret_cast<const f32x4*>(a.b2 + QW * c + cl);
            __syncthreads();
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int r = 8 * wave + 2 * it + hf;
                const float* e = Ex + r * EW + cl;
                f32x4 acc = *reinterpret_cast<const f32x4*>(e);
                acc += *reinterpret_cast<const f32x4*>(e + ERS);
                acc += *reinterpret_cast<const f32x4*>(e + 2 * ERS);
                acc += *reinterpret_cast<const f32x4*>(e + 3 * ERS);
                const f32x4 xr = *reinterpret_cast<const f32x4*>(Xs + r * XS + QW * c + cl);
                v[c][it] = cok ? a.alpha * (acc + b2v) + xr : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            __syncthreads();
        }
        auto half_sum = [](float (&t)[4]) {                              // four rows at once: sums over the 32 lanes of a half
#pragma unroll
            for (int o = 16; o > 0; o >>= 1)
#pragma unroll
                for (int it = 0; it < 4; ++it) t[it] += __shfl_xor(t[it], o, 64);
        };
        if constexpr (POST) {
            // the finished rows become the resident rows of the POST product; their LayerNorm statistics go to St.  (Every wave
            // has passed the barrier behind the last quarter: nobody reads the old rows any more.)
            float s[4], q[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                s[it] = 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    s[it] += (v[c][it][0] + v[c][it][1]) + (v[c][it][2] + v[c][it][3]);
                    if (cok) *reinterpret_cast<f32x4*>(Xs + (8 * wave + 2 * it + hf) * XS + QW * c + cl) = v[c][it];
                }
            }
            half_sum(s);
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const float mu = s[it] * (1.0f / (float)D);
                q[it] = 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (cok) {
                        const f32x4 dv = v[c][it] - mu;
                        q[it] += (dv[0] * dv[0] + dv[1] * dv[1]) + (dv[2] * dv[2] + dv[3] * dv[3]);
                    }
                }
                s[it] = mu;
            }
            half_sum(q);
#pragma unroll
            for (int it = 0; it < 4; ++it)
                if (li == 0) {
                    St[2 * (8 * wave + 2 * it + hf)] = s[it];
                    St[2 * (8 * wave + 2 * it + hf) + 1] = 1.0f / sqrtf(q[it] * (1.0f / (float)D) + a.post_eps);
                }
        }
        if constexpr (MODE == CH_LN) {
            f32x4 gam[4], bet[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                gam[c] = *reinterpret_cast<const f32x4*>(a.gamma2 + QW * c + cl);
                bet[c] = *reinterpret_cast<const f32x4*>(a.beta2 + QW * c + cl);
            }
            float s[4], q[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                s[it] = 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) s[it] += (v[c][it][0] + v[c][it][1]) + (v[c][it][2] + v[c][it][3]);
            }
            half_sum(s);
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const float mu = s[it] * (1.0f / (float)D);
                q[it] = 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (cok) {
                        v[c][it] = v[c][it] - mu;
                        q[it] += (v[c][it][0] * v[c][it][0] + v[c][it][1] * v[c][it][1]) + (v[c][it][2] * v[c][it][2] + v[c][it][3] * v[c][it][3]);
                    }
                }
            }
            half_sum(q);
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const float rs = 1.0f / sqrtf(q[it] * (1.0f / (float)D) + a.eps2);
                s[it] = 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (cok) {
                        v[c][it] = v[c][it] * rs * gam[c] + bet[c];
                        s[it] += (v[c][it][0] + v[c][it][1]) + (v[c][it][2] + v[c][it][3]);
                    }
                }
            }
            if (a.stats_out) {                                           // (kernel-uniform) ONE partial per output row
                half_sum(s);
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const float mo = s[it] * (1.0f / (float)D);
                    q[it] = 0.f;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        if (cok) {
                            const f32x4 tt = v[c][it] - mo;
                            q[it] += (tt[0] * tt[0] + tt[1] * tt[1]) + (tt[2] * tt[2] + tt[3] * tt[3]);
                        }
                    }
                }
                half_sum(q);
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int64_t row = m0 + 8 * wave + 2 * it + hf;
                    if (li == 0 && row < a.M) *reinterpret_cast<float2*>(a.stats_out + 2 * row) = float2{s[it], q[it]};
                }
            }
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int64_t row = m0 + 8 * wave + 2 * it + hf;
            const bool row_ok = row < a.M;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (row_ok && cok) *reinterpret_cast<f32x4*>(a.Y + row * a.ldy + QW * c + cl) = v[c][it];
                if constexpr (MODE == CH_STATS) {                        // (sum, M2 about their own mean) of every 32 stored columns
                    float s = (v[c][it][0] + v[c][it][1]) + (v[c][it][2] + v[c][it][3]);
                    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
                    const float m = s * (1.0f / 32.0f);
                    const f32x4 dv = v[c][it] - m;
                    float q = (dv[0] * dv[0] + dv[1] * dv[1]) + (dv[2] * dv[2] + dv[3] * dv[3]);
                    q += __shfl_xor(q, 1, 64); q += __shfl_xor(q, 2, 64); q += __shfl_xor(q, 4, 64);
                    if (row_ok && cok && (li & 7) == 0)
                        *reinterpret_cast<float2*>(a.stats_out + (row * (int64_t)ND + ((QW * c + cl) >> 5)) * 2) = float2{s, q};
                }
            }
        }
        if constexpr (POST) {
            __syncthreads();                                           // rows + St written by all waves
            mean = St[2 * li]; rstd = St[2 * li + 1];
        }
    }

    // =============================================================== POST: Z = epi(LN(rows).Wpost^T + bpost)
    if constexpr (POST != POST_NONE) {
        constexpr int NT = POST == POST_BIAS3 ? 3 * TPW : 2 * TPW;       // tiles per wave: columns [32 NT w, 32 NT (w + 1)) of q|k|v; GLU: TPW value + TPW gate tiles
        constexpr int NOUT = POST == POST_BIAS3 ? 3 * D : D;             // output columns
        gemm_rows(reinterpret_cast<const f32x4*>(a.Wpost) + (int64_t)wave * NL * NT * 64 + lane, std::integral_constant<int, NT>{});
        float* stg = Ys + wave * ERS;                                    // wave-private staging tile [32][EW]
        const bool cok = 4 * li < QW;
        const int cl = cok ? 4 * li : 0;
        constexpr int NP = POST == POST_BIAS3 ? 3 : 1;                   // passes of TPW output tiles
#pragma unroll
        for (int p = 0; p < NP; ++p) {
#pragma unroll
            for (int k = 0; k < TPW; ++k)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 o;
                    if constexpr (POST == POST_GLU) {
                        // value tile k and gate tile TPW + k of this wave: output columns QW w + 32 k ..; weight rows n (value), D + n (gate)
                        const int n = QW * wave + 32 * k + 8 * q + 4 * hf;
                        const f32x4 cv = *reinterpret_cast<const f32x4*>(a.cspost + n), cg = *reinterpret_cast<const f32x4*>(a.cspost + D + n);
                        const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bpost + n), bg = *reinterpret_cast<const f32x4*>(a.bpost + D + n);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float val = rstd * (y[k][4 * q + e] - mean * cv[e]) + bv[e];
                            const float gat = rstd * (y[TPW + k][4 * q + e] - mean * cg[e]) + bg[e];
                            o[e] = val * sigmoidf_acc(gat);
                        }
                    } else {
                        const int n = 32 * NT * wave + 32 * (TPW * p + k) + 8 * q + 4 * hf;
                        const f32x4 cv = *reinterpret_cast<const f32x4*>(a.cspost + n);
                        const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bpost + n);
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = rstd * (y[TPW * p + k][4 * q + e] - mean * cv[e]) + bv[e];
                    }
                    *reinterpret_cast<f32x4*>(stg + li * EW + 32 * k + 8 * q + 4 * hf) = o;
                }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int colbase = POST == POST_GLU ? QW * wave : 32 * NT * wave + QW * p;
#pragma unroll
            for (int it = 0; it < 16; ++it) {                            // rows 2 it + hf of the tile, 512-byte row segments per half wave
                const int r = 2 * it + hf;
                const f32x4 o = *reinterpret_cast<const f32x4*>(stg + r * EW + cl);
                if (cok && m0 + r < a.M) *reinterpret_cast<f32x4*>(a.Z + (m0 + r) * a.ldz + colbase + cl) = o;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // the reads have landed before the next pass rewrites the tile
            __builtin_amdgcn_wave_barrier();
        }
        (void)NOUT;
    }
}

// Weight (N, K = d) of a d-input product in MFMA A-fragment order for rowchain's gemm_rows: wave w, chunk c, tile t of the wave:
//   Wp[((w NL + c) NT + t) 64 + lane][e] = W[row(w, t) + li][8 c + 4 hf + e],  NL = d / 8, NT = tiles per wave
//   glu == 0: N = 128 NT, row(w, t) = 32 (NT w + t);  glu == 1: N = 2 d, NT = 2 TPW: t < TPW value rows 32 (TPW w + t), else gate rows
//   d + 32 (TPW w + t - TPW)
__global__ __launch_bounds__(256) void rowgemm_pack_kernel(const float* __restrict__ W, float* __restrict__ Wp, int N, int d, int glu) {
    const int nl = d >> 3, nt = N >> 7, tpw = d >> 7;
    const int64_t total = (int64_t)4 * nl * nt * 64;
    for (int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x; f < total; f += (int64_t)gridDim.x * 256) {
        const int lane = (int)(f & 63), li = lane & 31, hf = lane >> 5;
        int64_t r = f >> 6;
        const int t = (int)(r % nt); r /= nt;
        const int c = (int)(r % nl), w = (int)(r / nl);
        const int row = glu ? (t < tpw ? 32 * (tpw * w + t) : d + 32 * (tpw * w + t - tpw)) : 32 * (nt * w + t);
        reinterpret_cast<f32x4*>(Wp)[f] = *reinterpret_cast<const f32x4*>(W + (int64_t)(row + li) * d + 8 * c + 4 * hf);
    }
}

}  // namespace

extern "C" int cfm_rowgemm_pack_f32(const float* W, float* Wp, int N, int d, int glu, cfm_stream_t stream) {
    CFM_REQUIRE(W && Wp, CFM_ERR_NULL);
    CFM_REQUIRE((d == 128 || d == 256 || d == 512) && N > 0 && N % 128 == 0 && (!glu || N == 2 * d), CFM_ERR_UNSUPPORTED);
    CFM_REQUIRE(CFM_ALIGNED16(W) && CFM_ALIGNED16(Wp), CFM_ERR_ALIGN);
    const int64_t total = (int64_t)N * d / 4;
    const unsigned blocks = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(rowgemm_pack_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), W, Wp, N, d, glu);
    return cfm_launch_status();
}

extern "C" int64_t cfm_ffn_tile_stride_f4(int d);   // ffn_fused_f32.hip: f32x4 between packed feed-forward tiles
extern "C" int cfm_ffn_rotate(void);

// pre: 0 | 1; core: 0 | 1; post: 0 | 1 (q|k|v: Wpost (3 d, d)) | 2 (GLU: Wpost (2 d, d)); mode: finish of the CORE rows (0 plain | 1 +
// statistics partials | 2 closing LayerNorm); see include/conformer_hip.h for the argument rules
extern "C" int cfm_rowchain_f32(int pre, int core, int post, int mode, const float* X, int64_t ldx, const float* Wpre_packed,
                                const float* bpre, const float* R, int64_t ldr, float* Y1, int64_t ldy1, const float* ln_stats,
                                int ln_parts, float ln_eps, const float* Wffn_packed, const float* b1f, const float* colsum1,
                                const float* b2, float alpha, int hidden, float* Y, int64_t ldy, float* stats_out,
                                const float* gamma2, const float* beta2, float eps2, const float* Wpost_packed, const float* bpost,
                                const float* cspost, float post_eps, float* Z, int64_t ldz, int64_t M, int d, cfm_stream_t stream) {
    CFM_REQUIRE(X && M >= 0 && ldx >= d && ldx % 4 == 0 && CFM_ALIGNED16(X), CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(d == 128 || d == 256 || d == 512, CFM_ERR_UNSUPPORTED);
    CFM_REQUIRE((pre == 0 || pre == 1) && (core == 0 || core == 1) && post >= 0 && post <= 2 && mode >= 0 && mode <= 2, CFM_ERR_UNSUPPORTED);
    CFM_REQUIRE(core || (pre && post), CFM_ERR_UNSUPPORTED);
    if (pre) {
        CFM_REQUIRE(Wpre_packed && bpre && R, CFM_ERR_NULL);
        CFM_REQUIRE(ldr >= d && ldr % 4 == 0 && CFM_ALIGNED16(Wpre_packed) && CFM_ALIGNED16(bpre) && CFM_ALIGNED16(R), CFM_ERR_ALIGN);
        CFM_REQUIRE(!Y1 || (ldy1 >= d && ldy1 % 4 == 0 && CFM_ALIGNED16(Y1)), CFM_ERR_ALIGN);
    } else {
        CFM_REQUIRE(ln_stats, CFM_ERR_NULL);
        CFM_REQUIRE(ln_parts >= 1 && ln_parts <= 16 && (ln_parts & (ln_parts - 1)) == 0 && (reinterpret_cast<uintptr_t>(ln_stats) & 7u) == 0,
                    CFM_ERR_UNSUPPORTED);
    }
    if (core) {
        CFM_REQUIRE(Wffn_packed && b1f && colsum1 && b2 && Y, CFM_ERR_NULL);
        CFM_REQUIRE(hidden > 0 && hidden % 128 == 0, CFM_ERR_UNSUPPORTED);
        CFM_REQUIRE(ldy >= d && ldy % 4 == 0 && CFM_ALIGNED16(Wffn_packed) && CFM_ALIGNED16(b1f) && CFM_ALIGNED16(colsum1) && CFM_ALIGNED16(b2) &&
                    CFM_ALIGNED16(Y) && (reinterpret_cast<uintptr_t>(stats_out) & 7u) == 0, CFM_ERR_ALIGN);
        CFM_REQUIRE(mode != CH_STATS || stats_out, CFM_ERR_NULL);
        CFM_REQUIRE(mode != CH_LN || (gamma2 && beta2 && CFM_ALIGNED16(gamma2) && CFM_ALIGNED16(beta2)), CFM_ERR_NULL);
    }
    if (post) {
        const int nout = post == POST_BIAS3 ? 3 * d : d;
        CFM_REQUIRE(Wpost_packed && bpost && cspost && Z, CFM_ERR_NULL);
        CFM_REQUIRE(ldz >= nout && ldz % 4 == 0 && CFM_ALIGNED16(Wpost_packed) && CFM_ALIGNED16(bpost) && CFM_ALIGNED16(cspost) && CFM_ALIGNED16(Z),
                    CFM_ERR_ALIGN);
    }
    if (M == 0) return CFM_OK;
    ChainArgs a{X, ldx, Wpre_packed, bpre, R, ldr, Y1, ldy1, ln_stats, ln_parts, ln_eps, Wffn_packed, b1f, colsum1, b2, alpha, hidden,
                cfm_ffn_tile_stride_f4(d), cfm_ffn_rotate(), Y, ldy, stats_out, gamma2, beta2, eps2, Wpost_packed, bpost, cspost,
                post_eps, Z, ldz, M};
    const dim3 grid((unsigned)((M + 31) / 32)), block(256);
    hipStream_t s = static_cast<hipStream_t>(stream);
    // the chains of the block: K1 = (0,1,1,plain), K2 = (1,0,2,-), K3 = (1,1,0,LN)
#define CH_LAUNCH(ND, PRE, CORE, POST, MODE) hipLaunchKernelGGL((rowchain_f32_kernel<ND, PRE, CORE, POST, MODE>), grid, block, 0, s, a)
#define CH_BY_D(PRE, CORE, POST, MODE) do { if (d == 512) CH_LAUNCH(16, PRE, CORE, POST, MODE); else if (d == 256) CH_LAUNCH(8, PRE, CORE, POST, MODE); \
                                            else CH_LAUNCH(4, PRE, CORE, POST, MODE); } while (0)
    if (!pre && core && post == POST_BIAS3 && mode == CH_PLAIN) CH_BY_D(0, 1, POST_BIAS3, CH_PLAIN);
    else if (pre && !core && post == POST_GLU) CH_BY_D(1, 0, POST_GLU, CH_PLAIN);
    else if (pre && core && post == POST_NONE && mode == CH_LN) CH_BY_D(1, 1, POST_NONE, CH_LN);
    else return CFM_ERR_UNSUPPORTED;
#undef CH_BY_D
#undef CH_LAUNCH
    return cfm_launch_status();
}
