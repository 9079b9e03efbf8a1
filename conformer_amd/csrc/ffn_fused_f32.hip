// Fused Macaron feed-forward sub-layer, fp32 inference (ffn.py:15-23 + the residual of block.py:19,25, optionally + block.py:27):
//     Y = alpha * (swish(LN(X) W1^T + b1) W2^T + b2) + X
// in ONE kernel per 32 rows: the (rows, 4d) hidden activation never leaves the CU.  Why (rounds 1-3): the
// two FFN GEMMs are 43 % of the cfg-2 forward and sit at 0.73 / 0.76 of the fp32 matrix pipe where the long-K stem GEMM reaches
// 0.82 -- every 128x64 / 64x128 tile pays a prologue and an HBM-bound epilogue burst (65 MB of hidden activations written and
// read back per FFN), and the K-loop pays an L2 -> registers -> LDS refill with a barrier per K-tile.
//
// Structure (4 waves = one per SIMD, each with the whole 512-register budget; no barrier in the main loop):
//   * the workgroup's 32 un-normalised rows of X sit in LDS for the whole kernel (MFMA B operand of stage 1; residual at the end);
//   * the hidden dimension is cut into 32-unit SLICES; wave w owns slices w, w+4, ... (split-K of the second product over waves);
//   * stage 1 of a slice:  H^T (32 hidden x 32 rows) = W1f[slice] . X^T     d/2 MFMAs (v_mfma_f32_32x32x2_f32), ONE dependent chain
//     (two chains -- 32 accumulator registers beside the 256 of stage 2 -- made the compiler park two output tiles in VGPRs around
//     every slice: 268 us against 264)
//     epilogue (registers): LN fold  rstd * (acc - mean * colsum) + b1f,  Swish
//   * stage 2 of a slice:  Y^T (d x 32 rows) += W2[:, slice] . swish(H)^T -- the stage-1 accumulator layout (lane = row, registers
//     = 16 hidden units) IS the B-operand layout of a 32x32x2 MFMA whose contraction runs over those hidden units: no LDS round
//     trip, no barrier.  d/32 accumulator tiles per wave (256 registers at d = 512: the AGPR half of the file);
//   * both weights are PRE-PACKED in MFMA A-fragment order, slice by slice (cfm_ffn_pack_f32): a wave's whole weight stream is
//     one contiguous run of 1 KB wave loads (dwordx4 per lane = the operands of 4 MFMAs), L2 -> registers through a ring of 16
//     loads in flight; nothing but X passes through LDS;
//   * after the last slice the four waves' partial Y tiles are summed through LDS in a FIXED order (bit-reproducible), and a
//     row-major epilogue applies b2, alpha, the residual and one of: nothing | the LayerNorm-statistics partials of the stored rows
//     (consumed by the next sub-layer's folded LayerNorm) | the block-closing LayerNorm (block.py:27) + the statistics of ITS output.
// 249 workgroups at cfg-2 (7968 rows): one per CU, one round, every workgroup does identical work.
#include "cfm_common.h"

namespace {

struct FfnArgs {
    const float* X; int64_t ldx;                         // (M, d) un-normalised rows = the residual
    const float* ln_stats; int ln_parts; float ln_eps;   // [M][ln_parts][2] = (sum, M2 about their own mean) of equal column groups
    const float* Wp;                                     // packed weights (cfm_ffn_pack_f32)
    const float* b1f; const float* cs1; const float* b2; // folded hidden bias, column sums of the folded W1, output bias
    float* Y; int64_t ldy;
    float* stats_out;                                    // mode 1: [M][d/32][2]; mode 2: [M][1][2] or null
    const float* gamma2; const float* beta2; float eps2; // mode 2: the closing LayerNorm
    int64_t M; int hidden; float alpha;
    int64_t tile_stride;                                 // f32x4 between consecutive packed tiles (2 NL 64 + pad)
    int rotate;                                          // workgroups start at different slices (see the main loop)
    int trace_detail;                                    // diagnostics: per-slice stamps too (they perturb the loop)
    unsigned long long* trace;                           // diagnostics (cfm_debug_ffn_trace): s_memrealtime stamps of wave 0 of blocks 0 / 128
};

enum { FFN_PLAIN = 0, FFN_STATS = 1, FFN_LN = 2 };

template <int ND, int MODE, int DBG = 0, int RING = 16>    // DBG (diagnostics): 1 = no weight loads in the main loop (pure MFMA rate)
__global__ __launch_bounds__(256, 1) void ffn_fused_f32_kernel(const FfnArgs a) {
    constexpr int D = 32 * ND, XS = D + 4, NL = 4 * ND;     // NL = wave loads per stage and slice
    static_assert(NL % RING == 0 && ND % 2 == 0, "d = 128, 256 or 512");
    __shared__ __attribute__((aligned(16))) float smem[32 * XS + 32 * (D + 16)];
    float* Xs = smem;                      // [32][XS]  rows of X (516-float rows: conflict-free ds_read_b128 fragments)
    float* Ys = smem + 32 * XS;            // [4][32][D / 4 + 4]  exchange regions of the tile sum, then the summed output tile

    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, hf = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t m0 = (int64_t)blockIdx.x * 32;
    const int NS = a.hidden >> 7;                                       // slices per wave
    unsigned long long* tr = nullptr;                                   // [64] per traced block: start, prologue, (stage 1, stage 2) x NS, sum, end
    if (a.trace && tid == 0 && (blockIdx.x == 0 || blockIdx.x == 128)) tr = a.trace + (blockIdx.x ? 64 : 0);
#define FFN_STAMP(i) do { if (tr) tr[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
    FFN_STAMP(0);

    // ---- the weight stream of this wave: slice s = packed tile 4 s + wave = 2 NL consecutive 1 KB wave loads
    const f32x4* wp = reinterpret_cast<const f32x4*>(a.Wp) + lane;
    const int64_t TILE = a.tile_stride;                                 // f32x4 from one packed tile to the next
    // Every workgroup streams the SAME 2 x 4 MB of weights, in lockstep.  Two ways of pulling the workgroups apart were measured
    // and change NOTHING in time (seven layouts: 275-278 us): starting each workgroup at a different slice (rotation by its index
    // within its XCD), and padding the packed tiles apart (against channel-interleave periods).  What the rotation does change is
    // the traffic behind the L2: with all 31 workgroups of an XCD on the same slice the 4 MB L2 serves 30 of 31 reads (85 MB
    // fetched per launch = 8 XCDs x the weights + the rows), rotated they thrash it (1.1 GB per launch, PMC FETCH_SIZE).  Off.
    const int rot = a.rotate ? (int)((blockIdx.x >> 3) % (unsigned)NS) : 0;
    f32x4 ring[RING];
    {
        const f32x4* sb = wp + (4 * rot + wave) * TILE;
#pragma unroll
        for (int i = 0; i < RING; ++i) ring[i] = sb[i * 64];
    }

    // ---- X tile -> LDS (rows beyond M: a clamped row, never stored)
    {
        constexpr int V = D / 4;                                        // float4 per row
#pragma unroll
        for (int p = 0; p < 32 * V / 256; ++p) {
            const int f = tid + 256 * p, r = f / V, c4 = f % V;
            const f32x4 v = *reinterpret_cast<const f32x4*>(a.X + min(m0 + r, a.M - 1) * a.ldx + 4 * c4);
            *reinterpret_cast<f32x4*>(Xs + r * XS + 4 * c4) = v;
        }
    }
    // ---- LayerNorm statistics of this lane's row: Chan merge of the partials in index order
    float mean, rstd;
    {
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
    FFN_STAMP(1);
    if (tr) tr[60] = __builtin_amdgcn_s_memtime();                      // shader-clock counter: the clock the main loop really ran at

    f32x16 y[ND];
#pragma unroll
    for (int t = 0; t < ND; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) y[t][r] = 0.f;

    const float* xrow = Xs + li * XS + 4 * hf;
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
        // ---- stage 1: H^T = W1f[slice] . X^T, contraction over the d input dims (8 per load)
        f32x16 ha;
#pragma unroll
        for (int r = 0; r < 16; ++r) ha[r] = 0.f;
        f32x4 xv = *reinterpret_cast<const f32x4*>(xrow);
#pragma unroll
        for (int c = 0; c < NL; ++c) {
            // the X fragment of the NEXT step is requested ahead of this step's MFMAs (the last one of a slice reads 16 bytes past
            // the row: pad / the next row, unused); each step is pinned as one scheduling group -- left alone, the scheduler sinks
            // the weight loads of a whole ring round to its end and the ring is in effect zero deep at the start of every round
            const f32x4 xn = *reinterpret_cast<const f32x4*>(xrow + 8 * (c + 1));
            const f32x4 wv = ring[c % RING];
            if (DBG != 1) ring[c % RING] = sb[(c + RING) * 64];
            ha = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[0], xv[0], ha, 0, 0, 0);
            ha = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[1], xv[1], ha, 0, 0, 0);
            ha = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[2], xv[2], ha, 0, 0, 0);
            ha = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[3], xv[3], ha, 0, 0, 0);
            xv = xn;
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- folded LayerNorm + bias + Swish: lane = row, register r = hidden unit 8 (r >> 2) + 4 hf + (r & 3) of the slice
        float sw[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = rstd * (ha[r] - mean * csv[r >> 2][r & 3]) + b1v[r >> 2][r & 3];
            sw[r] = swishf_acc(v);
        }
        if (a.trace_detail) FFN_STAMP(2 + 2 * s);
        // ---- stage 2: Y^T += W2[:, slice] . swish(H)^T; loads in (q, t) order, two output tiles interleaved per step
#pragma unroll
        for (int j = 0; j < NL; j += 2) {
            const int q = j / ND, t = j % ND;
            const f32x4 w0 = ring[j % RING], w1 = ring[(j + 1) % RING];
            if (DBG != 1) {
                ring[j % RING] = j + RING < NL ? sb[(NL + j + RING) * 64] : nb[(j + RING - NL) * 64];
                ring[(j + 1) % RING] = j + 1 + RING < NL ? sb[(NL + j + 1 + RING) * 64] : nb[(j + 1 + RING - NL) * 64];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                y[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0[e], sw[4 * q + e], y[t], 0, 0, 0);
                y[t + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[e], sw[4 * q + e], y[t + 1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);                           // (pinned like the steps of stage 1)
        }
        if (a.trace_detail) FFN_STAMP(3 + 2 * s);
    }

    if (tr) { tr[61] = __builtin_amdgcn_s_memtime(); tr[62] = __builtin_amdgcn_s_memrealtime(); }
    // ---- sum the four waves' partial tiles and finish the rows, one QUARTER of the columns at a time (static register indices,
    // identical code in all waves): every wave leaves its partial of output tiles c TPW .. c TPW + TPW - 1 in its LDS region, then
    // wave w collects rows 8 w .. 8 w + 7 of that quarter: lane (li, hf) of step `it` owns row 8 w + 2 it + hf, columns QW c + 4 li ..
    // + 3, and adds the four partials in wave order (bit-reproducible).  (First version: the waves took turns adding their whole
    // 32 x d partial into one LDS tile, then a second pass read it back row by row: 8.4 + 4.7 us of serial passes.)
    constexpr int TPW = ND / 4, QW = 32 * TPW, EW = QW + 4, ERS = 32 * EW;   // tiles per quarter, its columns, region row / size (floats)
    float* Ex = Ys;                                                          // [4][32][EW]
    const bool cok = 4 * li < QW;                                            // (d < 512: a quarter is narrower than 32 float4)
    const int cl = cok ? 4 * li : 0;
    FFN_STAMP(2 + 2 * NS);
    f32x4 v[4][4];                                                           // [quarter][it]
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
        if (c < 3) __syncthreads();
    }
    auto half_sum = [](float (&t)[4]) {                                      // four rows at once: sums over the 32 lanes of a half
#pragma unroll
        for (int o = 16; o > 0; o >>= 1)
#pragma unroll
            for (int it = 0; it < 4; ++it) t[it] += __shfl_xor(t[it], o, 64);
    };
    if constexpr (MODE == FFN_LN) {
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
        if (a.stats_out) {                                                   // (kernel-uniform) ONE partial per output row
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
            if constexpr (MODE == FFN_STATS) {                               // (sum, M2 about their own mean) of every 32 stored columns
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
    FFN_STAMP(3 + 2 * NS);
#undef FFN_STAMP
}

// Wp tile g (32 hidden units), 2 NL wave loads of 1 KB:
//   loads 0 .. NL-1     (stage 1, c = input-dim chunk):        lane (li, hf), element e = W1f[32 g + li][8 c + 4 hf + e]
//   loads NL .. 2 NL-1  (stage 2, j = q ND + t):                lane (li, hf), element e = W2[32 t + li][32 g + 8 q + 4 hf + e]
__global__ __launch_bounds__(256) void ffn_pack_kernel(const float* __restrict__ W1f, const float* __restrict__ W2,
                                                       float* __restrict__ Wp, int d, int hidden, int64_t tile_stride) {
    const int nd = d >> 5, nl = 4 * nd;
    const int64_t total = (int64_t)(hidden >> 5) * 2 * nl * 64;        // float4 elements (pads are left as allocated)
    for (int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x; f < total; f += (int64_t)gridDim.x * 256) {
        const int lane = (int)(f & 63), li = lane & 31, hf = lane >> 5;
        const int64_t ld = f >> 6;
        const int i = (int)(ld % (2 * nl)), g = (int)(ld / (2 * nl));
        const float* src;
        if (i < nl) src = W1f + (int64_t)(32 * g + li) * d + 8 * i + 4 * hf;
        else {
            const int j = i - nl, q = j / nd, t = j % nd;
            src = W2 + (int64_t)(32 * t + li) * hidden + 32 * g + 8 * q + 4 * hf;
        }
        reinterpret_cast<f32x4*>(Wp)[(int64_t)g * tile_stride + (int64_t)i * 64 + lane] = *reinterpret_cast<const f32x4*>(src);
    }
}

}  // namespace

// diagnostics: pad between packed tiles (f32x4) and per-workgroup slice rotation -- both measured neutral in time; the rotation
// multiplies the fetch traffic behind the L2 by 13 (see the kernel): defaults 0 / off
static int g_ffn_pad = 0;
static int g_ffn_rotate = 0;
extern "C" int cfm_debug_ffn_layout(int pad_f4, int rotate) {           // diagnostics: affects cfm_ffn_pack_* AND cfm_ffn_fused_f32 -- re-pack after it
    if (pad_f4 >= 0) g_ffn_pad = pad_f4;
    g_ffn_rotate = rotate != 0;
    return CFM_OK;
}
static int64_t ffn_tile_stride(int d) { return (int64_t)2 * (d / 8) * 64 + g_ffn_pad; }
extern "C" int64_t cfm_ffn_tile_stride_f4(int d) { return ffn_tile_stride(d); }   // (rowchain_f32.hip runs the same packed stream)
extern "C" int cfm_ffn_rotate(void) { return g_ffn_rotate; }
extern "C" int64_t cfm_ffn_pack_elems(int d, int hidden) { return ffn_tile_stride(d) * (hidden / 32) * 4; }

extern "C" int cfm_ffn_pack_f32(const float* W1f, const float* W2, float* Wp, int d, int hidden, cfm_stream_t stream) {
    CFM_REQUIRE(W1f && W2 && Wp, CFM_ERR_NULL);
    CFM_REQUIRE((d == 128 || d == 256 || d == 512) && hidden > 0 && hidden % 128 == 0, CFM_ERR_UNSUPPORTED);
    CFM_REQUIRE(CFM_ALIGNED16(W1f) && CFM_ALIGNED16(W2) && CFM_ALIGNED16(Wp), CFM_ERR_ALIGN);
    const int64_t total = (int64_t)2 * d * hidden / 4;
    const unsigned blocks = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(ffn_pack_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), W1f, W2, Wp, d, hidden,
                       ffn_tile_stride(d));
    return cfm_launch_status();
}

static unsigned long long* g_ffn_trace = nullptr;
static int g_ffn_dbg = 0;
extern "C" int cfm_debug_ffn_variant(int v) { const int p = g_ffn_dbg; g_ffn_dbg = v; return p; }
// diagnostics (tools/ffn_fused_ab.py trace): device buffer of 128 x uint64 filled by the next launches, or NULL to stop
static int g_ffn_trace_detail = 0;
extern "C" int cfm_debug_ffn_trace(void* trace_or_null, int per_slice) {
    g_ffn_trace = static_cast<unsigned long long*>(trace_or_null);
    g_ffn_trace_detail = per_slice;
    return CFM_OK;
}

extern "C" int cfm_ffn_fused_f32(const float* X, int64_t ldx, const float* ln_stats, int ln_parts, float ln_eps, const float* Wp,
                                 const float* b1f, const float* colsum1, const float* b2, float alpha, float* Y, int64_t ldy,
                                 int mode, float* stats_out, const float* gamma2, const float* beta2, float eps2, int64_t M,
                                 int d, int hidden, cfm_stream_t stream) {
    CFM_REQUIRE(X && ln_stats && Wp && b1f && colsum1 && b2 && Y, CFM_ERR_NULL);
    CFM_REQUIRE(M >= 0 && ldx >= d && ldy >= d && ln_eps >= 0.f, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE((d == 128 || d == 256 || d == 512) && hidden > 0 && hidden % 128 == 0, CFM_ERR_UNSUPPORTED);
    CFM_REQUIRE(ln_parts >= 1 && ln_parts <= 16 && (ln_parts & (ln_parts - 1)) == 0, CFM_ERR_UNSUPPORTED);
    CFM_REQUIRE(mode == FFN_PLAIN || mode == FFN_STATS || mode == FFN_LN, CFM_ERR_UNSUPPORTED);
    CFM_REQUIRE(mode != FFN_STATS || stats_out, CFM_ERR_NULL);
    CFM_REQUIRE(mode != FFN_LN || (gamma2 && beta2 && eps2 >= 0.f), CFM_ERR_NULL);
    CFM_REQUIRE(CFM_ALIGNED16(X) && CFM_ALIGNED16(Wp) && CFM_ALIGNED16(b1f) && CFM_ALIGNED16(colsum1) && CFM_ALIGNED16(b2) &&
                CFM_ALIGNED16(Y) && ldx % 4 == 0 && ldy % 4 == 0 && (reinterpret_cast<uintptr_t>(ln_stats) & 7u) == 0 &&
                (reinterpret_cast<uintptr_t>(stats_out) & 7u) == 0, CFM_ERR_ALIGN);
    CFM_REQUIRE(mode != FFN_LN || (CFM_ALIGNED16(gamma2) && CFM_ALIGNED16(beta2)), CFM_ERR_ALIGN);
    if (M == 0) return CFM_OK;
    FfnArgs a{X, ldx, ln_stats, ln_parts, ln_eps, Wp, b1f, colsum1, b2, Y, ldy, stats_out, gamma2, beta2, eps2, M, hidden, alpha,
              ffn_tile_stride(d), g_ffn_rotate, g_ffn_trace_detail, hidden <= 128 * 30 ? g_ffn_trace : nullptr};
    const dim3 grid((unsigned)((M + 31) / 32)), block(256);
    hipStream_t s = static_cast<hipStream_t>(stream);
#define FFN_LAUNCH(ND)                                                                                             \
    do {                                                                                                           \
        if (mode == FFN_PLAIN) hipLaunchKernelGGL((ffn_fused_f32_kernel<ND, FFN_PLAIN>), grid, block, 0, s, a);    \
        else if (mode == FFN_STATS) hipLaunchKernelGGL((ffn_fused_f32_kernel<ND, FFN_STATS>), grid, block, 0, s, a); \
        else hipLaunchKernelGGL((ffn_fused_f32_kernel<ND, FFN_LN>), grid, block, 0, s, a);                         \
    } while (0)
    if (d == 512 && g_ffn_dbg == 1 && mode == FFN_STATS) hipLaunchKernelGGL((ffn_fused_f32_kernel<16, FFN_STATS, 1>), grid, block, 0, s, a);
    else if (d == 512) FFN_LAUNCH(16);
    else if (d == 256) FFN_LAUNCH(8);
    else FFN_LAUNCH(4);
#undef FFN_LAUNCH
    return cfm_launch_status();
}
