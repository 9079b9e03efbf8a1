// Decoder LSTM forward (reference: nn.LSTM(batch_first) over a packed sequence, decoder.py:10,17-22).
//
// The input projection X.W_ih^T + b_ih + b_hh is one dense GEMM done by the caller (gemm_f32 / gemm_mfma16); this file
// is the recurrence: for every frame t,  gates = gates_x[:,t] + h_{t-1}.W_hh^T;  i,f,o = sigmoid, g = tanh (PyTorch gate
// order i|f|g|o);  c = f*c + i*g;  h = o*tanh(c).  Utterance b stops at lengths[b]: later outputs are 0 (what
// pad_packed_sequence returns) and its state is frozen.
//
// One launch per frame, issued back to back from one C call (the kernel boundary is the grid-wide barrier the
// recurrence needs; no persistent kernel, no spin-wait).  A workgroup owns 4 hidden units = 16 gate rows of W_hh
// (16 x H floats, L2-resident across steps) for up to 64 utterances: its 4 waves split the contraction over H, each
// running v_mfma_f32_16x16x4_f32 on 16-byte loads of h_{t-1} and W_hh (k-order permuted identically for both
// operands), partial tiles are summed through LDS, then one thread per (utterance, unit) applies the gate math.
// h_{t-1} is read straight from the output tensor y[:, t-1, :].
#include "cfm_common.h"

namespace {

struct LstmArgs {
    const float* gx;              // (B, T, 4H)
    const float* whh;             // (4H, H)
    const int64_t* lengths;       // (B) or null
    float* y;                     // (B, T, H)  h_t
    float* c;                     // (B, H)     cell state, in place
    float* save_gates;            // (B, T, 4H) post-activation i|f|g|o, or null
    float* save_c;                // (B, T, H)  c_t, or null
    int B, T, H;
    float* hfrag;                 // FRAG kernels: 2 x (16-utterance blocks) x H, h_t in operand-fragment order (double buffer)
    unsigned long long* trace;    // diagnostics: s_memrealtime stamps (8 per step) of thread 0 of workgroups (0,0) and (last,last)
};

__device__ __forceinline__ float sigmoid_precise(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_precise(float x) { return 1.0f - 2.0f / (__expf(2.0f * x) + 1.0f); }

// FRAG: both contraction operands are stored in the order the MFMA lanes consume them -- per 16-wide chunk of the
// contraction a [kq 4][row 16][4 floats] block, so that a wave's 16-byte-per-lane load is ONE contiguous 1 KB run instead
// of 16 rows x 64 B at a row stride (the step is bound by memory round trips, and a load that touches 16 lines completes
// later than one that touches 8 consecutive ones).  W_hh is re-ordered by the caller once per call; h_t is written in
// that order by the gate stage into a double buffer (a.hfrag) next to the row-major output y.  Needs H % 16 == 0.
template <int RB, bool FRAG>      // 16-row blocks of utterances per workgroup
__global__ __launch_bounds__(256) void lstm_step_kernel(const LstmArgs a, const int t) {
    __shared__ float part[4][RB * 16][17];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, kq = lane >> 4;
    const int u0 = blockIdx.x * 4;                        // hidden units u0..u0+3
    const int b0 = blockIdx.y * (RB * 16);
    const int H = a.H;
    const bool first = blockIdx.x == 0 && blockIdx.y == 0, lastwg = blockIdx.x == gridDim.x - 1 && blockIdx.y == gridDim.y - 1;
    unsigned long long* tr = (a.trace && tid == 0 && (first || lastwg)) ? a.trace + ((int64_t)t * 2 + (lastwg ? 1 : 0)) * 8 : nullptr;
#define LSTM_STAMP(i) do { if (tr) tr[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
    LSTM_STAMP(0);

    // Gate-stage operands of this thread (one thread per (utterance, unit)): requested BEFORE the contraction so that their
    // latency overlaps it -- the step is a chain of dependent memory round trips, not arithmetic.
    const int bl = tid >> 2, u = tid & 3;
    const int b = b0 + bl, unit = u0 + u;
    const bool mine = bl < RB * 16 && b < a.B && unit < H;
    const int bc = min(b, a.B - 1), uc = min(unit, H - 1);
    const float* gxr = a.gx + ((int64_t)bc * a.T + t) * 4 * H + uc;
    float gxv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) gxv[q] = gxr[(int64_t)q * H];
    const float cprev = t > 0 ? a.c[(int64_t)bc * H + uc] : 0.f;
    const bool live = !a.lengths || t < a.lengths[bc];

    f32x4 acc[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (FRAG && t > 0) {
        const int nchunk = H >> 4;
        const float* wf = a.whh + ((int64_t)blockIdx.x * nchunk) * 256 + lane * 4;
        const float* hf[RB];
        const int64_t hpar = (int64_t)((t - 1) & 1) * gridDim.y * RB * 16 * H;
#pragma unroll
        for (int r = 0; r < RB; ++r) hf[r] = a.hfrag + hpar + ((int64_t)(blockIdx.y * RB + r) * nchunk) * 256 + lane * 4;
        constexpr int NBAT = 5;
        for (int c0 = wave; c0 < nchunk; c0 += 4 * NBAT) {
            f32x4 wv[NBAT], hv[NBAT][RB];
#pragma unroll
            for (int j = 0; j < NBAT; ++j) {
                const int ch = min(c0 + 4 * j, nchunk - 1);
                wv[j] = *reinterpret_cast<const f32x4*>(wf + ch * 256);
#pragma unroll
                for (int r = 0; r < RB; ++r) hv[j][r] = *reinterpret_cast<const f32x4*>(hf[r] + ch * 256);
            }
#pragma unroll
            for (int j = 0; j < NBAT; ++j) {
                const f32x4 w = c0 + 4 * j < nchunk ? wv[j] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int r = 0; r < RB; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[j][r][e], w[e], acc[r], 0, 0, 0);
            }
        }
    }
    if (!FRAG && t > 0) {
        // B operand: column l16 = gate q (l16 >> 2), unit u0 + (l16 & 3)  ->  W_hh row q*H + unit
        // (columns of units >= H and rows of utterances >= B are computed on clamped addresses and never read back)
        const float* wrow = a.whh + ((int64_t)(l16 >> 2) * H + min(u0 + (l16 & 3), H - 1)) * H;
        const float* hrow[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) hrow[r] = a.y + ((int64_t)min(b0 + 16 * r + l16, a.B - 1) * a.T + (t - 1)) * H;
        const int nchunk = (H + 15) / 16;
        constexpr int NBAT = 5;                               // chunks in flight per wave: ALL loads of a batch are issued first
        for (int c0 = wave; c0 < nchunk; c0 += 4 * NBAT) {
            f32x4 wv[NBAT], hv[NBAT][RB];
#pragma unroll
            for (int j = 0; j < NBAT; ++j) {
                const int k = 16 * (c0 + 4 * j) + 4 * kq;     // H % 4 == 0: a 16-byte chunk is all in or all out
                const int kc = min(k, H - 4);
                wv[j] = *reinterpret_cast<const f32x4*>(wrow + kc);
#pragma unroll
                for (int r = 0; r < RB; ++r) hv[j][r] = *reinterpret_cast<const f32x4*>(hrow[r] + kc);
            }
#pragma unroll
            for (int j = 0; j < NBAT; ++j) {
                const int k = 16 * (c0 + 4 * j) + 4 * kq;
                const f32x4 w = k < H ? wv[j] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int r = 0; r < RB; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[j][r][e], w[e], acc[r], 0, 0, 0);
            }
        }
    }
    // D layout: lane holds column l16, rows 4*kq + {0..3} of each 16-row block
    LSTM_STAMP(1);
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) part[wave][16 * r + 4 * kq + i][l16] = acc[r][i];
    LSTM_STAMP(2);
    __syncthreads();
    LSTM_STAMP(3);
    if (!mine) return;
    float* yo = a.y + ((int64_t)b * a.T + t) * H + unit;
    // fragment-order slot of h_t[b][unit]: block (utterance block, unit >> 4), lane (kq = (unit >> 2) & 3, row b & 15), float unit & 3
    float* hfo = FRAG ? a.hfrag + (int64_t)(t & 1) * gridDim.y * RB * 16 * H + ((int64_t)(b >> 4) * (H >> 4) + (unit >> 4)) * 256 +
                            ((((unit >> 2) & 3) * 16 + (b & 15)) * 4 + (unit & 3))
                      : nullptr;
    if (!live) {                                           // beyond the utterance: zero output, state frozen
        *yo = 0.f;
        if (FRAG) *hfo = 0.f;
        if (a.save_c) a.save_c[((int64_t)b * a.T + t) * H + unit] = cprev;
        if (a.save_gates) {
            float* sg = a.save_gates + ((int64_t)b * a.T + t) * 4 * H + unit;
            sg[0] = 0.f; sg[H] = 0.f; sg[2 * H] = 0.f; sg[3 * H] = 0.f;
        }
        return;
    }
    float pre[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
        pre[q] = gxv[q] + ((part[0][bl][4 * q + u] + part[1][bl][4 * q + u]) + (part[2][bl][4 * q + u] + part[3][bl][4 * q + u]));
    const float ig = sigmoid_precise(pre[0]), fg = sigmoid_precise(pre[1]), gg = tanh_precise(pre[2]),
                og = sigmoid_precise(pre[3]);
    const float cn = fmaf(fg, cprev, ig * gg);             // explicit: the same contraction in every instantiation
    a.c[(int64_t)b * H + unit] = cn;
    const float hn = og * tanh_precise(cn);
    *yo = hn;
    if (FRAG) *hfo = hn;
    if (a.save_c) a.save_c[((int64_t)b * a.T + t) * H + unit] = cn;
    if (a.save_gates) {
        float* sg = a.save_gates + ((int64_t)b * a.T + t) * 4 * H + unit;
        sg[0] = ig; sg[H] = fg; sg[2 * H] = gg; sg[3 * H] = og;
    }
    LSTM_STAMP(4);
    if (tr) { __builtin_amdgcn_s_waitcnt(0); tr[5] = __builtin_amdgcn_s_memrealtime(); }
#undef LSTM_STAMP
}

// ---- backward through time ---------------------------------------------------------------------------------------------
// Step t (T-1 .. 0), one launch:  dh_rec[b,u] = sum_r dG_{t+1}[b,r] * W_hh[r,u]   (dG = gradients of the gate
// pre-activations; contraction over the 4H gate rows, W_hh^T rows are contiguous in `whh_t`), then for the workgroup's
// 16 hidden units:  dh = dy_t + dh_rec;  do = dh*tanh(c_t)*o(1-o);  dc = dh*o*(1-tanh^2 c_t) + dc_next;
// di = dc*g*i(1-i);  df = dc*c_{t-1}*f(1-f);  dg = dc*i*(1-g^2);  dc_next = dc*f.   Frames beyond an utterance's length
// carry no gradient (dG = 0, dc = 0).  8 waves split the contraction; partial 16x16 tiles are summed through LDS.
struct LstmBwdArgs {
    const float* dy;              // (B, T, H)
    const float* gates;           // (B, T, 4H) saved i|f|g|o
    const float* cells;           // (B, T, H)  saved c_t
    const float* whh_t;           // (H, 4H) = W_hh^T
    const int64_t* lengths;
    float* dgates;                // (B, T, 4H) out
    float* dc;                    // (B, H) running dc_next, in place
    int B, T, H;
    float* dgfrag;                // FRAG kernels: 2 x (16-utterance blocks) x 4H, dG_t in operand-fragment order (double buffer)
};

template <int RB, bool FRAG>      // FRAG: dG_{t+1} and W_hh^T in operand-fragment order (see the forward)
__global__ __launch_bounds__(512) void lstm_bwd_step_kernel(const LstmBwdArgs a, const int t) {
    __shared__ float part[8][RB * 16][17];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, kq = lane >> 4;
    const int u0 = blockIdx.x * 16;
    const int b0 = blockIdx.y * (RB * 16);
    const int H = a.H, H4 = 4 * a.H;

    // gate-stage operands of this thread (one (utterance, unit) per thread), requested before the contraction (see the forward)
    static_assert(RB * 256 <= 512, "one (utterance, unit) per thread");
    const int bl = tid >> 4, u = tid & 15;
    const int b = b0 + bl, unit = u0 + u;
    const bool mine = tid < RB * 256 && b < a.B && unit < H;
    const int bc = min(b, a.B - 1), uc = min(unit, H - 1);
    const int64_t bt = (int64_t)bc * a.T + t;
    const bool live = !a.lengths || t < a.lengths[bc];
    const float dy = a.dy[bt * H + uc];
    const float* sg = a.gates + bt * H4 + uc;
    const float ig = sg[0], fg = sg[H], gg = sg[2 * H], og = sg[3 * H];
    const float ct = a.cells[bt * H + uc];
    const float cprev = t > 0 ? a.cells[(bt - 1) * H + uc] : 0.f;
    const float dcn = t + 1 < a.T ? a.dc[(int64_t)bc * H + uc] : 0.f;

    f32x4 acc[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (FRAG && t + 1 < a.T) {
        const int nchunk = H4 >> 4;
        const float* wf = a.whh_t + ((int64_t)blockIdx.x * nchunk) * 256 + lane * 4;
        const float* gf[RB];
        const int64_t gpar = (int64_t)((t + 1) & 1) * gridDim.y * RB * 16 * H4;
#pragma unroll
        for (int r = 0; r < RB; ++r) gf[r] = a.dgfrag + gpar + ((int64_t)(blockIdx.y * RB + r) * nchunk) * 256 + lane * 4;
        constexpr int NBAT = 5;
        for (int c0 = wave; c0 < nchunk; c0 += 8 * NBAT) {
            f32x4 wv[NBAT], gv[NBAT][RB];
#pragma unroll
            for (int j = 0; j < NBAT; ++j) {
                const int ch = min(c0 + 8 * j, nchunk - 1);
                wv[j] = *reinterpret_cast<const f32x4*>(wf + ch * 256);
#pragma unroll
                for (int r = 0; r < RB; ++r) gv[j][r] = *reinterpret_cast<const f32x4*>(gf[r] + ch * 256);
            }
#pragma unroll
            for (int j = 0; j < NBAT; ++j) {
                const f32x4 w = c0 + 8 * j < nchunk ? wv[j] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int r = 0; r < RB; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(gv[j][r][e], w[e], acc[r], 0, 0, 0);
            }
        }
    }
    if (!FRAG && t + 1 < a.T) {
        // (columns of units >= H and rows of utterances >= B are computed on clamped addresses and never read back)
        const float* wrow = a.whh_t + (int64_t)min(u0 + l16, H - 1) * H4;
        const float* grow[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) grow[r] = a.dgates + ((int64_t)min(b0 + 16 * r + l16, a.B - 1) * a.T + (t + 1)) * H4;
        const int nchunk = H4 / 16;                              // H % 4 == 0 -> 4H % 16 == 0
        constexpr int NBAT = 5;                                  // chunks in flight per wave: all loads of a batch are issued first
        for (int c0 = wave; c0 < nchunk; c0 += 8 * NBAT) {
            f32x4 wv[NBAT], gv[NBAT][RB];
#pragma unroll
            for (int j = 0; j < NBAT; ++j) {
                const int k = 16 * min(c0 + 8 * j, nchunk - 1) + 4 * kq;
                wv[j] = *reinterpret_cast<const f32x4*>(wrow + k);
#pragma unroll
                for (int r = 0; r < RB; ++r) gv[j][r] = *reinterpret_cast<const f32x4*>(grow[r] + k);
            }
#pragma unroll
            for (int j = 0; j < NBAT; ++j) {
                const f32x4 w = c0 + 8 * j < nchunk ? wv[j] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int r = 0; r < RB; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(gv[j][r][e], w[e], acc[r], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) part[wave][16 * r + 4 * kq + i][l16] = acc[r][i];
    __syncthreads();
    if (!mine) return;
    float* dg = a.dgates + bt * H4 + unit;
    // fragment-order slots of dG_t[b][q*H + unit]: row r = q*H + unit -> block (utterance block, r >> 4), lane ((r >> 2) & 3, b & 15), float r & 3
    float* dgf = nullptr;
    int fo[4] = {0, 0, 0, 0};
    if (FRAG) {
        dgf = a.dgfrag + (int64_t)(t & 1) * gridDim.y * RB * 16 * H4 + (int64_t)(b >> 4) * (H4 >> 4) * 256 + (b & 15) * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = q * H + unit;
            fo[q] = (r >> 4) * 256 + ((r >> 2) & 3) * 64 + (r & 3);
        }
    }
    if (!live) {
        dg[0] = 0.f; dg[H] = 0.f; dg[2 * H] = 0.f; dg[3 * H] = 0.f;
        if (FRAG) { dgf[fo[0]] = 0.f; dgf[fo[1]] = 0.f; dgf[fo[2]] = 0.f; dgf[fo[3]] = 0.f; }
        a.dc[(int64_t)b * H + unit] = 0.f;
        return;
    }
    float dh = dy;
#pragma unroll
    for (int w = 0; w < 8; ++w) dh += part[w][bl][u];
    const float th = tanh_precise(ct);
    const float dct = fmaf(dh * og, fmaf(-th, th, 1.0f), dcn);   // explicit contractions: identical in every instantiation
    const float d0 = dct * gg * ig * (1.0f - ig), d1 = dct * cprev * fg * (1.0f - fg), d2 = dct * ig * fmaf(-gg, gg, 1.0f),
                d3 = dh * th * og * (1.0f - og);
    dg[0] = d0; dg[H] = d1; dg[2 * H] = d2; dg[3 * H] = d3;
    if (FRAG) { dgf[fo[0]] = d0; dgf[fo[1]] = d1; dgf[fo[2]] = d2; dgf[fo[3]] = d3; }
    a.dc[(int64_t)b * H + unit] = dct * fg;
}

// y = BatchNorm1d(eval)(swish(h)) per channel (decoder.py:23-26 with running statistics)
__global__ __launch_bounds__(256) void swish_bn_eval_kernel(const float* __restrict__ h, const float* __restrict__ mean,
                                                            const float* __restrict__ var, const float* __restrict__ w,
                                                            const float* __restrict__ b, float eps, float* __restrict__ out,
                                                            int64_t n4, int C) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const int c = (int)((4 * i) % C);
    const f32x4 x = reinterpret_cast<const f32x4*>(h)[i];
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e)
        o[e] = (swishf_acc(x[e]) - mean[c + e]) * (1.0f / sqrtf(var[c + e] + eps)) * w[c + e] + b[c + e];
    reinterpret_cast<f32x4*>(out)[i] = o;
}

// ---- Swish + BatchNorm1d in train mode (decoder.py:23-26 under .train()): batch statistics over all B*T rows (padded
// frames included, as the reference does), running-statistics update, and the coupled backward.
//   RMODE 0: out0[c] += sum swish(h)            RMODE 1: out0[c] += sum (swish(h) - mean[c])^2
//   RMODE 2: out0[c] (dgamma) += sum dz*xhat ;  out1[c] (dbeta) += sum dz          xhat = (swish(h) - mean[c]) * inv[c]
// block = 64 channels (16 lanes x float4) x 16 row lanes x 64 rows; LDS combine; one atomic per channel per workgroup.
template <int RMODE>
__global__ __launch_bounds__(256) void swish_bn_reduce_kernel(const float* __restrict__ h, const float* __restrict__ dz,
                                                              const float* __restrict__ mean, const float* __restrict__ var,
                                                              float eps, int64_t rows, int C, float* __restrict__ out0,
                                                              float* __restrict__ out1) {
    __shared__ float red[2][16][64];
    const int cq = threadIdx.x & 15, ry = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + cq * 4;
    const int64_t r0 = (int64_t)blockIdx.y * 64, r1 = min(rows, r0 + 64);
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
    if (c < C) {
        f32x4 mu = {0.f, 0.f, 0.f, 0.f}, inv = {1.f, 1.f, 1.f, 1.f};
        if (RMODE >= 1) mu = *reinterpret_cast<const f32x4*>(mean + c);
        if (RMODE == 2) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(var + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) inv[e] = 1.0f / sqrtf(v[e] + eps);
        }
        for (int64_t r = r0 + ry; r < r1; r += 16) {
            const f32x4 x = *reinterpret_cast<const f32x4*>(h + r * C + c);
            f32x4 d = {0.f, 0.f, 0.f, 0.f};
            if (RMODE == 2) d = *reinterpret_cast<const f32x4*>(dz + r * C + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float s = swishf_acc(x[e]);
                if (RMODE == 0) a0[e] += s;
                if (RMODE == 1) a0[e] += (s - mu[e]) * (s - mu[e]);
                if (RMODE == 2) { a0[e] += d[e] * (s - mu[e]) * inv[e]; a1[e] += d[e]; }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[0][ry][cq * 4 + e] = a0[e]; red[1][ry][cq * 4 + e] = a1[e]; }
    __syncthreads();
    const int cx = threadIdx.x & 63, which = threadIdx.x >> 6;
    const int cc = blockIdx.x * 64 + cx;
    if (which < (RMODE == 2 ? 2 : 1) && cc < C) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) s += red[which][j][cx];
        atomicAdd((which == 0 ? out0 : out1) + cc, s);
    }
}

__global__ void dec_bn_mean_kernel(float* __restrict__ sum_to_mean, float inv_n, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) sum_to_mean[c] *= inv_n;
}
__global__ void dec_bn_var_kernel(float* __restrict__ m2_to_var, const float* __restrict__ mean, float* __restrict__ run_mean,
                                  float* __restrict__ run_var, float inv_n, float unbias, float momentum, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float var = m2_to_var[c] * inv_n;
    m2_to_var[c] = var;
    if (run_mean) run_mean[c] = (1.0f - momentum) * run_mean[c] + momentum * mean[c];
    if (run_var) run_var[c] = (1.0f - momentum) * run_var[c] + momentum * var * unbias;
}

// dh = swish'(h) * gamma*inv * (dz - inv_n*(dbeta + xhat*dgamma))      (inv_n = 0: fixed statistics)
__global__ __launch_bounds__(256) void swish_bn_bwd_kernel(const float* __restrict__ h, const float* __restrict__ dz,
                                                           const float* __restrict__ mean, const float* __restrict__ var,
                                                           const float* __restrict__ w, const float* __restrict__ dgamma,
                                                           const float* __restrict__ dbeta, float eps, float inv_n,
                                                           float* __restrict__ dh, int64_t n4, int C) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const int c = (int)((4 * i) % C);
    const f32x4 x = reinterpret_cast<const f32x4*>(h)[i];
    const f32x4 d = reinterpret_cast<const f32x4*>(dz)[i];
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float sg = sigmoidf_acc(x[e]);
        const float s = x[e] * sg;
        const float inv = 1.0f / sqrtf(var[c + e] + eps);
        const float xh = (s - mean[c + e]) * inv;
        const float ds = w[c + e] * inv * (d[e] - inv_n * (dbeta[c + e] + xh * dgamma[c + e]));
        o[e] = ds * sg * (1.0f + x[e] * (1.0f - sg));
    }
    reinterpret_cast<f32x4*>(dh)[i] = o;
}

}  // namespace

static unsigned long long* g_lstm_trace = nullptr;
// diagnostics: T x 2 x 8 uint64 stamps (100 MHz) of the forward steps; NULL switches the trace off
extern "C" int cfm_debug_lstm_trace(void* trace_or_null) {
    g_lstm_trace = static_cast<unsigned long long*>(trace_or_null);
    return CFM_OK;
}

// gates_x (B,T,4H) = X.W_ih^T + b_ih + b_hh (caller's GEMM); w_hh (4H,H) PyTorch layout (gate order i|f|g|o);
// lengths_or_null (B) int64: frames per utterance (pack_padded_sequence); y (B,T,H) <- h_t (0 beyond the length);
// c_state (B,H) scratch for the cell state (need not be initialised); save_gates_or_null (B,T,4H) / save_c_or_null (B,T,H):
// activations for a backward pass.  H % 4 == 0.  Enqueues T launches.
extern "C" int cfm_lstm_fwd_f32(const float* gates_x, const float* w_hh, const int64_t* lengths_or_null, float* y,
                                float* c_state, float* save_gates_or_null, float* save_c_or_null, int B, int T, int H,
                                cfm_stream_t stream) {
    CFM_REQUIRE(gates_x && w_hh && y && c_state, CFM_ERR_NULL);
    CFM_REQUIRE(B > 0 && T > 0 && H > 0 && (H & 3) == 0, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(CFM_ALIGNED16(w_hh) && CFM_ALIGNED16(y), CFM_ERR_ALIGN);
    const LstmArgs a{gates_x, w_hh, lengths_or_null, y, c_state, save_gates_or_null, save_c_or_null, B, T, H, nullptr, g_lstm_trace};
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)(H / 4), (unsigned)((B + 15) / 16));
    for (int t = 0; t < T; ++t) hipLaunchKernelGGL((lstm_step_kernel<1, false>), grid, dim3(256), 0, s, a, t);
    return cfm_launch_status();
}

// cfm_lstm_fwd_f32 with both recurrence operands in MFMA-fragment order (H % 16 == 0): w_hh_frag = W_hh re-ordered to
// [H/4 unit blocks][H/16 chunks][kq 4][gate q 4][unit 4][4 floats]  (element W_hh[q*H + 4*ub + u][16*ch + 4*kq + e]);
// h_frag_scratch: 2 * ceil(B/16)*16 * H floats (need not be initialised).  Same results as cfm_lstm_fwd_f32 bit for bit
// (same products, same summation order).
extern "C" int cfm_lstm_fwd_frag_f32(const float* gates_x, const float* w_hh_frag, const int64_t* lengths_or_null, float* y,
                                     float* c_state, float* h_frag_scratch, float* save_gates_or_null, float* save_c_or_null,
                                     int B, int T, int H, cfm_stream_t stream) {
    CFM_REQUIRE(gates_x && w_hh_frag && y && c_state && h_frag_scratch, CFM_ERR_NULL);
    CFM_REQUIRE(B > 0 && T > 0 && H > 0 && (H & 15) == 0, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(CFM_ALIGNED16(w_hh_frag) && CFM_ALIGNED16(h_frag_scratch), CFM_ERR_ALIGN);
    const LstmArgs a{gates_x, w_hh_frag, lengths_or_null, y, c_state, save_gates_or_null, save_c_or_null, B, T, H, h_frag_scratch,
                     g_lstm_trace};
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)(H / 4), (unsigned)((B + 15) / 16));
    for (int t = 0; t < T; ++t) hipLaunchKernelGGL((lstm_step_kernel<1, true>), grid, dim3(256), 0, s, a, t);
    return cfm_launch_status();
}

// Backward of cfm_lstm_fwd_f32 through time: dy (B,T,H) = gradient w.r.t. y; gates / cells as saved by the forward;
// whh_t (H,4H) = W_hh transposed; dgates (B,T,4H) <- gradient w.r.t. the gate pre-activations (= w.r.t. gates_x: the
// caller's GEMMs turn it into dX, dW_ih, dW_hh, db); dc_state (B,H) scratch.  Enqueues T launches (t = T-1 .. 0).
extern "C" int cfm_lstm_bwd_f32(const float* dy, const float* gates, const float* cells, const float* whh_t,
                                const int64_t* lengths_or_null, float* dgates, float* dc_state, int B, int T, int H,
                                cfm_stream_t stream) {
    CFM_REQUIRE(dy && gates && cells && whh_t && dgates && dc_state, CFM_ERR_NULL);
    CFM_REQUIRE(B > 0 && T > 0 && H > 0 && (H & 3) == 0, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(CFM_ALIGNED16(whh_t) && CFM_ALIGNED16(dgates), CFM_ERR_ALIGN);
    const LstmBwdArgs a{dy, gates, cells, whh_t, lengths_or_null, dgates, dc_state, B, T, H, nullptr};
    hipStream_t s = static_cast<hipStream_t>(stream);
    // 16 utterances x 16 hidden units per workgroup: the step is latency-bound, so more, smaller workgroups win
    const dim3 grid((unsigned)((H + 15) / 16), (unsigned)((B + 15) / 16));
    for (int t = T - 1; t >= 0; --t) hipLaunchKernelGGL((lstm_bwd_step_kernel<1, false>), grid, dim3(512), 0, s, a, t);
    return cfm_launch_status();
}

// cfm_lstm_bwd_f32 with both operands of the recurrent product in MFMA-fragment order (H % 16 == 0): whh_t_frag = W_hh^T
// re-ordered to [H/16 unit blocks][4H/16 chunks][kq 4][unit 16][4 floats]  (element W_hh^T[16*ub + u][16*ch + 4*kq + e]);
// dg_frag_scratch: 2 * ceil(B/16)*16 * 4H floats.  Same results as cfm_lstm_bwd_f32 bit for bit.
extern "C" int cfm_lstm_bwd_frag_f32(const float* dy, const float* gates, const float* cells, const float* whh_t_frag,
                                     const int64_t* lengths_or_null, float* dgates, float* dc_state, float* dg_frag_scratch,
                                     int B, int T, int H, cfm_stream_t stream) {
    CFM_REQUIRE(dy && gates && cells && whh_t_frag && dgates && dc_state && dg_frag_scratch, CFM_ERR_NULL);
    CFM_REQUIRE(B > 0 && T > 0 && H > 0 && (H & 15) == 0, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(CFM_ALIGNED16(whh_t_frag) && CFM_ALIGNED16(dg_frag_scratch), CFM_ERR_ALIGN);
    const LstmBwdArgs a{dy, gates, cells, whh_t_frag, lengths_or_null, dgates, dc_state, B, T, H, dg_frag_scratch};
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)(H / 16), (unsigned)((B + 15) / 16));
    for (int t = T - 1; t >= 0; --t) hipLaunchKernelGGL((lstm_bwd_step_kernel<1, true>), grid, dim3(512), 0, s, a, t);
    return cfm_launch_status();
}

extern "C" int cfm_swish_bn_eval_f32(const float* h, const float* bn_mean, const float* bn_var, const float* bn_weight,
                                     const float* bn_bias, float eps, float* out, int64_t rows, int C, cfm_stream_t stream) {
    CFM_REQUIRE(h && bn_mean && bn_var && bn_weight && bn_bias && out, CFM_ERR_NULL);
    CFM_REQUIRE(rows > 0 && C > 0 && (C & 3) == 0, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(CFM_ALIGNED16(h) && CFM_ALIGNED16(out), CFM_ERR_ALIGN);
    const int64_t n4 = rows * C / 4;
    hipLaunchKernelGGL(swish_bn_eval_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       h, bn_mean, bn_var, bn_weight, bn_bias, eps, out, n4, C);
    return cfm_launch_status();
}

// Batch statistics of swish(h) over `rows` rows (decoder.py:25 in .train()): batch_mean, batch_var (biased); the running
// buffers (may be NULL) are updated with `momentum` and the unbiased variance.  Normalise with cfm_swish_bn_eval_f32
// (bn_mean = batch_mean, bn_var = batch_var).  C % 4 == 0.
extern "C" int cfm_swish_bn_stats_f32(const float* h, float* batch_mean, float* batch_var, float* running_mean_or_null,
                                      float* running_var_or_null, float momentum, int64_t rows, int C, cfm_stream_t stream) {
    CFM_REQUIRE(h && batch_mean && batch_var, CFM_ERR_NULL);
    CFM_REQUIRE(rows > 0 && C > 0 && (C & 3) == 0, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(CFM_ALIGNED16(h) && CFM_ALIGNED16(batch_mean), CFM_ERR_ALIGN);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)((C + 63) / 64), (unsigned)((rows + 63) / 64));
    const unsigned cb = (unsigned)((C + 255) / 256);
    if (hipMemsetAsync(batch_mean, 0, sizeof(float) * C, s) != hipSuccess) return CFM_ERR_LAUNCH;
    if (hipMemsetAsync(batch_var, 0, sizeof(float) * C, s) != hipSuccess) return CFM_ERR_LAUNCH;
    hipLaunchKernelGGL(swish_bn_reduce_kernel<0>, grid, dim3(256), 0, s, h, nullptr, nullptr, nullptr, 0.f, rows, C, batch_mean,
                       nullptr);
    hipLaunchKernelGGL(dec_bn_mean_kernel, dim3(cb), dim3(256), 0, s, batch_mean, 1.0f / (float)rows, C);
    hipLaunchKernelGGL(swish_bn_reduce_kernel<1>, grid, dim3(256), 0, s, h, nullptr, batch_mean, nullptr, 0.f, rows, C, batch_var,
                       nullptr);
    hipLaunchKernelGGL(dec_bn_var_kernel, dim3(cb), dim3(256), 0, s, batch_var, batch_mean, running_mean_or_null,
                       running_var_or_null, 1.0f / (float)rows, rows > 1 ? (float)rows / (float)(rows - 1) : 1.0f, momentum, C);
    return cfm_launch_status();
}

// Backward of z = BatchNorm(swish(h)): dh, dgamma, dbeta (the latter two accumulated: caller zero-fills).  train_stats = 1:
// bn_mean / bn_var are the batch statistics and the mean/variance coupling is applied; 0: running statistics (constants).
extern "C" int cfm_swish_bn_bwd_f32(const float* h, const float* dz, const float* bn_mean, const float* bn_var,
                                    const float* bn_weight, float eps, int train_stats, float* dh, float* dgamma,
                                    float* dbeta, int64_t rows, int C, cfm_stream_t stream) {
    CFM_REQUIRE(h && dz && bn_mean && bn_var && bn_weight && dh && dgamma && dbeta, CFM_ERR_NULL);
    CFM_REQUIRE(rows > 0 && C > 0 && (C & 3) == 0, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(CFM_ALIGNED16(h) && CFM_ALIGNED16(dz) && CFM_ALIGNED16(dh) && CFM_ALIGNED16(bn_mean) && CFM_ALIGNED16(bn_var),
                CFM_ERR_ALIGN);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)((C + 63) / 64), (unsigned)((rows + 63) / 64));
    hipLaunchKernelGGL(swish_bn_reduce_kernel<2>, grid, dim3(256), 0, s, h, dz, bn_mean, bn_var, eps, rows, C, dgamma, dbeta);
    const int64_t n4 = rows * C / 4;
    hipLaunchKernelGGL(swish_bn_bwd_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, h, dz, bn_mean, bn_var,
                       bn_weight, dgamma, dbeta, eps, train_stats ? 1.0f / (float)rows : 0.f, dh, n4, C);
    return cfm_launch_status();
}
