// Decoder LSTM recurrence with the h.W_hh^T product on the 16-bit matrix pipe (the arithmetic torch.autocast gives nn.LSTM on
// a GPU: reference decoder.py:10,17-22 under train.py:232's autocast).  Gate math, cell state and every stored tensor stay
// fp32; only the recurrent product's operands are 16-bit (fp32 accumulation).
//
// Why a second pair of kernels: the fp32 step (lstm.hip) is bound by re-streaming W_hh from L2 every frame -- 52 MB per
// step at B = 64, H = 640 (26 MB of W_hh: a workgroup owns 16 utterances, so the matrix is read B/16 times; 26 MB of h: it is
// broadcast to H/4 workgroups), ~8.5 TB/s of L2 -> CU traffic = 6 of the step's 11-12 us (tools/lstm_probe.py).  Here:
//   * W_hh is the per-optimizer-step 16-bit weight copy, h_{t-1} is exchanged between steps through a small 16-bit double
//     buffer: half the bytes;
//   * a workgroup owns 32 utterances x 8 hidden units (32 gate columns): W_hh is read B/32 times, h is broadcast to H/8
//     workgroups: 13 MB per step instead of 52;
//   * operands go straight from global memory into the MFMA register layout (v_mfma_f32_32x32x16: lane (row li, k-half hf)
//     holds 8 consecutive k = one 16-byte load), ALL loads of a wave are issued before its first MFMA, and the thread's
//     gate-stage operands before that: one memory round trip per step, no LDS staging;
//   * the waves split the contraction; partial 32x32 tiles are summed through LDS.
// One launch per frame (the kernel boundary is the grid-wide barrier of the recurrence), issued back to back from one C call.
// Both operands of the recurrent product are stored in MFMA FRAGMENT ORDER (W_hh: re-ordered once per optimizer step on the
// host side; h_t / dG_t: written in that order by the gate threads of the previous step): every wave-level load is one
// contiguous 512 B - 1 KB block.  What the measurements said on the way (B = 64, H = 640, per step; fp32 kernels: 11.0 forward,
// 12.1 backward):
//   * forward, row-major 16-bit operands + the 32 x 8 tiling:                         5.9 us
//   * backward, row-major 16-bit operands, 32 x 16 tiles (80 workgroups):            13.4 us
//   * backward, contraction split over 8 workgroups per tile + fp32 atomics + a ticketed closing workgroup: 31.9 us (correct,
//     but the device-scope release/acquire between workgroups costs an L2 write-back per workgroup: the XCDs' L2s are not
//     coherent with each other)
//   * backward, row-major 16-bit operands with the fp32 kernel's 16 x 16 tiling:      12.9 us -- HALF the bytes of the fp32
//     kernel and the same time: the step is bound by the NUMBER of cache-line pieces a load instruction touches (16 bytes per
//     lane at a 5 KB row stride = 32 pieces per instruction), not by bytes
//   * fragment order:                                          forward 4.4 us, backward 6.7 us.
#include "cfm_common.h"

namespace {

__device__ __forceinline__ float sigmoid_p(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_p(float x) { return 1.0f - 2.0f / (__expf(2.0f * x) + 1.0f); }

struct Lstm16Args {
    const float* gx;              // (B, T, 4H) input projection + biases
    const void* whh16;            // 16-bit W_hh in B-fragment order (H/8, H/16, 2, 32, 8): see cfm_lstm_fwd_mfma16_f32
    const int64_t* lengths;       // (B) or null
    float* y;                     // (B, T, H)
    float* c;                     // (B, H) cell state, in place
    void* h16;                    // 16-bit copy of h_t in A-fragment order, double buffered by t & 1: 2 * ceil(B/32)*32 * H elements
    float* save_gates;            // (B, T, 4H) or null
    float* save_c;                // (B, T, H) or null
    int B, T, H;
};

// Forward step.  grid = (H/8, ceil(B/32)), 256 threads.  Gate column c of the workgroup = 8*q + u  (gate q, unit u0 + u).
// H % 32 == 0 (every wave contracts whole 16-element MFMA steps... H % 16 == 0 suffices; the launcher asks for % 32).
template <typename T16>
__global__ __launch_bounds__(256) void lstm_step16_kernel(const Lstm16Args a, const int t) {
    using x8 = typename Lowp<T16>::x8;
    __shared__ float part[4][32][33];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, hf = lane >> 5;
    const int u0 = blockIdx.x * 8, b0 = blockIdx.y * 32;
    const int H = a.H;
    // h16: (2, ceil(B/32), H/16, 2, 32, 8) -- MFMA A-fragment order (see the header): utterance block, k-step, k-half, row, 8 k
    const int nstep = H / 16;
    const int64_t hbuf = (int64_t)gridDim.y * nstep * 512;
    const T16* h_prev = static_cast<const T16*>(a.h16) + (int64_t)((t + 1) & 1) * hbuf + (int64_t)blockIdx.y * nstep * 512;
    T16* h_next = static_cast<T16*>(a.h16) + (int64_t)(t & 1) * hbuf + (int64_t)blockIdx.y * nstep * 512;

    // gate-stage operands of this thread (utterance tid >> 3, unit tid & 7): requested first
    const int bl = tid >> 3, u = tid & 7;
    const int b = b0 + bl, unit = u0 + u;
    const bool mine = b < a.B && unit < H;
    const int bc = min(b, a.B - 1), uc = min(unit, H - 1);
    const float* gxr = a.gx + ((int64_t)bc * a.T + t) * 4 * H + uc;
    float gxv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) gxv[q] = gxr[(int64_t)q * H];
    const float cprev = t > 0 ? a.c[(int64_t)bc * H + uc] : 0.f;
    const bool live = !a.lengths || t < a.lengths[bc];

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (t > 0) {
        // both operands are stored in MFMA FRAGMENT ORDER: the 64 lanes of a wave read one contiguous 1 KB block per operand and
        // step (lane (li, hf) -> element block (hf * 32 + li) * 8).  The row-major form (16 bytes per lane at a 1-5 KB row
        // stride: 32 cache-line pieces per instruction) was bound by the number of such pieces, not by bytes.
        const T16* afrag = h_prev + (hf * 32 + li) * 8;
        const T16* bfrag = static_cast<const T16*>(a.whh16) + (int64_t)blockIdx.x * nstep * 512 + (hf * 32 + li) * 8;
        constexpr int NBAT = 10;                                  // wave w takes steps w, w+4, ...
        for (int s0 = wave; s0 < nstep; s0 += 4 * NBAT) {
            x8 av[NBAT], bv[NBAT];
#pragma unroll
            for (int j = 0; j < NBAT; ++j) {
                const int s = min(s0 + 4 * j, nstep - 1);
                av[j] = *reinterpret_cast<const x8*>(afrag + s * 512);
                bv[j] = *reinterpret_cast<const x8*>(bfrag + s * 512);
            }
#pragma unroll
            for (int j = 0; j < NBAT; ++j) {
                if (s0 + 4 * j < nstep) acc = Lowp<T16>::mfma(av[j], bv[j], acc);        // (wave-uniform)
            }
        }
    }
    // accumulator: lane li = column, register r = row (r&3) + 8 (r>>2) + 4 hf
#pragma unroll
    for (int r = 0; r < 16; ++r) part[wave][(r & 3) + 8 * (r >> 2) + 4 * hf][li] = acc[r];
    __syncthreads();
    if (!mine) return;
    float* yo = a.y + ((int64_t)b * a.T + t) * H + unit;
    // fragment slot of h[b][unit]: k = unit -> step unit/16, half (unit%16)/8, element unit%8; row bl of the utterance block
    const int hidx = ((unit >> 4) * 2 + ((unit >> 3) & 1)) * 256 + bl * 8 + (unit & 7);
    if (!live) {                                                  // beyond the utterance: zero output, state frozen
        *yo = 0.f;
        h_next[hidx] = (T16)0.f;                                   // (a finished utterance's rows are never read back)
        if (a.save_c) a.save_c[((int64_t)b * a.T + t) * H + unit] = cprev;
        if (a.save_gates) {
            float* sg = a.save_gates + ((int64_t)b * a.T + t) * 4 * H + unit;
            sg[0] = 0.f; sg[H] = 0.f; sg[2 * H] = 0.f; sg[3 * H] = 0.f;
        }
        return;
    }
    float pre[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int cidx = 8 * q + u;
        pre[q] = gxv[q] + ((part[0][bl][cidx] + part[1][bl][cidx]) + (part[2][bl][cidx] + part[3][bl][cidx]));
    }
    const float ig = sigmoid_p(pre[0]), fg = sigmoid_p(pre[1]), gg = tanh_p(pre[2]), og = sigmoid_p(pre[3]);
    const float cn = fg * cprev + ig * gg;
    const float hn = og * tanh_p(cn);
    a.c[(int64_t)b * H + unit] = cn;
    *yo = hn;
    h_next[hidx] = (T16)hn;
    if (a.save_c) a.save_c[((int64_t)b * a.T + t) * H + unit] = cn;
    if (a.save_gates) {
        float* sg = a.save_gates + ((int64_t)b * a.T + t) * 4 * H + unit;
        sg[0] = ig; sg[H] = fg; sg[2 * H] = gg; sg[3 * H] = og;
    }
}

struct Lstm16BwdArgs {
    const float* dy;              // (B, T, H)
    const float* gates;           // (B, T, 4H) saved i|f|g|o
    const float* cells;           // (B, T, H)
    const void* whh_t16;          // 16-bit W_hh^T in B-fragment order (H/16, 4H/16, 2, 16, 8): see cfm_lstm_bwd_mfma16_f32
    const int64_t* lengths;
    float* dgates;                // (B, T, 4H) out
    float* dc;                    // (B, H) running dc_next
    void* dg16;                   // 16-bit copy of dG_t in A-fragment order, double buffered by t & 1: 2 * ceil(B/16)*16 * 4H elements
    int B, T, H;
};

// Backward step t with the fp32 kernel's tiling (16 utterances x 16 units per workgroup, 8 waves split the contraction over
// the 4H gate rows) and 16-bit operands: dG_{t+1} is exchanged through a 16-bit double buffer, W_hh^T is the cached 16-bit
// copy: 164 KB of operands per workgroup instead of 328.  Rows / columns 16..31 of the 32x32 MFMA tile repeat 0..15 (same
// addresses: cache hits) and are not read back.  grid = (H/16, ceil(B/16)), 512 threads.
template <typename T16>
__global__ __launch_bounds__(512) void lstm_bwd_step16_kernel(const Lstm16BwdArgs a, const int t) {
    using x8 = typename Lowp<T16>::x8;
    __shared__ float part[8][16][17];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, hf = lane >> 5;
    const int u0 = blockIdx.x * 16, b0 = blockIdx.y * 16;
    const int H = a.H, H4 = 4 * a.H;
    // dg16: (2, ceil(B/16), 4H/16, 2, 16, 8) -- A-fragment order: utterance block, k-step, k-half, row, 8 k
    const int nstep = H4 / 16;
    const int64_t gbuf = (int64_t)gridDim.y * nstep * 256;
    const T16* dg_next = static_cast<const T16*>(a.dg16) + (int64_t)((t + 1) & 1) * gbuf + (int64_t)blockIdx.y * nstep * 256;
    T16* dg_cur = static_cast<T16*>(a.dg16) + (int64_t)(t & 1) * gbuf + (int64_t)blockIdx.y * nstep * 256;

    // gate-stage operands of this thread (utterance tid >> 4, unit tid & 15; threads 256.. only multiply)
    const int bl = tid >> 4, u = tid & 15;
    const int b = b0 + bl, unit = u0 + u;
    const bool mine = bl < 16 && b < a.B && unit < H;
    const int bc = min(b0 + (bl & 15), a.B - 1), uc = min(unit, H - 1);
    const int64_t bt = (int64_t)bc * a.T + t;
    const bool live = !a.lengths || t < a.lengths[bc];
    const float dyv = a.dy[bt * H + uc];
    const float* sg = a.gates + bt * H4 + uc;
    const float ig = sg[0], fg = sg[H], gg = sg[2 * H], og = sg[3 * H];
    const float ct = a.cells[bt * H + uc];
    const float cprev = t > 0 ? a.cells[(bt - 1) * H + uc] : 0.f;
    const float dcn = t + 1 < a.T ? a.dc[(int64_t)bc * H + uc] : 0.f;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (t + 1 < a.T) {
        // fragment-ordered operands (see the forward): lane (li, hf) -> element block (hf * 16 + (li & 15)) * 8 of a 512-byte step block
        const T16* afrag = dg_next + (hf * 16 + (li & 15)) * 8;
        const T16* bfrag = static_cast<const T16*>(a.whh_t16) + (int64_t)blockIdx.x * nstep * 256 + (hf * 16 + (li & 15)) * 8;
        constexpr int NBAT = 10;                                  // wave w takes steps w, w+8, ...
        for (int s0 = wave; s0 < nstep; s0 += 8 * NBAT) {
            x8 av[NBAT], bv[NBAT];
#pragma unroll
            for (int j = 0; j < NBAT; ++j) {
                const int s = min(s0 + 8 * j, nstep - 1);
                av[j] = *reinterpret_cast<const x8*>(afrag + s * 256);
                bv[j] = *reinterpret_cast<const x8*>(bfrag + s * 256);
            }
#pragma unroll
            for (int j = 0; j < NBAT; ++j) {
                if (s0 + 8 * j < nstep) acc = Lowp<T16>::mfma(av[j], bv[j], acc);        // (wave-uniform)
            }
        }
    }
    // accumulator: lane li = column (unit), register r = row (utterance) (r&3) + 8 (r>>2) + 4 hf: rows < 16 are registers 0..7
    if (li < 16) {
#pragma unroll
        for (int r = 0; r < 8; ++r) part[wave][(r & 3) + 8 * (r >> 2) + 4 * hf][li] = acc[r];
    }
    __syncthreads();
    if (!mine) return;
    float* dg = a.dgates + bt * H4 + unit;
    // fragment slots of dG[b][r], r = q*H + unit: step r/16, half (r%16)/8, element r%8; row bl of the utterance block
    int gidx[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = q * H + unit;
        gidx[q] = ((r >> 4) * 2 + ((r >> 3) & 1)) * 128 + bl * 8 + (r & 7);
    }
    if (!live) {
        dg[0] = 0.f; dg[H] = 0.f; dg[2 * H] = 0.f; dg[3 * H] = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) dg_cur[gidx[q]] = (T16)0.f;
        a.dc[(int64_t)b * H + unit] = 0.f;
        return;
    }
    float dh = dyv;
#pragma unroll
    for (int w = 0; w < 8; ++w) dh += part[w][bl][u];
    const float th = tanh_p(ct);
    const float dct = dh * og * (1.0f - th * th) + dcn;
    const float di = dct * gg * ig * (1.0f - ig), df = dct * cprev * fg * (1.0f - fg);
    const float dgg = dct * ig * (1.0f - gg * gg), dog = dh * th * og * (1.0f - og);
    dg[0] = di; dg[H] = df; dg[2 * H] = dgg; dg[3 * H] = dog;
    dg_cur[gidx[0]] = (T16)di; dg_cur[gidx[1]] = (T16)df; dg_cur[gidx[2]] = (T16)dgg; dg_cur[gidx[3]] = (T16)dog;
    a.dc[(int64_t)b * H + unit] = dct * fg;
}

template <typename T16>
int lstm_bwd16(const Lstm16BwdArgs& a, hipStream_t s) {
    const dim3 grid((unsigned)((a.H + 15) / 16), (unsigned)((a.B + 15) / 16));
    for (int t = a.T - 1; t >= 0; --t) hipLaunchKernelGGL(lstm_bwd_step16_kernel<T16>, grid, dim3(512), 0, s, a, t);
    return cfm_launch_status();
}

template <typename T16>
int lstm_fwd16(const Lstm16Args& a, hipStream_t s) {
    const dim3 grid((unsigned)(a.H / 8), (unsigned)((a.B + 31) / 32));
    for (int t = 0; t < a.T; ++t) hipLaunchKernelGGL(lstm_step16_kernel<T16>, grid, dim3(256), 0, s, a, t);
    return cfm_launch_status();
}

}  // namespace

// cfm_lstm_fwd_f32 with the recurrent product on the 16-bit matrix pipe.  w_hh16: the 16-bit copy of W_hh (4H,H) re-ordered into
// MFMA fragment order (H/8, H/16, 2, 32, 8): element [ub][s][hf][8q+u][e] = W_hh[q*H + 8*ub + u][16*s + 8*hf + e];
// h16_scratch: 2 * ceil(B/32)*32 * H 16-bit elements (any contents).  H % 16 == 0.  Everything else as cfm_lstm_fwd_f32.
extern "C" int cfm_lstm_fwd_mfma16_f32(int prec, const float* gates_x, const void* w_hh16, const int64_t* lengths_or_null, float* y,
                                       float* c_state, void* h16_scratch, float* save_gates_or_null, float* save_c_or_null, int B,
                                       int T, int H, cfm_stream_t stream) {
    CFM_REQUIRE(gates_x && w_hh16 && y && c_state && h16_scratch, CFM_ERR_NULL);
    CFM_REQUIRE(B > 0 && T > 0 && H > 0 && (H & 15) == 0, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(CFM_ALIGNED16(w_hh16) && CFM_ALIGNED16(h16_scratch), CFM_ERR_ALIGN);
    const Lstm16Args a{gates_x, w_hh16, lengths_or_null, y, c_state, h16_scratch, save_gates_or_null, save_c_or_null, B, T, H};
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (prec == CFM_PREC_BF16) return lstm_fwd16<__bf16>(a, s);
    if (prec == CFM_PREC_FP16) return lstm_fwd16<_Float16>(a, s);
    return CFM_ERR_UNSUPPORTED;
}

// cfm_lstm_bwd_f32 with the recurrent product on the 16-bit matrix pipe.  whh_t16: 16-bit W_hh^T in fragment order
// (H/16, 4H/16, 2, 16, 8): element [ub][s][hf][u][e] = W_hh[16*s + 8*hf + e][16*ub + u]; dg16_scratch: 2 * ceil(B/16)*16 * 4H
// 16-bit elements (any contents).  H % 16 == 0.
extern "C" int cfm_lstm_bwd_mfma16_f32(int prec, const float* dy, const float* gates, const float* cells, const void* whh_t16,
                                       const int64_t* lengths_or_null, float* dgates, float* dc_state, void* dg16_scratch, int B,
                                       int T, int H, cfm_stream_t stream) {
    CFM_REQUIRE(dy && gates && cells && whh_t16 && dgates && dc_state && dg16_scratch, CFM_ERR_NULL);
    CFM_REQUIRE(B > 0 && T > 0 && H > 0 && (H & 15) == 0, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(CFM_ALIGNED16(whh_t16) && CFM_ALIGNED16(dg16_scratch), CFM_ERR_ALIGN);
    const Lstm16BwdArgs a{dy, gates, cells, whh_t16, lengths_or_null, dgates, dc_state, dg16_scratch, B, T, H};
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (prec == CFM_PREC_BF16) return lstm_bwd16<__bf16>(a, s);
    if (prec == CFM_PREC_FP16) return lstm_bwd16<_Float16>(a, s);
    return CFM_ERR_UNSUPPORTED;
}

