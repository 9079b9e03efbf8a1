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
//     holds 8 consecutive k = one 16-byte load of a row-major 16-bit row), ALL loads of a wave are issued before its first
//     MFMA, and the thread's gate-stage operands before that: one memory round trip per step, no LDS staging;
//   * the waves split the contraction; partial 32x32 tiles are summed through LDS.
// One launch per frame (the kernel boundary is the grid-wide barrier of the recurrence), issued back to back from one C call.
// 5.9 us per step at B = 64, H = 640 against 11.0 for the fp32 kernel (tools/lstm_probe.py).
//
// The BACKWARD stays on lstm.hip's fp32 kernel also under autocast.  Its contraction is over the 4H gate rows -- four times
// the forward's -- and its A operand (dG_{t+1}, 4H wide) has to reach every workgroup.  Two 16-bit forms were built and
// measured at B = 64, H = 640 (fp32 kernel: 12.0 us per step): (a) 32-utterance x 16-unit tiles owning the whole contraction:
// 80 workgroups pulling 246 KB each through one CU, 13.4 us; (b) the contraction split over 8 workgroups per tile (40 KB each),
// partial sums added with fp32 atomics, the workgroup drawing the last ticket of a tile doing the gate math: correct, but the
// device-scope release/acquire it needs between workgroups (on this part: an L2 write-back per workgroup, the XCDs' L2s are
// not coherent with each other) made it 31.9 us.  Neither beats the fp32 kernel, which is also the more accurate one.
#include "cfm_common.h"

namespace {

__device__ __forceinline__ float sigmoid_p(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_p(float x) { return 1.0f - 2.0f / (__expf(2.0f * x) + 1.0f); }

struct Lstm16Args {
    const float* gx;              // (B, T, 4H) input projection + biases
    const void* whh16;            // (4H, H) 16-bit
    const int64_t* lengths;       // (B) or null
    float* y;                     // (B, T, H)
    float* c;                     // (B, H) cell state, in place
    void* h16;                    // (2, B, H) 16-bit copy of h_t, double buffered by t & 1
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
    const T16* h_prev = static_cast<const T16*>(a.h16) + (int64_t)((t + 1) & 1) * a.B * H;
    T16* h_next = static_cast<T16*>(a.h16) + (int64_t)(t & 1) * a.B * H;

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
        // A: row li = utterance b0 + li; B: column li = gate column -> W_hh row (li >> 3) * H + u0 + (li & 7)
        const T16* arow = h_prev + (int64_t)min(b0 + li, a.B - 1) * H + 8 * hf;
        const T16* brow = static_cast<const T16*>(a.whh16) + ((int64_t)(li >> 3) * H + min(u0 + (li & 7), H - 1)) * H + 8 * hf;
        const int nstep = H / 16;                                 // MFMA steps of 16 k; wave w takes steps w, w+4, ...
        constexpr int NBAT = 10;
        for (int s0 = wave; s0 < nstep; s0 += 4 * NBAT) {
            x8 av[NBAT], bv[NBAT];
#pragma unroll
            for (int j = 0; j < NBAT; ++j) {
                const int s = min(s0 + 4 * j, nstep - 1);
                av[j] = *reinterpret_cast<const x8*>(arow + 16 * s);
                bv[j] = *reinterpret_cast<const x8*>(brow + 16 * s);
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
    if (!live) {                                                  // beyond the utterance: zero output, state frozen
        *yo = 0.f;
        h_next[(int64_t)b * H + unit] = t > 0 ? h_prev[(int64_t)b * H + unit] : (T16)0.f;      // (never read again for this b)
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
    h_next[(int64_t)b * H + unit] = (T16)hn;
    if (a.save_c) a.save_c[((int64_t)b * a.T + t) * H + unit] = cn;
    if (a.save_gates) {
        float* sg = a.save_gates + ((int64_t)b * a.T + t) * 4 * H + unit;
        sg[0] = ig; sg[H] = fg; sg[2 * H] = gg; sg[3 * H] = og;
    }
}

template <typename T16>
int lstm_fwd16(const Lstm16Args& a, hipStream_t s) {
    const dim3 grid((unsigned)(a.H / 8), (unsigned)((a.B + 31) / 32));
    for (int t = 0; t < a.T; ++t) hipLaunchKernelGGL(lstm_step16_kernel<T16>, grid, dim3(256), 0, s, a, t);
    return cfm_launch_status();
}

}  // namespace

// cfm_lstm_fwd_f32 with the recurrent product on the 16-bit matrix pipe: w_hh16 (4H,H) is the 16-bit copy of W_hh
// (cfm_cast16_f32), h16_scratch holds 2*B*H 16-bit elements.  H % 16 == 0.  Everything else as cfm_lstm_fwd_f32.
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
