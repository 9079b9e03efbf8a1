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
};

__device__ __forceinline__ float sigmoid_precise(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_precise(float x) { return 1.0f - 2.0f / (__expf(2.0f * x) + 1.0f); }

template <int RB>                 // 16-row blocks of utterances per workgroup
__global__ __launch_bounds__(256) void lstm_step_kernel(const LstmArgs a, const int t) {
    __shared__ float part[4][RB * 16][17];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, kq = lane >> 4;
    const int u0 = blockIdx.x * 4;                        // hidden units u0..u0+3
    const int b0 = blockIdx.y * (RB * 16);
    const int H = a.H;

    f32x4 acc[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (t > 0) {
        // B operand: column l16 = gate q (l16 >> 2), unit u0 + (l16 & 3)  ->  W_hh row q*H + unit
        const int unit = u0 + (l16 & 3);
        const float* wrow = a.whh + ((int64_t)(l16 >> 2) * H + min(unit, H - 1)) * H;
        const float* hrow[RB];
        bool hok[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int b = b0 + 16 * r + l16;
            hok[r] = b < a.B;
            hrow[r] = a.y + ((int64_t)min(b, a.B - 1) * a.T + (t - 1)) * H;
        }
        const int nchunk = (H + 15) / 16;
        for (int ch = wave; ch < nchunk; ch += 4) {
            const int k = 16 * ch + 4 * kq;
            const bool kok = k < H;                           // H % 4 == 0: a 16-byte chunk is all in or all out
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            const f32x4 wv = (kok && unit < H) ? *reinterpret_cast<const f32x4*>(wrow + k) : z;
            f32x4 hv[RB];
#pragma unroll
            for (int r = 0; r < RB; ++r) hv[r] = (kok && hok[r]) ? *reinterpret_cast<const f32x4*>(hrow[r] + k) : z;
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int r = 0; r < RB; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[r][e], wv[e], acc[r], 0, 0, 0);
        }
    }
    // D layout: lane holds column l16, rows 4*kq + {0..3} of each 16-row block
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) part[wave][16 * r + 4 * kq + i][l16] = acc[r][i];
    __syncthreads();
    const int bl = tid >> 2, u = tid & 3;                  // one thread per (utterance, unit)
    const int b = b0 + bl, unit = u0 + u;
    if (bl >= RB * 16 || b >= a.B || unit >= H) return;
    const bool live = !a.lengths || t < a.lengths[b];
    float* yo = a.y + ((int64_t)b * a.T + t) * H + unit;
    if (!live) {                                           // beyond the utterance: zero output, state frozen
        *yo = 0.f;
        if (a.save_c) a.save_c[((int64_t)b * a.T + t) * H + unit] = a.c[(int64_t)b * H + unit];
        if (a.save_gates) {
            float* sg = a.save_gates + ((int64_t)b * a.T + t) * 4 * H + unit;
            sg[0] = 0.f; sg[H] = 0.f; sg[2 * H] = 0.f; sg[3 * H] = 0.f;
        }
        return;
    }
    const float* gxr = a.gx + ((int64_t)b * a.T + t) * 4 * H + unit;
    float pre[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
        pre[q] = gxr[(int64_t)q * H] + ((part[0][bl][4 * q + u] + part[1][bl][4 * q + u]) +
                                        (part[2][bl][4 * q + u] + part[3][bl][4 * q + u]));
    const float ig = sigmoid_precise(pre[0]), fg = sigmoid_precise(pre[1]), gg = tanh_precise(pre[2]),
                og = sigmoid_precise(pre[3]);
    const float cprev = t > 0 ? a.c[(int64_t)b * H + unit] : 0.f;
    const float cn = fg * cprev + ig * gg;
    a.c[(int64_t)b * H + unit] = cn;
    *yo = og * tanh_precise(cn);
    if (a.save_c) a.save_c[((int64_t)b * a.T + t) * H + unit] = cn;
    if (a.save_gates) {
        float* sg = a.save_gates + ((int64_t)b * a.T + t) * 4 * H + unit;
        sg[0] = ig; sg[H] = fg; sg[2 * H] = gg; sg[3 * H] = og;
    }
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

}  // namespace

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
    const LstmArgs a{gates_x, w_hh, lengths_or_null, y, c_state, save_gates_or_null, save_c_or_null, B, T, H};
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int rb = B <= 16 ? 1 : (B <= 32 ? 2 : 4);
    const dim3 grid((unsigned)(H / 4), (unsigned)((B + 16 * rb - 1) / (16 * rb)));
    for (int t = 0; t < T; ++t) {
        if (rb == 1) hipLaunchKernelGGL(lstm_step_kernel<1>, grid, dim3(256), 0, s, a, t);
        else if (rb == 2) hipLaunchKernelGGL(lstm_step_kernel<2>, grid, dim3(256), 0, s, a, t);
        else hipLaunchKernelGGL(lstm_step_kernel<4>, grid, dim3(256), 0, s, a, t);
    }
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
