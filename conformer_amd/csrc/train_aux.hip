// "Next" rows of the scope table: N2 fused multi-tensor Adam step, N4 greedy CTC decode on device.
#include "cfm_common.h"

namespace {

// ---- N2: Adam (torch.optim.Adam defaults: no weight decay, no amsgrad; train.py:188) over MANY tensors per launch ----------
// The tensor table travels in the kernel arguments (no device-side table, no H2D copy, graph-capturable): up to
// ADAM_MAX_T tensors per launch, block -> tensor by a search over the per-tensor first-block prefix.
struct AdamTensor { float* p; const float* g; float* m; float* v; int64_t n; };
constexpr int ADAM_MAX_T = 48;
constexpr int ADAM_CHUNK = 4096;     // elements per workgroup: 256 threads x 4 float4
struct AdamBatch {
    AdamTensor t[ADAM_MAX_T];
    unsigned first_block[ADAM_MAX_T + 1];
    int count;
};

__global__ __launch_bounds__(256) void adam_kernel(const AdamBatch batch, float lr, float beta1, float beta2, float eps,
                                                   float bias_c1, float sqrt_bias_c2) {
    int ti = 0;
    while (ti + 1 < batch.count && batch.first_block[ti + 1] <= blockIdx.x) ++ti;     // uniform: scalar kernarg loads
    const AdamTensor t = batch.t[ti];
    const int64_t base = (int64_t)(blockIdx.x - batch.first_block[ti]) * ADAM_CHUNK;
    const float step_size = lr / bias_c1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t e = base + ((int64_t)i * 256 + threadIdx.x) * 4;
        if (e >= t.n) continue;
        if (e + 3 < t.n && ((reinterpret_cast<uintptr_t>(t.p + e) | reinterpret_cast<uintptr_t>(t.g + e) |
                             reinterpret_cast<uintptr_t>(t.m + e) | reinterpret_cast<uintptr_t>(t.v + e)) & 15) == 0) {
            f32x4 p = *reinterpret_cast<f32x4*>(t.p + e), m = *reinterpret_cast<f32x4*>(t.m + e);
            f32x4 v = *reinterpret_cast<f32x4*>(t.v + e);
            const f32x4 g = *reinterpret_cast<const f32x4*>(t.g + e);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                m[j] = m[j] + (g[j] - m[j]) * (1.0f - beta1);                   // lerp form, as torch's foreach/fused Adam
                v[j] = beta2 * v[j] + (1.0f - beta2) * g[j] * g[j];
                p[j] -= step_size * m[j] / (sqrtf(v[j]) / sqrt_bias_c2 + eps);
            }
            *reinterpret_cast<f32x4*>(t.p + e) = p;
            *reinterpret_cast<f32x4*>(t.m + e) = m;
            *reinterpret_cast<f32x4*>(t.v + e) = v;
        } else {
            for (int64_t k = e; k < t.n && k < e + 4; ++k) {
                const float g = t.g[k];
                const float m = t.m[k] + (g - t.m[k]) * (1.0f - beta1);
                const float v = beta2 * t.v[k] + (1.0f - beta2) * g * g;
                t.m[k] = m; t.v[k] = v;
                t.p[k] -= step_size * m / (sqrtf(v) / sqrt_bias_c2 + eps);
            }
        }
    }
}

// ---- N4: greedy CTC decode (processor.py:301-328): per-frame argmax, drop pad/unk, collapse repeats ---------------------------
// NOTE the reference's collapse does NOT reset on a skipped pad/unk frame ("a <pad> a" -> "a"): reproduced as is.
// Two launches: (1) one wave per frame, lanes over the vocabulary, shuffle arg-max (ties -> lowest index);
// (2) one wave per utterance compacts 64 frames at a time with ballots -- "previous kept-or-skipped id" is the id of
// the nearest earlier non-pad/unk frame, found from the ballot mask, so the collapse needs no serial loop over frames.
__global__ __launch_bounds__(256) void frame_argmax_kernel(const float* __restrict__ logits, int64_t* __restrict__ frame_ids,
                                                           int64_t rows, int V) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* r = logits + row * V;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = lane; c < V; c += 64) {
        const float x = r[c];
        if (x > best || (x == best && c < bi)) { best = x; bi = c; }       // NaN-free logits assumed
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) frame_ids[row] = bi;
}

__global__ __launch_bounds__(64) void ctc_collapse_kernel(const int64_t* __restrict__ frame_ids,
                                                          const int64_t* __restrict__ lengths, int64_t* __restrict__ tokens,
                                                          int64_t* __restrict__ counts, int T, int pad_id, int unk_id) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int64_t* ids = frame_ids + (int64_t)b * T;
    int64_t* out = tokens + (int64_t)b * T;
    const int n = lengths ? (int)min((int64_t)T, max((int64_t)0, lengths[b])) : T;
    int cnt = 0;
    int64_t carry = -1;                                   // id of the last non-skipped frame so far (-1: none yet)
    for (int t0 = 0; t0 < n; t0 += 64) {
        const int t = t0 + lane;
        const int64_t id = t < n ? ids[t] : -1;
        const bool valid = t < n && id != pad_id && id != unk_id;
        const unsigned long long vm = __ballot(valid);
        const unsigned long long below = vm & ((1ull << lane) - 1ull);
        const int src = below ? 63 - __clzll(below) : 0;
        const int64_t prev_in = __shfl(id, src, 64);
        const int64_t prev = below ? prev_in : carry;
        const bool keep = valid && id != prev;
        const unsigned long long km = __ballot(keep);
        if (keep) out[cnt + __popcll(km & ((1ull << lane) - 1ull))] = id;
        cnt += __popcll(km);
        if (vm) carry = __shfl(id, 63 - __clzll(vm), 64);
    }
    for (int t = cnt + lane; t < T; t += 64) out[t] = -1;
    if (lane == 0) counts[b] = cnt;
}

}  // namespace

// One Adam step over n_tensors tensors described by a HOST array of {param, grad, exp_avg, exp_avg_sq, numel} (all device
// fp32 pointers).  bias_c1 = 1 - beta1^t, sqrt_bias_c2 = sqrt(1 - beta2^t) for the step count t AFTER the increment
// (torch semantics).  ceil(n_tensors / 48) launches; the table rides in the kernel arguments.
extern "C" int cfm_adam_step_f32(const cfm_adam_tensor* tensors, int n_tensors, float lr, float beta1, float beta2, float eps,
                                 float bias_c1, float sqrt_bias_c2, cfm_stream_t stream) {
    CFM_REQUIRE(tensors, CFM_ERR_NULL);
    CFM_REQUIRE(n_tensors > 0 && bias_c1 > 0.f && sqrt_bias_c2 > 0.f, CFM_ERR_BAD_SHAPE);
    static_assert(sizeof(cfm_adam_tensor) == sizeof(AdamTensor), "ABI struct and kernel struct must agree");
    for (int i = 0; i < n_tensors; ++i) {
        CFM_REQUIRE(tensors[i].param && tensors[i].grad && tensors[i].exp_avg && tensors[i].exp_avg_sq, CFM_ERR_NULL);
        CFM_REQUIRE(tensors[i].numel > 0, CFM_ERR_BAD_SHAPE);
    }
    for (int t0 = 0; t0 < n_tensors; t0 += ADAM_MAX_T) {
        AdamBatch batch;
        batch.count = n_tensors - t0 < ADAM_MAX_T ? n_tensors - t0 : ADAM_MAX_T;
        unsigned blocks = 0;
        for (int i = 0; i < batch.count; ++i) {
            const cfm_adam_tensor& s = tensors[t0 + i];
            batch.t[i] = AdamTensor{s.param, s.grad, s.exp_avg, s.exp_avg_sq, s.numel};
            batch.first_block[i] = blocks;
            blocks += (unsigned)((s.numel + ADAM_CHUNK - 1) / ADAM_CHUNK);
        }
        batch.first_block[batch.count] = blocks;
        hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), batch, lr, beta1, beta2,
                           eps, bias_c1, sqrt_bias_c2);
        const int st = cfm_launch_status();
        if (st != CFM_OK) return st;
    }
    return CFM_OK;
}

// logits (B,T,V) fp32 -> frame_ids (B,T) int64 per-frame argmax (the "CTC alignment indices"); tokens (B,T) int64 decoded ids
// padded with -1; counts (B) int64.  lengths_or_null: frames to decode per utterance (NULL = all T, as processor.py:301-328).
extern "C" int cfm_greedy_ctc_decode_f32(const float* logits, const int64_t* lengths_or_null, int64_t* frame_ids,
                                         int64_t* tokens, int64_t* counts, int B, int T, int V, int pad_id, int unk_id,
                                         cfm_stream_t stream) {
    CFM_REQUIRE(logits && frame_ids && tokens && counts, CFM_ERR_NULL);
    CFM_REQUIRE(B > 0 && T > 0 && V > 0, CFM_ERR_BAD_SHAPE);
    const int64_t rows = (int64_t)B * T;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(frame_argmax_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, logits, frame_ids, rows, V);
    hipLaunchKernelGGL(ctc_collapse_kernel, dim3((unsigned)B), dim3(64), 0, s, frame_ids, lengths_or_null, tokens, counts, T,
                       pad_id, unk_id);
    return cfm_launch_status();
}
