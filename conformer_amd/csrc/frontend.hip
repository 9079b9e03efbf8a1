// Audio front end of the path (processing/processor.py:53-63,155-158,373-394 and processing/augment.py:7-19):
// log-mel filterbank features and SpecAugment band masking.  The arithmetic the reference delegates to
// torchaudio 2.1.0 (MelSpectrogram: centre reflect padding, periodic Hann window, 400-point power STFT, 80 slaney mel
// filters, then log(clamp(., 1e-5))) is restated here as
//     reflect_pad  ->  framing * window * DFT as ONE batched MFMA GEMM over overlapping rows (lda = hop; the window is
//                      folded into the (402 x 400) cos/-sin basis; gemm_bwd_f32.hip)  ->  power + mel + log (this file).
// SpecAugment: the band positions are drawn on the host exactly where torchaudio draws them (torch.rand on the CPU
// generator); the kernel applies the time / frequency bands in place.
#include "cfm_common.h"

namespace {

// xp[b][i] = x[b][reflect(i - pad)], i in [0, L + 2*pad)   (torch.stft center=True, pad_mode="reflect")
__global__ __launch_bounds__(256) void reflect_pad_kernel(const float* __restrict__ x, float* __restrict__ xp, int B,
                                                          int64_t L, int pad, int64_t Lp) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)B * Lp) return;
    const int64_t b = idx / Lp, i = idx - b * Lp;
    int64_t j = i - pad;
    if (j < 0) j = -j;
    if (j >= L) j = 2 * (L - 1) - j;
    xp[idx] = (i < L + 2 * pad) ? x[b * L + j] : 0.f;
}

// spec: (B*T, lds) rows = frames, columns [0,NB) = Re, [NB,2NB) = Im of the NB one-sided bins.
// out[b][m][t] = log(max(sum_k fb[k][m] * (Re^2 + Im^2), floor)).  A workgroup handles 32 frames of one utterance:
// the 32 x 2NB spectrum tile is staged through LDS (coalesced loads), reduced to powers in place, then thread
// (t = tid & 31, m = tid >> 5 + 8 j) accumulates its mel bins; stores are 128-byte runs along t.
template <int NB>
__global__ __launch_bounds__(256) void power_mel_log_kernel(const float* __restrict__ spec, int64_t lds_,
                                                            const float* __restrict__ fb, int n_mels, float floor_,
                                                            float* __restrict__ out, int T) {
    __shared__ float tile[32 * (2 * NB + 1)];
    constexpr int ROW = 2 * NB + 1;
    const int b = blockIdx.y, t0 = blockIdx.x * 32;
    const int nt = min(32, T - t0);
    for (int i = threadIdx.x; i < 32 * 2 * NB; i += 256) {
        const int r = i / (2 * NB), c = i - r * (2 * NB);
        tile[r * ROW + c] = r < nt ? spec[((int64_t)b * T + t0 + r) * lds_ + c] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 32 * NB; i += 256) {
        const int r = i / NB, k = i - r * NB;
        const float re = tile[r * ROW + k], im = tile[r * ROW + NB + k];
        tile[r * ROW + k] = re * re + im * im;
    }
    __syncthreads();
    const int t = threadIdx.x & 31;
    for (int m = threadIdx.x >> 5; m < n_mels; m += 8) {
        float acc = 0.f;
        for (int k = 0; k < NB; ++k) acc = fmaf(fb[k * n_mels + m], tile[t * ROW + k], acc);
        if (t < nt) out[((int64_t)b * n_mels + m) * T + t0 + t] = logf(fmaxf(acc, floor_));
    }
}

// Round 3: power + mel + log on the fp32 matrix pipe.  The first form (above) multiplied the 201 x 80 filterbank on the VALU with
// one LDS read and one global read per FMA: ~220 of the front end's 359 us.  Here a workgroup takes 32 frames of one utterance;
// the spectrum row layout is [Re 0..200 | 0-pad to 208 | Im 0..200 | 0-pad to 416] (the DFT basis is stored with those zero rows,
// so Re and Im fragments are 16-byte aligned).  mel^T[m][t] = sum_bin fbT[m][bin] * power[bin][t] on v_mfma_f32_32x32x2_f32: the
// B operand (lane = frame t, half hf, bins 8c+4hf+e at step e) is computed in registers from two 16-byte loads (Re, Im); the A
// operand is the zero-padded transposed filterbank fbT (96 x 208) in the same k order.  The four waves split the 26 eight-bin
// groups (7,7,6,6), their partial 96 x 32 tiles are summed through LDS in a fixed order, then log(max(., floor)) and 128-byte
// runs along t.  Memory-bound: the spectrum is read once (53 MB at B = 32, 10 s).
__global__ __launch_bounds__(256) void power_mel_log_mfma_kernel(const float* __restrict__ spec, int64_t lds_, int im_off,
                                                                 const float* __restrict__ fbT, int nk, int n_mels, float floor_,
                                                                 float* __restrict__ out, int T, int rpu) {
    __shared__ __attribute__((aligned(16))) float part[4 * 96 * 32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, hf = lane >> 5;
    const int b = blockIdx.y, t0 = blockIdx.x * 32;
    const int nt = min(32, T - t0);
    const int ngroups = nk >> 3;                                        // 8-bin groups (nk = 208 -> 26)
    const int g0 = (ngroups * wave) >> 2, g1 = (ngroups * (wave + 1)) >> 2;
    const float* row = spec + ((int64_t)b * rpu + t0 + min(li, nt - 1)) * lds_ + 4 * hf;     // (frames beyond nt: clamped, never stored)
    f32x16 acc[3];
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    for (int c = g0; c < g1; ++c) {
        const f32x4 re = *reinterpret_cast<const f32x4*>(row + 8 * c);
        const f32x4 im = *reinterpret_cast<const f32x4*>(row + im_off + 8 * c);
        f32x4 a[3];
#pragma unroll
        for (int m = 0; m < 3; ++m) a[m] = *reinterpret_cast<const f32x4*>(fbT + (int64_t)(32 * m + li) * nk + 8 * c + 4 * hf);
        const f32x4 pw = re * re + im * im;
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int m = 0; m < 3; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][e], pw[e], acc[m], 0, 0, 0);
    }
    float* mine = part + wave * 96 * 32;
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) mine[(32 * m + (r & 3) + 8 * (r >> 2) + 4 * hf) * 32 + li] = acc[m][r];
    __syncthreads();
    for (int i = tid; i < 96 * 32; i += 256) {
        const int m = i >> 5, t = i & 31;
        const float v = ((part[i] + part[96 * 32 + i]) + part[2 * 96 * 32 + i]) + part[3 * 96 * 32 + i];
        if (m < n_mels && t < nt) out[((int64_t)b * n_mels + m) * T + t0 + t] = logf(fmaxf(v, floor_));
    }
}

// bands: (n, 3) int32 rows {axis (1 = frequency, 2 = time), start, end}; every band masks all utterances (iid_masks=False)
__global__ __launch_bounds__(256) void specaugment_kernel(float* __restrict__ spec, int B, int F, int T,
                                                          const int* __restrict__ bands, int nbands, float value) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)B * F * T) return;
    const int t = (int)(idx % T);
    const int f = (int)((idx / T) % F);
    bool hit = false;
    for (int i = 0; i < nbands; ++i) {
        const int ax = bands[3 * i], s = bands[3 * i + 1], e = bands[3 * i + 2];
        const int p = ax == 1 ? f : t;
        hit |= (p >= s && p < e);
    }
    if (hit) spec[idx] = value;
}

}  // namespace

extern "C" int cfm_reflect_pad_f32(const float* x, float* xp, int B, int64_t L, int pad, int64_t ld_out,
                                   cfm_stream_t stream) {
    CFM_REQUIRE(x && xp, CFM_ERR_NULL);
    CFM_REQUIRE(B > 0 && L > pad && pad >= 0 && ld_out >= L + 2 * pad, CFM_ERR_BAD_SHAPE);
    const int64_t total = (int64_t)B * ld_out;
    hipLaunchKernelGGL(reflect_pad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, xp, B, L, pad, ld_out);
    return cfm_launch_status();
}

extern "C" int cfm_power_mel_log_f32(const float* spec, int64_t ld_spec, const float* fb, float* out, int B, int T,
                                     int n_bins, int n_mels, float floor_value, cfm_stream_t stream) {
    CFM_REQUIRE(spec && fb && out, CFM_ERR_NULL);
    CFM_REQUIRE(B > 0 && T > 0 && n_mels > 0 && ld_spec >= 2 * n_bins, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(n_bins == 201, CFM_ERR_UNSUPPORTED);            // n_fft = 400 (processor.py:19)
    const dim3 grid((unsigned)((T + 31) / 32), (unsigned)B);
    hipLaunchKernelGGL(power_mel_log_kernel<201>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), spec, ld_spec, fb,
                       n_mels, floor_value, out, T);
    return cfm_launch_status();
}

// spec rows: [Re bins 0..n_bins-1 | zeros to nk | Im bins | zeros to 2 nk], nk = n_bins rounded up to 8 (208); fbT: (96, nk) fp32 =
// the filterbank transposed, zero rows / columns beyond (n_mels, n_bins).  ld_spec >= 2 nk, ld_spec % 4 == 0, n_mels <= 96.
// rows_per_utt >= T: spectrum rows of utterance b start at row b * rows_per_utt (cfm_dft_frames_f32's layout).
extern "C" int cfm_power_mel_log_mfma_f32(const float* spec, int64_t ld_spec, const float* fbT, float* out, int B, int T,
                                          int rows_per_utt, int n_bins, int n_mels, float floor_value, cfm_stream_t stream) {
    CFM_REQUIRE(spec && fbT && out, CFM_ERR_NULL);
    const int nk = (n_bins + 7) / 8 * 8;
    CFM_REQUIRE(B > 0 && T > 0 && rows_per_utt >= T && n_mels > 0 && n_bins > 0 && ld_spec >= 2 * nk && (ld_spec & 3) == 0, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(n_mels <= 96 && B <= 65535, CFM_ERR_UNSUPPORTED);
    CFM_REQUIRE(CFM_ALIGNED16(spec) && CFM_ALIGNED16(fbT), CFM_ERR_ALIGN);
    const dim3 grid((unsigned)((T + 31) / 32), (unsigned)B);
    hipLaunchKernelGGL(power_mel_log_mfma_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), spec, ld_spec, nk, fbT, nk,
                       n_mels, floor_value, out, T, rows_per_utt);
    return cfm_launch_status();
}

extern "C" int cfm_specaugment_apply_f32(float* spec, int B, int F, int T, const int* bands, int nbands, float value,
                                         cfm_stream_t stream) {
    CFM_REQUIRE(spec && (bands || nbands == 0), CFM_ERR_NULL);
    CFM_REQUIRE(B > 0 && F > 0 && T > 0 && nbands >= 0, CFM_ERR_BAD_SHAPE);
    if (nbands == 0) return CFM_OK;
    const int64_t total = (int64_t)B * F * T;
    hipLaunchKernelGGL(specaugment_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), spec, B, F, T, bands, nbands, value);
    return cfm_launch_status();
}
