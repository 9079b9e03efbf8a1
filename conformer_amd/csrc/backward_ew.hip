// Bandwidth-bound backward kernels of the Conformer block: LayerNorm, column sums (bias gradients), GLU, and the
// depthwise-conv + BatchNorm(eval) + Swish core.  All fp32, (rows, channels) row-major, lanes along channels.
// Parameter gradients are accumulated with fp32 atomics into caller-zeroed buffers (one atomic per wave-partial per
// channel: contention is negligible next to the streaming reads).
#include "cfm_common.h"

namespace {

// ---- LayerNorm backward, input gradient: one wave per row (same mapping as the forward kernel) --------------------
//   xhat = (x-mean)*rstd, g = dy*gamma;  dx = rstd*(g - mean_d(g) - xhat*mean_d(g*xhat)) [+ dres]
template <int VPL>
__global__ __launch_bounds__(256) void layernorm_bwd_dx_kernel(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ dy,
    const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ dres,
    float* __restrict__ dx, int64_t rows, int d) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nvec = d >> 2;
    const f32x4* xr = reinterpret_cast<const f32x4*>(x + row * d);
    const f32x4* dyr = reinterpret_cast<const f32x4*>(dy + row * d);
    const f32x4* g4 = reinterpret_cast<const f32x4*>(gamma);
    const float mu = mean[row], rs = rstd[row];
    f32x4 xh[VPL], gg[VPL];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = lane + i * 64;
        if (c < nvec) {
            xh[i] = (xr[c] - mu) * rs;
            gg[i] = dyr[c] * g4[c];
            s1 += (gg[i].x + gg[i].y) + (gg[i].z + gg[i].w);
            s2 += (gg[i].x * xh[i].x + gg[i].y * xh[i].y) + (gg[i].z * xh[i].z + gg[i].w * xh[i].w);
        }
    }
    s1 = wave_sum(s1) / (float)d;
    s2 = wave_sum(s2) / (float)d;
    f32x4* dxr = reinterpret_cast<f32x4*>(dx + row * d);
    const f32x4* rr = dres ? reinterpret_cast<const f32x4*>(dres + row * d) : nullptr;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = lane + i * 64;
        if (c < nvec) {
            f32x4 v = (gg[i] - s1 - xh[i] * s2) * rs;
            if (rr) v = v + rr[c];
            dxr[c] = v;
        }
    }
}

// ---- LayerNorm backward, input AND parameter gradients in one pass (training path) --------------------------------------
// The two-kernel form reads x and dy twice (dx kernel, then the dgamma/dbeta column reduction).  Here a wave walks
// `rows_per_wave` rows (4-row stride inside the workgroup), keeps sum_r dy*xhat and sum_r dy for its columns in registers,
// the four waves combine through LDS and the workgroup issues one atomic per column per output.  The next row's operands
// are requested before the current row is reduced (a wave has two rows in flight).
template <int VPL>
__global__ __launch_bounds__(256) void layernorm_bwd_fused_kernel(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ dy,
    const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ dres,
    float* __restrict__ dx, float* __restrict__ partials, int64_t rows, int d, int rows_per_wave) {
    __shared__ f32x4 red[2][3][VPL][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nvec = d >> 2;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_wave * 4 + wave;
    const int64_t rend = min(rows, (int64_t)(blockIdx.x + 1) * rows_per_wave * 4);
    const f32x4* g4 = reinterpret_cast<const f32x4*>(gamma);
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 gam[VPL], dgs[VPL], dbs[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = lane + i * 64;
        gam[i] = c < nvec ? g4[c] : z4;
        dgs[i] = z4; dbs[i] = z4;
    }
    f32x4 xn[VPL], dn[VPL], rn[VPL];
    float mun = 0.f, rsn = 0.f;
    auto fetch = [&](int64_t row) {
        const int64_t rc = min(row, rows - 1);
        const f32x4* xr = reinterpret_cast<const f32x4*>(x + rc * d);
        const f32x4* dyr = reinterpret_cast<const f32x4*>(dy + rc * d);
        const f32x4* rr = reinterpret_cast<const f32x4*>((dres ? dres : dy) + rc * d);
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = min(lane + i * 64, nvec - 1);
            xn[i] = xr[c]; dn[i] = dyr[c];
            if (dres) rn[i] = rr[c];
        }
        mun = mean[rc]; rsn = rstd[rc];
    };
    if (r0 < rend) fetch(r0);
    for (int64_t row = r0; row < rend; row += 4) {
        f32x4 xh[VPL], dyv[VPL], res[VPL];
        const float rs = rsn;
#pragma unroll
        for (int i = 0; i < VPL; ++i) { xh[i] = (xn[i] - mun) * rsn; dyv[i] = dn[i]; res[i] = dres ? rn[i] : z4; }
        if (row + 4 < rend) fetch(row + 4);
        float s1 = 0.f, s2 = 0.f;
        f32x4 gg[VPL];
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const bool ok = lane + i * 64 < nvec;
            gg[i] = ok ? dyv[i] * gam[i] : z4;
            if (!ok) { xh[i] = z4; dyv[i] = z4; }
            s1 += (gg[i].x + gg[i].y) + (gg[i].z + gg[i].w);
            s2 += (gg[i].x * xh[i].x + gg[i].y * xh[i].y) + (gg[i].z * xh[i].z + gg[i].w * xh[i].w);
            dgs[i] = dgs[i] + dyv[i] * xh[i];
            dbs[i] = dbs[i] + dyv[i];
        }
        s1 = wave_sum(s1) / (float)d;
        s2 = wave_sum(s2) / (float)d;
        f32x4* dxr = reinterpret_cast<f32x4*>(dx + row * d);
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = lane + i * 64;
            if (c < nvec) dxr[c] = (gg[i] - s1 - xh[i] * s2) * rs + res[i];
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < VPL; ++i) { red[0][wave - 1][i][lane] = dgs[i]; red[1][wave - 1][i][lane] = dbs[i]; }
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = lane + i * 64;
            if (c >= nvec) continue;
            const f32x4 a = (dgs[i] + red[0][0][i][lane]) + (red[0][1][i][lane] + red[0][2][i][lane]);
            const f32x4 b = (dbs[i] + red[1][0][i][lane]) + (red[1][1][i][lane] + red[1][2][i][lane]);
            f32x4* prow = reinterpret_cast<f32x4*>(partials + (int64_t)blockIdx.x * 2 * d);      // [workgroup][dgamma | dbeta]
            prow[c] = a;
            prow[nvec + c] = b;
        }
    }
}

// Second stage: out[c] += sum over the workgroups' partial rows (fixed order: the result does not depend on scheduling).
// grid.x = 2d/64 column blocks (16 lanes x float4) x 16 row lanes; columns [0,d) -> dgamma, [d,2d) -> dbeta.
__global__ __launch_bounds__(256) void layernorm_bwd_partials_kernel(const float* __restrict__ partials, int nrows, int d,
                                                                    float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ f32x4 red[16][16];
    const int cq = threadIdx.x & 15, ry = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + cq * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (c < 2 * d) {
#pragma unroll 4
        for (int r = ry; r < nrows; r += 16) acc = acc + *reinterpret_cast<const f32x4*>(partials + (int64_t)r * 2 * d + c);
    }
    red[ry][cq] = acc;
    __syncthreads();
    if (ry == 0 && c < 2 * d) {
#pragma unroll
        for (int j = 1; j < 16; ++j) acc = acc + red[j][cq];
        float* out = c < d ? dgamma + c : dbeta + (c - d);
#pragma unroll
        for (int e = 0; e < 4; ++e) out[e] += acc[e];
    }
}

// ---- LayerNorm backward, parameter gradients; also the generic column-sum (MODE 0) --------------------------------
//   MODE 0: out0[c] += alpha * sum_r X[r][c]
//   MODE 1: out0[c] (dgamma) += sum_r dy*xhat ; out1[c] (dbeta) += sum_r dy       (X = x, Y = dy)
// block = 64 columns (16 lanes x float4) x 16 row-lanes; each block walks `rows_per_block` rows with 16-byte loads; LDS
// combine; one atomic per column per block.  VEC = false: scalar loads for unaligned / ragged column counts.
template <int MODE, bool VEC>
__global__ __launch_bounds__(256) void colreduce_kernel(const float* __restrict__ X, const float* __restrict__ Y,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        int64_t ld, int64_t rows, int cols, int rows_per_block,
                                                        float alpha, float* __restrict__ out0, float* __restrict__ out1) {
    constexpr int RL = VEC ? 16 : 4;                       // row lanes
    __shared__ float red[2][RL][64];
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    const int64_t r1 = min(rows, r0 + rows_per_block);
    if (VEC) {
        const int cq = threadIdx.x & 15, ry = threadIdx.x >> 4;
        const int c = blockIdx.x * 64 + cq * 4;
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
        if (c < cols) {
#pragma unroll 4
            for (int64_t r = r0 + ry; r < r1; r += RL) {
                if (MODE == 0) {
                    a0 = a0 + *reinterpret_cast<const f32x4*>(X + r * ld + c);
                } else {
                    const f32x4 dyv = *reinterpret_cast<const f32x4*>(Y + r * ld + c);
                    const f32x4 xv = *reinterpret_cast<const f32x4*>(X + r * ld + c);
                    const float mu = mean[r], rs = rstd[r];
#pragma unroll
                    for (int e = 0; e < 4; ++e) a0[e] += dyv[e] * (xv[e] - mu) * rs;
                    a1 = a1 + dyv;
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { red[0][ry][cq * 4 + e] = a0[e]; red[1][ry][cq * 4 + e] = a1[e]; }
    } else {
        const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
        const int c = blockIdx.x * 64 + cx;
        float a0 = 0.f, a1 = 0.f;
        if (c < cols) {
            for (int64_t r = r0 + ry; r < r1; r += RL) {
                if (MODE == 0) {
                    a0 += X[r * ld + c];
                } else {
                    const float dyv = Y[r * ld + c];
                    a0 += dyv * (X[r * ld + c] - mean[r]) * rstd[r];
                    a1 += dyv;
                }
            }
        }
        red[0][ry][cx] = a0;
        red[1][ry][cx] = a1;
    }
    __syncthreads();
    const int cx = threadIdx.x & 63, which = threadIdx.x >> 6;          // wave 0 -> out0, wave 1 -> out1
    const int c = blockIdx.x * 64 + cx;
    if (which < (MODE == 1 ? 2 : 1) && c < cols) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < RL; ++j) s += red[which][j][cx];
        if (which == 0) atomicAdd(out0 + c, (MODE == 0 ? alpha : 1.0f) * s);
        else atomicAdd(out1 + c, s);
    }
}

template <int MODE>
void colreduce_launch(const float* X, const float* Y, const float* mean, const float* rstd, int64_t ld, int64_t rows, int cols,
                      float alpha, float* out0, float* out1, hipStream_t s) {
    const bool vec = (cols & 3) == 0 && (ld & 3) == 0 && CFM_ALIGNED16(X) && (!Y || CFM_ALIGNED16(Y));
    const int rpb = vec ? 64 : 128;
    const dim3 grid((unsigned)((cols + 63) / 64), (unsigned)((rows + rpb - 1) / rpb));
    if (vec) hipLaunchKernelGGL((colreduce_kernel<MODE, true>), grid, dim3(256), 0, s, X, Y, mean, rstd, ld, rows, cols, rpb, alpha, out0, out1);
    else hipLaunchKernelGGL((colreduce_kernel<MODE, false>), grid, dim3(256), 0, s, X, Y, mean, rstd, ld, rows, cols, rpb, alpha, out0, out1);
}

// ---- GLU forward on a stored pre-activation (training path: z is kept for the backward) --------------------------------
__global__ __launch_bounds__(256) void glu_fwd_kernel(const float* __restrict__ z, float* __restrict__ y, int64_t rows,
                                                      int n) {
    const int nv = n >> 2;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * nv) return;
    const int64_t r = idx / nv;
    const int c = (int)(idx - r * nv) * 4;
    const f32x4 a = *reinterpret_cast<const f32x4*>(z + r * 2 * n + c);
    const f32x4 gt = *reinterpret_cast<const f32x4*>(z + r * 2 * n + n + c);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = a[e] * sigmoidf_acc(gt[e]);
    *reinterpret_cast<f32x4*>(y + r * n + c) = o;
}

// ---- GLU backward: y = a * sigmoid(g), z = [a | g] (rows, 2n) -> dz (rows, 2n) ----------------------------------------
// TOUT = float, or a 16-bit matrix-pipe type: dz only feeds the pointwise_conv_1 gradient GEMMs (and its bias gradient).
template <typename TOUT>
__global__ __launch_bounds__(256) void glu_bwd_kernel(const float* __restrict__ z, const float* __restrict__ dy,
                                                      TOUT* __restrict__ dz, int64_t rows, int n) {
    const int nv = n >> 2;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * nv) return;
    const int64_t r = idx / nv;
    const int c = (int)(idx - r * nv) * 4;
    const f32x4 a = *reinterpret_cast<const f32x4*>(z + r * 2 * n + c);
    const f32x4 gt = *reinterpret_cast<const f32x4*>(z + r * 2 * n + n + c);
    const f32x4 d = *reinterpret_cast<const f32x4*>(dy + r * n + c);
    f32x4 da, dg;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float sg = sigmoidf_acc(gt[e]);
        da[e] = d[e] * sg;
        dg[e] = d[e] * a[e] * sg * (1.0f - sg);
    }
    if constexpr (sizeof(TOUT) == 4) {
        *reinterpret_cast<f32x4*>(dz + r * 2 * n + c) = da;
        *reinterpret_cast<f32x4*>(dz + r * 2 * n + n + c) = dg;
    } else {
        *reinterpret_cast<typename Lowp<TOUT>::x4*>(dz + r * 2 * n + c) = Lowp<TOUT>::cvt4(da);
        *reinterpret_cast<typename Lowp<TOUT>::x4*>(dz + r * 2 * n + n + c) = Lowp<TOUT>::cvt4(dg);
    }
}

// ---- depthwise conv + BatchNorm + Swish, backward ---------------------------------------------------------------------
// forward: c = b + sum_j w[j] g[t+j-H];  xhat = (c - mu)*inv;  u = xhat*gamma + beta;  y = u*sigmoid(u)
// (mu, inv) are either the running statistics (eval) or the batch statistics (train).
// pass A: recompute c,u from g; du = dy*swish'(u); dcf = du*inv*gamma -> dc tensor; dbeta += sum du, dgamma += sum du*xhat
// pass B: train only: dc = dcf - gamma*inv/n * (dbeta + xhat*dgamma)   (the batch-statistics coupling), in place;
//         both modes: dbias += sum dc, dw[j] += sum_t dc[t]*g[t+j-H]
// pass C (dwconv_plain_kernel with flipped taps): dg[t] = sum_j w[j] dc[t-j+H]
// Work split: grid (C/64, time segments, B); a workgroup owns 64 channels x one segment of one utterance, its 4 waves
// stride over the segment in TT-frame chunks keeping the per-channel partial sums (and the K tap gradients) in
// registers; the 4 waves are combined through LDS and ONE atomic per (channel[, tap]) per workgroup is issued --
// the first version issued one per wave per chunk and spent 0.9 ms per layer in atomic contention.
template <int K, int TT, int PASS>
__global__ __launch_bounds__(256) void dwconv_bn_swish_bwd_kernel(
    const float* __restrict__ g, const float* __restrict__ dy, const float* __restrict__ w,
    const float* __restrict__ bias, const float* __restrict__ bn_w, const float* __restrict__ bn_b,
    const float* __restrict__ bn_mean, const float* __restrict__ bn_var, float eps, float* __restrict__ dc,
    float* __restrict__ dw, float* __restrict__ dbias, float* __restrict__ dgamma, float* __restrict__ dbeta,
    int T, int C, int seg_len, float inv_n /* 0 = fixed statistics */) {
    constexpr int HALF = (K - 1) / 2;
    constexpr int NRED = PASS == 0 ? 2 : K + 1;
    __shared__ float red[4][NRED][65];                 // 65: the tap-major read-out below walks q at fixed channel
    __shared__ float taps[64 * K];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int b = blockIdx.z;
    const int seg0 = blockIdx.y * seg_len, seg1 = min(T, seg0 + seg_len);
    const bool cok = c < C;
    const int cc = cok ? c : C - 1;
    float wr[K], dwacc[K];
    load_taps<K>(w, blockIdx.x * 64, C, taps, wr);
#pragma unroll
    for (int j = 0; j < K; ++j) dwacc[j] = 0.f;
    const float inv = 1.0f / sqrtf(bn_var[cc] + eps);
    const float mu = bn_mean[cc], ga = bn_w[cc], be = bn_b[cc], bi = bias[cc];
    const float k1 = PASS == 1 ? ga * inv * inv_n * dbeta[cc] : 0.f;      // sums are complete: pass A has finished
    const float k2 = PASS == 1 ? ga * inv * inv_n * dgamma[cc] : 0.f;
    const float* gb = g + (int64_t)b * T * C + cc;
    const float* dyb = dy + (int64_t)b * T * C + cc;
    float* dcb = dc + (int64_t)b * T * C + cc;
    float s_du = 0.f, s_dux = 0.f, s_dc = 0.f;
    for (int t0 = seg0 + wave * TT; t0 < seg1; t0 += 4 * TT) {
        float gwin[TT + K - 1], dwin[TT];                 // dwin: dy (pass A) or dc (pass B) of the chunk's TT frames
        load_window<TT + K - 1>(gb, t0 - HALF, T, C, gwin);
        load_window<TT>(PASS == 0 ? dyb : dcb, t0, seg1, C, dwin);
#pragma unroll
        for (int o = 0; o < TT; ++o) {
            const int t = t0 + o;
            const bool tok = t < seg1;
            float cv = bi;
#pragma unroll
            for (int j = 0; j < K; ++j) cv = fmaf(wr[j], gwin[o + j], cv);
            const float xh = (cv - mu) * inv;
            if (PASS == 0) {
                const float u = xh * ga + be;
                const float sg = sigmoidf_acc(u);
                const float dyv = dwin[o];
                const float du = dyv * sg * (1.0f + u * (1.0f - sg));
                if (tok && cok) dcb[(int64_t)t * C] = du * inv * ga;
                s_du += du; s_dux += du * xh;
            } else {
                float dcv = dwin[o];
                if (tok) dcv -= k1 + xh * k2;
                if (tok && cok && inv_n != 0.f) dcb[(int64_t)t * C] = dcv;
                s_dc += dcv;
#pragma unroll
                for (int j = 0; j < K; ++j) dwacc[j] = fmaf(dcv, gwin[o + j], dwacc[j]);
            }
        }
    }
    if (PASS == 0) { red[wave][0][lane] = s_du; red[wave][1][lane] = s_dux; }
    else {
        red[wave][K][lane] = s_dc;
#pragma unroll
        for (int j = 0; j < K; ++j) red[wave][j][lane] = dwacc[j];
    }
    __syncthreads();
    if (PASS == 0) {
        if (wave < 2 && cok)
            atomicAdd((wave == 0 ? dbeta : dgamma) + c,
                      (red[0][wave][lane] + red[1][wave][lane]) + (red[2][wave][lane] + red[3][wave][lane]));
    } else {
        // dw is (C,K): the 64 x K gradients of this workgroup are one contiguous run -- walk it in memory order so every
        // atomic instruction covers whole 256-byte segments (scattered lanes run ~17x slower, MI355X_MICROARCH.md)
        const int64_t base = (int64_t)blockIdx.x * 64 * K, lim = (int64_t)C * K;
        for (int f = threadIdx.x; f < 64 * K; f += 256) {
            const int cl = f / K, q = f - cl * K;
            if (base + f < lim)
                atomicAdd(dw + base + f, (red[0][q][cl] + red[1][q][cl]) + (red[2][q][cl] + red[3][q][cl]));
        }
        if (wave == 0 && cok)
            atomicAdd(dbias + c, (red[0][K][lane] + red[1][K][lane]) + (red[2][K][lane] + red[3][K][lane]));
    }
}

// ---- train-mode BatchNorm statistics of c = dwconv(g) + bias: ONE pass over g, deterministic ----------------------------
// The conv output of a chunk (TT frames x 64 channels) exists only in registers; each lane (channel) folds it into a running
// (count, mean, M2 = sum of squared deviations) triple with Chan's pairwise update -- no E[x^2] - mean^2 cancellation -- the
// four waves are merged through LDS in a fixed order and every workgroup writes ONE partial triple per channel to the
// workspace.  A second kernel merges the partials of a channel in workgroup order (bit-reproducible: the round-1 version
// accumulated sum and centred second moment with fp32 atomics in two passes over g, and its run-to-run rounding noise grew
// to a full bf16 ulp in the logits of a 16-block model under autocast) and applies the running-statistics update.
__device__ __forceinline__ void chan_merge(float& n, float& mean, float& m2, float nb, float mb, float m2b) {
    if (nb > 0.f) {
        const float nn = n + nb, d = mb - mean, f = nb / nn;
        mean = fmaf(d, f, mean);
        m2 = m2 + m2b + d * d * n * f;
        n = nn;
    }
}

template <int K, int TT>
__global__ __launch_bounds__(256) void dwconv_stats_kernel(const float* __restrict__ g, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ part,
                                                           int T, int C, int seg_len) {
    constexpr int HALF = (K - 1) / 2;
    __shared__ float taps[64 * K];
    __shared__ float red[4][3][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int b = blockIdx.z;
    const int seg0 = blockIdx.y * seg_len, seg1 = min(T, seg0 + seg_len);
    const bool cok = c < C;
    const int cc = cok ? c : C - 1;
    float wr[K];
    load_taps<K>(w, blockIdx.x * 64, C, taps, wr);
    const float bi = bias[cc];
    const float* gb = g + (int64_t)b * T * C + cc;
    float n = 0.f, mean = 0.f, m2 = 0.f;
    for (int t0 = seg0 + wave * TT; t0 < seg1; t0 += 4 * TT) {
        float acc[TT];
#pragma unroll
        for (int o = 0; o < TT; ++o) acc[o] = bi;
        float win[TT + K - 1];
        load_window<TT + K - 1>(gb, t0 - HALF, T, C, win);
#pragma unroll
        for (int o = 0; o < TT; ++o)
#pragma unroll
            for (int j = 0; j < K; ++j) acc[o] = fmaf(wr[j], win[o + j], acc[o]);
        const int m = min(TT, seg1 - t0);                                 // valid frames of this chunk (wave-uniform)
        float cs = 0.f;
#pragma unroll
        for (int o = 0; o < TT; ++o) cs += o < m ? acc[o] : 0.f;
        const float cm = cs / (float)m;
        float cm2 = 0.f;
#pragma unroll
        for (int o = 0; o < TT; ++o) cm2 += o < m ? (acc[o] - cm) * (acc[o] - cm) : 0.f;
        chan_merge(n, mean, m2, (float)m, cm, cm2);
    }
    red[wave][0][lane] = n; red[wave][1][lane] = mean; red[wave][2][lane] = m2;
    __syncthreads();
    if (wave == 0 && cok) {
        float N = red[0][0][lane], M = red[0][1][lane], Q = red[0][2][lane];
#pragma unroll
        for (int v = 1; v < 4; ++v) chan_merge(N, M, Q, red[v][0][lane], red[v][1][lane], red[v][2][lane]);
        float* p = part + (int64_t)(blockIdx.z * gridDim.y + blockIdx.y) * 3 * C;
        p[c] = N; p[C + c] = M; p[2 * C + c] = Q;
    }
}

// mean / biased variance of a channel = the ordered merge of its workgroup partials (workgroup = 64 channels x 16 waves; wave v
// folds the v-th sixteenth of the partials in index order, wave 0 then folds the sixteen results in wave order: a fixed tree);
// running <- (1-mom)*running + mom*{mean, var*n/(n-1)}   (convolution.py:16 defaults).
// The partial triples of a wave are requested in batches of 8 before they are merged: the first version (4 waves, one
// dependent load + merge per partial) took as long as the statistics pass itself (20 us at 128 partials).
__global__ __launch_bounds__(1024) void bn_merge_update_kernel(const float* __restrict__ part, int nblk,
                                                               float* __restrict__ batch_mean, float* __restrict__ batch_var,
                                                               float* __restrict__ run_mean, float* __restrict__ run_var,
                                                               float momentum, int C) {
    constexpr int NW = 16, NBAT = 8;
    __shared__ float red[NW][3][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int cc = c < C ? c : C - 1;
    const int per = (nblk + NW - 1) / NW, i0 = wave * per, i1 = min(nblk, i0 + per);
    float n = 0.f, mean = 0.f, m2 = 0.f;
    for (int ib = i0; ib < i1; ib += NBAT) {
        float pn[NBAT], pm[NBAT], pq[NBAT];
#pragma unroll
        for (int j = 0; j < NBAT; ++j) {
            const float* p = part + (int64_t)min(ib + j, nblk - 1) * 3 * C;
            pn[j] = p[cc]; pm[j] = p[C + cc]; pq[j] = p[2 * C + cc];
        }
#pragma unroll
        for (int j = 0; j < NBAT; ++j)
            if (ib + j < i1) chan_merge(n, mean, m2, pn[j], pm[j], pq[j]);           // (wave-uniform)
    }
    red[wave][0][lane] = n; red[wave][1][lane] = mean; red[wave][2][lane] = m2;
    __syncthreads();
    if (wave != 0 || c >= C) return;
#pragma unroll
    for (int v = 1; v < NW; ++v) chan_merge(n, mean, m2, red[v][0][lane], red[v][1][lane], red[v][2][lane]);
    const float var = m2 / n;
    batch_mean[c] = mean;
    batch_var[c] = var;
    if (run_mean) run_mean[c] = (1.0f - momentum) * run_mean[c] + momentum * mean;
    if (run_var) run_var[c] = (1.0f - momentum) * run_var[c] + momentum * (n > 1.f ? m2 / (n - 1.f) : var);
}

// plain depthwise correlation y[t] = sum_j w[FLIP ? K-1-j : j] x[t+j-H] (no bias): the input-gradient pass
template <int K, int TT, bool FLIP>
__global__ __launch_bounds__(256) void dwconv_plain_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           float* __restrict__ y, int T, int C) {
    constexpr int HALF = (K - 1) / 2;
    __shared__ float taps[64 * K];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int t0 = (blockIdx.y * 4 + wave) * TT;
    const int b = blockIdx.z;
    float wr[K];
    load_taps<K, FLIP>(w, blockIdx.x * 64, C, taps, wr);
    if (t0 >= T) return;
    const bool cok = c < C;
    const int cc = cok ? c : C - 1;
    float acc[TT];
#pragma unroll
    for (int o = 0; o < TT; ++o) acc[o] = 0.f;
    const float* xb = x + (int64_t)b * T * C + cc;
    float win[TT + K - 1];
    load_window<TT + K - 1>(xb, t0 - HALF, T, C, win);
#pragma unroll
    for (int o = 0; o < TT; ++o)
#pragma unroll
        for (int j = 0; j < K; ++j) acc[o] = fmaf(wr[j], win[o + j], acc[o]);
    float* yb = y + (int64_t)b * T * C + cc;
#pragma unroll
    for (int o = 0; o < TT; ++o) {
        const int t = t0 + o;
        if (t < T && cok) yb[(int64_t)t * C] = acc[o];
    }
}

}  // namespace

// dx = LayerNorm-backward(dy) [+ dres].  mean/rstd are the forward's saved per-row statistics.
extern "C" int cfm_layernorm_bwd_dx_f32(const float* x, const float* gamma, const float* dy, const float* mean,
                                        const float* rstd, const float* dres_or_null, float* dx, int64_t rows, int d,
                                        cfm_stream_t stream) {
    CFM_REQUIRE(x && gamma && dy && mean && rstd && dx, CFM_ERR_NULL);
    CFM_REQUIRE(rows > 0 && d > 0 && (d & 3) == 0, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(d <= 8192, CFM_ERR_UNSUPPORTED);
    CFM_REQUIRE(CFM_ALIGNED16(x) && CFM_ALIGNED16(dy) && CFM_ALIGNED16(dx) && CFM_ALIGNED16(gamma), CFM_ERR_ALIGN);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
#define LNB(V) hipLaunchKernelGGL(layernorm_bwd_dx_kernel<V>, grid, block, 0, s, x, gamma, dy, mean, rstd, dres_or_null, dx, rows, d)
    if (d <= 256) LNB(1);
    else if (d <= 512) LNB(2);
    else if (d <= 1024) LNB(4);
    else if (d <= 2048) LNB(8);
    else LNB(32);
#undef LNB
    return cfm_launch_status();
}

// Both of the above in one pass over x and dy (d <= 2048; wider rows: call the two functions).  dgamma / dbeta are
// accumulated.  The per-workgroup partial sums go through `workspace` (cfm_layernorm_bwd_workspace_bytes(rows, d) bytes,
// 16-byte aligned) and are combined in a fixed order: the parameter gradients are reproducible bit for bit.
static int ln_bwd_rows_per_wave(int64_t rows) {
    // rows per wave: as many as keep >= ~512 workgroups in the grid (two per CU), at most 8
    return (int)std::max<int64_t>(1, std::min<int64_t>(8, rows / 2048));
}
extern "C" size_t cfm_layernorm_bwd_workspace_bytes(int64_t rows, int d) {
    if (rows <= 0 || d <= 0) return 0;
    const int rpw = ln_bwd_rows_per_wave(rows);
    return (size_t)((rows + 4 * rpw - 1) / (4 * rpw)) * 2 * (size_t)d * sizeof(float);
}
extern "C" int cfm_layernorm_bwd_f32(const float* x, const float* gamma, const float* dy, const float* mean, const float* rstd,
                                     const float* dres_or_null, float* dx, float* dgamma, float* dbeta, int64_t rows, int d,
                                     void* workspace, size_t workspace_bytes, cfm_stream_t stream) {
    CFM_REQUIRE(x && gamma && dy && mean && rstd && dx && dgamma && dbeta && workspace, CFM_ERR_NULL);
    CFM_REQUIRE(workspace_bytes >= cfm_layernorm_bwd_workspace_bytes(rows, d) && CFM_ALIGNED16(workspace), CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(rows > 0 && d > 0 && (d & 3) == 0, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(d <= 2048, CFM_ERR_UNSUPPORTED);
    CFM_REQUIRE(CFM_ALIGNED16(x) && CFM_ALIGNED16(dy) && CFM_ALIGNED16(dx) && CFM_ALIGNED16(gamma) &&
                (!dres_or_null || CFM_ALIGNED16(dres_or_null)), CFM_ERR_ALIGN);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int rpw = ln_bwd_rows_per_wave(rows);
    const dim3 grid((unsigned)((rows + 4 * rpw - 1) / (4 * rpw))), block(256);
    float* partials = static_cast<float*>(workspace);
#define LNF(V) hipLaunchKernelGGL(layernorm_bwd_fused_kernel<V>, grid, block, 0, s, x, gamma, dy, mean, rstd, dres_or_null, dx, \
                                  partials, rows, d, rpw)
    if (d <= 256) LNF(1);
    else if (d <= 512) LNF(2);
    else if (d <= 1024) LNF(4);
    else LNF(8);
#undef LNF
    hipLaunchKernelGGL(layernorm_bwd_partials_kernel, dim3((unsigned)((2 * d + 63) / 64)), block, 0, s, partials, (int)grid.x, d,
                       dgamma, dbeta);
    return cfm_launch_status();
}

// dgamma[c] += sum_r dy*xhat, dbeta[c] += sum_r dy   (caller zero-fills both)
extern "C" int cfm_layernorm_bwd_params_f32(const float* x, const float* dy, const float* mean, const float* rstd,
                                            float* dgamma, float* dbeta, int64_t rows, int d, cfm_stream_t stream) {
    CFM_REQUIRE(x && dy && mean && rstd && dgamma && dbeta, CFM_ERR_NULL);
    CFM_REQUIRE(rows > 0 && d > 0, CFM_ERR_BAD_SHAPE);
    colreduce_launch<1>(x, dy, mean, rstd, (int64_t)d, rows, d, 1.0f, dgamma, dbeta, static_cast<hipStream_t>(stream));
    return cfm_launch_status();
}

// out[c] += alpha * sum_r X[r][c]   (bias gradients; caller zero-fills out)
extern "C" int cfm_colsum_f32(const float* X, int64_t ld, int64_t rows, int cols, float alpha, float* out,
                              cfm_stream_t stream) {
    CFM_REQUIRE(X && out, CFM_ERR_NULL);
    CFM_REQUIRE(rows > 0 && cols > 0 && ld >= cols, CFM_ERR_BAD_SHAPE);
    colreduce_launch<0>(X, nullptr, nullptr, nullptr, ld, rows, cols, alpha, out, nullptr, static_cast<hipStream_t>(stream));
    return cfm_launch_status();
}

extern "C" int cfm_glu_fwd_f32(const float* z, float* y, int64_t rows, int n, cfm_stream_t stream) {
    CFM_REQUIRE(z && y, CFM_ERR_NULL);
    CFM_REQUIRE(rows > 0 && n > 0 && (n & 3) == 0, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(CFM_ALIGNED16(z) && CFM_ALIGNED16(y), CFM_ERR_ALIGN);
    const int64_t total = rows * (n / 4);
    hipLaunchKernelGGL(glu_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), z, y, rows, n);
    return cfm_launch_status();
}

extern "C" int cfm_glu_bwd_f32(const float* z, const float* dy, float* dz, int64_t rows, int n, cfm_stream_t stream) {
    CFM_REQUIRE(z && dy && dz, CFM_ERR_NULL);
    CFM_REQUIRE(rows > 0 && n > 0 && (n & 3) == 0, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(CFM_ALIGNED16(z) && CFM_ALIGNED16(dy) && CFM_ALIGNED16(dz), CFM_ERR_ALIGN);
    const int64_t total = rows * (n / 4);
    hipLaunchKernelGGL(glu_bwd_kernel<float>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), z, dy, dz, rows, n);
    return cfm_launch_status();
}

// cfm_glu_bwd_f32 with dz stored in the 16-bit type `prec` (a gradient that only feeds 16-bit GEMM operands)
extern "C" int cfm_glu_bwd_out16_f32(int prec, const float* z, const float* dy, void* dz16, int64_t rows, int n, cfm_stream_t stream) {
    CFM_REQUIRE(z && dy && dz16, CFM_ERR_NULL);
    CFM_REQUIRE(rows > 0 && n > 0 && (n & 3) == 0, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(CFM_ALIGNED16(z) && CFM_ALIGNED16(dy) && (reinterpret_cast<uintptr_t>(dz16) & 7) == 0, CFM_ERR_ALIGN);
    const int64_t total = rows * (n / 4);
    const dim3 grid((unsigned)((total + 255) / 256));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (prec == CFM_PREC_BF16) hipLaunchKernelGGL(glu_bwd_kernel<__bf16>, grid, dim3(256), 0, s, z, dy, static_cast<__bf16*>(dz16), rows, n);
    else if (prec == CFM_PREC_FP16) hipLaunchKernelGGL(glu_bwd_kernel<_Float16>, grid, dim3(256), 0, s, z, dy, static_cast<_Float16*>(dz16), rows, n);
    else return CFM_ERR_UNSUPPORTED;
    return cfm_launch_status();
}

// Backward of cfm_dwconv_bn_swish_fwd_f32.  train_stats = 0: bn_mean/bn_var are constants (eval);
// train_stats = 1: they are the BATCH statistics (biased variance) and the mean/variance coupling of BatchNorm's
// backward is applied (n = B*T).  Outputs: dg (B,T,C); dc_ws (B,T,C) workspace; dw (C,K), dbias, dgamma, dbeta (C)
// accumulated (caller zero-fills).
extern "C" int cfm_dwconv_bn_swish_bwd_f32(const float* g, const float* dy, const float* w, const float* bias,
                                           const float* bn_weight, const float* bn_bias, const float* bn_mean,
                                           const float* bn_var, float bn_eps, int train_stats, float* dc_ws, float* dg,
                                           float* dw, float* dbias, float* dgamma, float* dbeta, int B, int T, int C,
                                           int K, cfm_stream_t stream) {
    CFM_REQUIRE(g && dy && w && bias && bn_weight && bn_bias && bn_mean && bn_var && dc_ws && dg && dw && dbias &&
                dgamma && dbeta, CFM_ERR_NULL);
    CFM_REQUIRE(B > 0 && T > 0 && C > 0 && K > 0 && (K & 1) == 1, CFM_ERR_BAD_SHAPE);
    hipStream_t s = static_cast<hipStream_t>(stream);
    constexpr int TT = 8;
    const float inv_n = train_stats ? 1.0f / ((float)B * (float)T) : 0.f;
    // enough workgroups to fill the chip (>= ~4 per CU) but as few partial-sum atomics as possible
    const int cblocks = (C + 63) / 64;
    int nseg = (1024 + cblocks * B - 1) / (cblocks * B);
    nseg = nseg < 1 ? 1 : nseg;
    int seg_len = (T + nseg - 1) / nseg;
    seg_len = (seg_len + 4 * TT - 1) / (4 * TT) * (4 * TT);
    nseg = (T + seg_len - 1) / seg_len;
    const dim3 rgrid((unsigned)cblocks, (unsigned)nseg, (unsigned)B), block(256);
    const dim3 grid((unsigned)cblocks, (unsigned)((T + 4 * TT - 1) / (4 * TT)), (unsigned)B);
#define DWB(KK)                                                                                                       \
    hipLaunchKernelGGL((dwconv_bn_swish_bwd_kernel<KK, TT, 0>), rgrid, block, 0, s, g, dy, w, bias, bn_weight, bn_bias, \
                       bn_mean, bn_var, bn_eps, dc_ws, dw, dbias, dgamma, dbeta, T, C, seg_len, inv_n);                \
    hipLaunchKernelGGL((dwconv_bn_swish_bwd_kernel<KK, TT, 1>), rgrid, block, 0, s, g, dy, w, bias, bn_weight, bn_bias, \
                       bn_mean, bn_var, bn_eps, dc_ws, dw, dbias, dgamma, dbeta, T, C, seg_len, inv_n);                \
    hipLaunchKernelGGL((dwconv_plain_kernel<KK, TT, true>), grid, block, 0, s, dc_ws, w, dg, T, C)
    switch (K) {
        case 31: DWB(31); break;
        case 15: DWB(15); break;
        case 7: DWB(7); break;
        case 3: DWB(3); break;
        default: return CFM_ERR_UNSUPPORTED;
    }
#undef DWB
    return cfm_launch_status();
}

// Train-mode BatchNorm statistics of the depthwise-conv output (convolution.py:26-27 in .train()): batch_mean,
// batch_var (biased) over all B*T positions (padded frames included); running_mean/var (may be NULL)
// updated in place with `momentum` and the unbiased variance.  The conv output is recomputed, never stored; one pass
// over g; bit-reproducible (no atomics).  workspace: cfm_dwconv_bn_stats_workspace_bytes(B, T, C) bytes.
static void dwconv_stats_geometry(int B, int T, int C, int* cblocks, int* nseg, int* seg_len) {
    constexpr int TT = 16;
    *cblocks = (C + 63) / 64;
    int ns = (1024 + *cblocks * B - 1) / (*cblocks * B);                 // >= ~4 workgroups per CU
    ns = ns < 1 ? 1 : ns;
    int sl = (T + ns - 1) / ns;
    sl = (sl + 4 * TT - 1) / (4 * TT) * (4 * TT);
    *seg_len = sl;
    *nseg = (T + sl - 1) / sl;
}

extern "C" size_t cfm_dwconv_bn_stats_workspace_bytes(int B, int T, int C) {
    if (B <= 0 || T <= 0 || C <= 0) return 0;
    int cblocks, nseg, seg_len;
    dwconv_stats_geometry(B, T, C, &cblocks, &nseg, &seg_len);
    return (size_t)B * nseg * 3 * C * sizeof(float);
}

extern "C" int cfm_dwconv_bn_stats_f32(const float* g, const float* w, const float* bias, float* batch_mean,
                                       float* batch_var, float* running_mean_or_null, float* running_var_or_null,
                                       float momentum, int B, int T, int C, int K, void* workspace,
                                       size_t workspace_bytes, cfm_stream_t stream) {
    CFM_REQUIRE(g && w && bias && batch_mean && batch_var && workspace, CFM_ERR_NULL);
    CFM_REQUIRE(B > 0 && T > 0 && C > 0 && K > 0 && (K & 1) == 1, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(workspace_bytes >= cfm_dwconv_bn_stats_workspace_bytes(B, T, C), CFM_ERR_BAD_SHAPE);
    hipStream_t s = static_cast<hipStream_t>(stream);
    constexpr int TT = 16;
    int cblocks, nseg, seg_len;
    dwconv_stats_geometry(B, T, C, &cblocks, &nseg, &seg_len);
    const dim3 grid((unsigned)cblocks, (unsigned)nseg, (unsigned)B), block(256);
    float* part = static_cast<float*>(workspace);
#define DWS(KK) hipLaunchKernelGGL((dwconv_stats_kernel<KK, TT>), grid, block, 0, s, g, w, bias, part, T, C, seg_len)
    switch (K) {
        case 31: DWS(31); break;
        case 15: DWS(15); break;
        case 7: DWS(7); break;
        case 3: DWS(3); break;
        default: return CFM_ERR_UNSUPPORTED;
    }
#undef DWS
    hipLaunchKernelGGL(bn_merge_update_kernel, dim3((unsigned)cblocks), dim3(1024), 0, s, part, B * nseg, batch_mean,
                       batch_var, running_mean_or_null, running_var_or_null, momentum, C);
    return cfm_launch_status();
}
