// Row N1, loss half: ConformerCriterion.ctc_loss (evaluation.py:12-16) =
//     nn.CTCLoss(blank, reduction='mean', zero_infinity=True)(logits.float().log_softmax(-1).transpose(0,1), ...)
// with the log-softmax folded in: nothing of shape (T,B,V) other than the logits gradient is ever written.
//
//   prepare : one wave per frame  -> lse (B,T), log-prob of blank lpb (B,T), log-prob of every target label lpl (B,T,Lmax)
//   alpha   : one wave per utterance, the whole 2L+1 state lattice in registers: lane owns P consecutive
//             (blank, label) state pairs, so a time step needs ONE cross-lane value (the label state left of the lane's
//             first pair); log-probs are prefetched a block of steps ahead, the recursion itself touches no memory but
//             its own alpha row store.  nll (B), and the mean loss in a second one-wave kernel.  The lattice is kept
//             in fp32 but RE-CENTRED every prefetch block (its maximum is moved into a float64 per-frame offset), so
//             the rounding error depends on the drift inside one block (~1e-5), not on |log-likelihood| of a long
//             utterance (fp32 lattices lose the gradient beyond a few thousand frames).
//   beta    : mirror image (two cross-lane values), run in the backward.
//   grad    : one wave per frame: d loss / d logits = scale_b * (softmax - state occupancy folded onto the vocabulary),
//             occupancy_s = exp(alpha_s + beta_s + nll - lp_s); repeated labels are summed by their first occurrence
//             (deterministic, no atomics).
// HBM-bound on the logits (B*T*V read twice, written once); the chains are latency-bound (T sequential steps).
#include "cfm_common.h"

namespace {

struct CtcArgs {
    const float* logits; const int64_t* targets; const int64_t* tgt_off; const int64_t* in_len; const int64_t* tgt_len;
    float* lse; float* lpb; float* lpl; float* alpha; float* beta; float* nll; float* loss;
    double* ca; double* cb; double* ll;          // per-frame lattice offsets (alpha, beta) and log-likelihood per utterance
    const float* grad_out; float* dlogits;
    int64_t tgt_stride, tgt_numel;
    int B, T, V, Lmax, blank, P;
};

constexpr float NEG_INF = -__builtin_inff();

__device__ __forceinline__ float log_fast(float x) { return __builtin_amdgcn_logf(x) * 0.69314718055994530942f; }
__device__ __forceinline__ float lse2(float a, float b) {
    const float m = fmaxf(a, b);
    return m == NEG_INF ? NEG_INF : m + log_fast(exp_fast(a - m) + exp_fast(b - m));
}
__device__ __forceinline__ float lse3(float a, float b, float c) {
    const float m = fmaxf(fmaxf(a, b), c);
    return m == NEG_INF ? NEG_INF : m + log_fast(exp_fast(a - m) + exp_fast(b - m) + exp_fast(c - m));
}

// utterance geometry, clamped so that no index derived from it can leave a buffer
__device__ __forceinline__ void ctc_geometry(const CtcArgs& a, int b, int& Tb, int& Lb, int64_t& off) {
    off = a.tgt_off ? a.tgt_off[b] : (int64_t)b * a.tgt_stride;
    off = off < 0 ? 0 : (off > a.tgt_numel ? a.tgt_numel : off);
    int64_t L = a.tgt_len[b];
    L = L < 0 ? 0 : L;
    L = L > a.Lmax ? a.Lmax : L;
    L = L > a.tgt_numel - off ? a.tgt_numel - off : L;
    Lb = (int)L;
    int64_t t = a.in_len[b];
    Tb = (int)(t < 0 ? 0 : (t > a.T ? a.T : t));
}
__device__ __forceinline__ int ctc_label(const CtcArgs& a, int64_t off, int i) {
    const int64_t v = a.targets[off + i];
    return (int)(v < 0 ? 0 : (v >= a.V ? a.V - 1 : v));
}

__global__ __launch_bounds__(256) void ctc_prepare_kernel(const CtcArgs a) {
    const int64_t frame = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (frame >= (int64_t)a.B * a.T) return;
    const int b = (int)(frame / a.T), t = (int)(frame - (int64_t)b * a.T);
    int Tb, Lb; int64_t off;
    ctc_geometry(a, b, Tb, Lb, off);
    if (t >= Tb) return;
    const float* row = a.logits + frame * a.V;
    float m = NEG_INF;
    for (int v = lane; v < a.V; v += 64) m = fmaxf(m, row[v]);
    m = wave_max(m);
    float s = 0.f;
    for (int v = lane; v < a.V; v += 64) s += exp_fast(row[v] - m);
    s = wave_sum(s);
    const float lse = m + logf(s);
    if (lane == 0) { a.lse[frame] = lse; a.lpb[frame] = row[a.blank] - lse; }
    float* out = a.lpl + frame * a.Lmax;
    for (int i = lane; i < Lb; i += 64) out[i] = row[ctc_label(a, off, i)] - lse;
}

// One wave per utterance.  Lane l owns the pairs i = l*P .. l*P+P-1: blank state 2i (exists for i <= L) and label state
// 2i+1 (exists for i < L).  Rows of alpha / beta: [t][2*64*P], element 2i (+1).
template <int P, bool BETA>
__global__ __launch_bounds__(64) void ctc_chain_kernel(const CtcArgs a) {
    constexpr int U = P >= 16 ? 1 : 16 / P;                   // time steps per prefetch block
    constexpr int NP = 64 * P;
    const int b = blockIdx.x, lane = threadIdx.x;
    int Tb, Lb; int64_t off;
    ctc_geometry(a, b, Tb, Lb, off);
    if (Tb == 0) {
        if (!BETA && lane == 0) { a.nll[b] = Lb == 0 ? 0.f : -NEG_INF; a.ll[b] = Lb == 0 ? 0.0 : (double)NEG_INF; }
        return;
    }
    double* coff = (BETA ? a.cb : a.ca) + (int64_t)b * a.T;
    double C = 0.0;                                           // stored value + C = true log alpha (beta)
    float* out = (BETA ? a.beta : a.alpha) + (int64_t)b * a.T * 2 * NP;
    const float* lplb = a.lpl + (int64_t)b * a.T * a.Lmax;
    const float* lpbb = a.lpb + (int64_t)b * a.T;

    int lab[P];
    bool hop[P];                  // alpha: state 2i+1 may be entered from 2i-1; beta: 2i+1 may go to 2i+3
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const int i = lane * P + p;
        lab[p] = i < Lb ? ctc_label(a, off, i) : -1 - i;     // distinct negatives: never equal to a neighbour
    }
    const int nb_lab = BETA ? __shfl_down(lab[0], 1, 64) : __shfl_up(lab[P - 1], 1, 64);
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const int i = lane * P + p;
        if (BETA) {
            const int nxt = p == P - 1 ? nb_lab : lab[p + 1];
            hop[p] = (i + 1 < Lb) && lab[p] != nxt;
        } else {
            const int prv = p == 0 ? nb_lab : lab[p - 1];
            hop[p] = (i >= 1) && (i < Lb) && lab[p] != prv;
        }
    }
    int idx[P];
#pragma unroll
    for (int p = 0; p < P; ++p) idx[p] = min(lane * P + p, a.Lmax - 1);

    float sb[P], sl[P];          // current alpha (beta) of the lane's blank / label states
    float cb[U], cl[U][P], nb[U], nl[U][P];
    auto load_block = [&](int t_first, float (&vb)[U], float (&vl)[U][P]) {      // clamped: always in range
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = min(max(BETA ? t_first - u : t_first + u, 0), Tb - 1);
            vb[u] = lpbb[t];
#pragma unroll
            for (int p = 0; p < P; ++p) vl[u][p] = lplb[(int64_t)t * a.Lmax + idx[p]];
        }
    };
    const int t_begin = BETA ? Tb - 1 : 0;
    load_block(t_begin, cb, cl);
    for (int done = 0; done < Tb; done += U) {
        const int t0 = BETA ? t_begin - done : done;
        load_block(BETA ? t0 - U : t0 + U, nb, nl);
        __builtin_amdgcn_sched_barrier(0);
        if (done > 0) {                                                      // re-centre the lattice on its maximum
            float m = NEG_INF;
#pragma unroll
            for (int p = 0; p < P; ++p) m = fmaxf(m, fmaxf(sb[p], sl[p]));
            m = wave_max(m);
            if (m > NEG_INF) {
#pragma unroll
                for (int p = 0; p < P; ++p) { sb[p] -= m; sl[p] -= m; }
                C += (double)m;
            }
        }
        if (lane < U && done + lane < Tb) coff[BETA ? t0 - lane : t0 + lane] = C;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (done + u >= Tb) break;                                       // uniform
            const int t = BETA ? t0 - u : t0 + u;
            if (done + u == 0) {
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    const int i = lane * P + p;
                    if (BETA) {
                        sb[p] = i == Lb ? cb[u] : NEG_INF;                       // state S-1
                        sl[p] = i == Lb - 1 ? cl[u][p] : NEG_INF;                // state S-2
                    } else {
                        sb[p] = i == 0 ? cb[u] : NEG_INF;                        // state 0
                        sl[p] = (i == 0 && Lb > 0) ? cl[u][p] : NEG_INF;         // state 1
                    }
                }
            } else if (BETA) {
                float right_b = __shfl_down(sb[0], 1, 64), right_l = __shfl_down(sl[0], 1, 64);
                if (lane == 63) { right_b = NEG_INF; right_l = NEG_INF; }
                float nsb[P], nsl[P];
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    const int i = lane * P + p;
                    const float rb = p == P - 1 ? right_b : sb[p + 1], rl = p == P - 1 ? right_l : sl[p + 1];
                    const float vb = lse2(sb[p], sl[p]) + cb[u];
                    const float vl = lse3(sl[p], rb, hop[p] ? rl : NEG_INF) + cl[u][p];
                    nsb[p] = i <= Lb ? vb : NEG_INF;
                    nsl[p] = i < Lb ? vl : NEG_INF;
                }
#pragma unroll
                for (int p = 0; p < P; ++p) { sb[p] = nsb[p]; sl[p] = nsl[p]; }
            } else {
                float left = __shfl_up(sl[P - 1], 1, 64);
                if (lane == 0) left = NEG_INF;
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    const int i = lane * P + p;
                    const float ob = sb[p], ol = sl[p];
                    const float vb = lse2(ob, left) + cb[u];
                    const float vl = lse3(ol, ob, hop[p] ? left : NEG_INF) + cl[u][p];
                    sb[p] = i <= Lb ? vb : NEG_INF;
                    sl[p] = i < Lb ? vl : NEG_INF;
                    left = ol;
                }
            }
            float* orow = out + (int64_t)t * 2 * NP + (int64_t)lane * 2 * P;
#pragma unroll
            for (int p = 0; p < P; ++p) { orow[2 * p] = sb[p]; orow[2 * p + 1] = sl[p]; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            cb[u] = nb[u];
#pragma unroll
            for (int p = 0; p < P; ++p) cl[u][p] = nl[u][p];
        }
    }
    if (!BETA) {
        float fb = NEG_INF, fl = NEG_INF;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int i = lane * P + p;
            if (i == Lb) fb = sb[p];
            if (i == Lb - 1) fl = sl[p];
        }
        fb = wave_max(fb); fl = wave_max(fl);
        if (lane == 0) {
            const double ll = C + (double)lse2(fb, fl);
            a.ll[b] = ll;
            a.nll[b] = (float)(-ll);
        }
    }
}

// mean over the batch of nll_b / max(L_b, 1), infinite terms dropped (zero_infinity=True)
__global__ __launch_bounds__(64) void ctc_mean_kernel(const CtcArgs a) {
    float s = 0.f;
    for (int b = threadIdx.x; b < a.B; b += 64) {
        int Tb, Lb; int64_t off;
        ctc_geometry(a, b, Tb, Lb, off);
        const double ll = a.ll[b];
        s += (ll > (double)NEG_INF && ll == ll) ? (float)(-ll / (double)max(Lb, 1)) : 0.f;
    }
    s = wave_sum(s);
    if (threadIdx.x == 0) a.loss[0] = s / (float)a.B;
}

// one wave per frame; dynamic LDS per wave: V floats (row) + 64P floats (label occupancy) + 64P ints (labels)
__global__ __launch_bounds__(256) void ctc_grad_kernel(const CtcArgs a, int waves) {
    extern __shared__ float lds[];
    const int NP = 64 * a.P;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* g = lds + (size_t)w * (a.V + 2 * NP);
    float* occ = g + a.V;
    int* labs = reinterpret_cast<int*>(occ + NP);
    const int64_t frame = (int64_t)blockIdx.x * waves + w;
    const bool in_grid = frame < (int64_t)a.B * a.T;
    const int b = in_grid ? (int)(frame / a.T) : 0, t = in_grid ? (int)(frame - (int64_t)b * a.T) : 0;
    int Tb, Lb; int64_t off;
    ctc_geometry(a, b, Tb, Lb, off);
    const double ll = a.ll[b];
    const bool live = in_grid && t < Tb && ll > (double)NEG_INF && ll == ll;
    float blank_sum = 0.f;
    if (live) {
        const float* al = a.alpha + (frame * 2) * NP;
        const float* be = a.beta + (frame * 2) * NP;
        const float* lpl = a.lpl + frame * a.Lmax;
        const float lpb = a.lpb[frame], lse = a.lse[frame];
        const float shift = (float)(a.ca[frame] + a.cb[frame] - ll);          // offsets of the two lattices - log-likelihood
        for (int i = lane; i <= Lb; i += 64) {
            blank_sum += exp_fast((al[2 * i] + be[2 * i]) + (shift - lpb));
            if (i < Lb) {
                occ[i] = exp_fast((al[2 * i + 1] + be[2 * i + 1]) + (shift - lpl[i]));
                labs[i] = ctc_label(a, off, i);
            }
        }
        blank_sum = wave_sum(blank_sum);
        const float* row = a.logits + frame * a.V;
        for (int v = lane; v < a.V; v += 64) g[v] = exp_fast(row[v] - lse);
    }
    __syncthreads();
    if (live) {
        for (int i = lane; i < Lb; i += 64) {
            const int li = labs[i];
            bool first = true;
            for (int j = 0; j < i; ++j) first = first && labs[j] != li;
            if (!first) continue;
            float s = 0.f;
            for (int j = i; j < Lb; ++j) s += labs[j] == li ? occ[j] : 0.f;
            g[li] -= s;
        }
    }
    __syncthreads();
    if (live && lane == 0) g[a.blank] -= blank_sum;
    __syncthreads();
    if (!in_grid) return;
    float* drow = a.dlogits + frame * a.V;
    if (live) {
        const float scale = a.grad_out[0] / ((float)a.B * (float)max(Lb, 1));
        for (int v = lane; v < a.V; v += 64) drow[v] = scale * g[v];
    } else {
        for (int v = lane; v < a.V; v += 64) drow[v] = 0.f;
    }
}

int pairs_per_lane(int Lmax) {                       // 64*P >= Lmax + 1
    int P = 1;
    while (64 * P < Lmax + 1) P *= 2;
    return P;
}

template <bool BETA>
void launch_chain(const CtcArgs& a, hipStream_t s) {
    const dim3 grid((unsigned)a.B), block(64);
    switch (a.P) {
        case 1: hipLaunchKernelGGL((ctc_chain_kernel<1, BETA>), grid, block, 0, s, a); break;
        case 2: hipLaunchKernelGGL((ctc_chain_kernel<2, BETA>), grid, block, 0, s, a); break;
        case 4: hipLaunchKernelGGL((ctc_chain_kernel<4, BETA>), grid, block, 0, s, a); break;
        case 8: hipLaunchKernelGGL((ctc_chain_kernel<8, BETA>), grid, block, 0, s, a); break;
        default: hipLaunchKernelGGL((ctc_chain_kernel<16, BETA>), grid, block, 0, s, a); break;
    }
}

int fill_args(CtcArgs& a, const float* logits, const int64_t* targets, const int64_t* tgt_off_or_null, int64_t tgt_stride,
              int64_t tgt_numel, const int64_t* in_len, const int64_t* tgt_len, int B, int T, int V, int Lmax, int blank,
              float* workspace) {
    CFM_REQUIRE(logits && targets && in_len && tgt_len && workspace, CFM_ERR_NULL);
    CFM_REQUIRE(B > 0 && T > 0 && V > 0 && Lmax >= 1 && blank >= 0 && blank < V && tgt_numel >= 0, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(Lmax <= CFM_CTC_MAX_TARGET, CFM_ERR_UNSUPPORTED);
    a = CtcArgs{};
    a.logits = logits; a.targets = targets; a.tgt_off = tgt_off_or_null; a.in_len = in_len; a.tgt_len = tgt_len;
    a.tgt_stride = tgt_stride; a.tgt_numel = tgt_numel;
    a.B = B; a.T = T; a.V = V; a.Lmax = Lmax; a.blank = blank; a.P = pairs_per_lane(Lmax);
    CFM_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 7) == 0, CFM_ERR_ALIGN);
    const int64_t frames = (int64_t)B * T, lattice = frames * 2 * 64 * a.P;
    a.ca = reinterpret_cast<double*>(workspace); a.cb = a.ca + frames; a.ll = a.cb + frames;      // float64 part first
    a.lse = reinterpret_cast<float*>(a.ll + B); a.lpb = a.lse + frames; a.lpl = a.lpb + frames;
    a.alpha = a.lpl + frames * Lmax; a.beta = a.alpha + lattice; a.nll = a.beta + lattice;
    return CFM_OK;
}

}  // namespace

extern "C" int64_t cfm_ctc_workspace_floats(int B, int T, int Lmax) {
    if (B <= 0 || T <= 0 || Lmax < 1 || Lmax > CFM_CTC_MAX_TARGET) return -1;
    const int64_t frames = (int64_t)B * T;
    return 2 * (2 * frames + B) + frames * (2 + Lmax) + 2 * frames * 2 * 64 * pairs_per_lane(Lmax) + B;
}

extern "C" int cfm_ctc_loss_fwd_f32(const float* logits, const int64_t* targets, const int64_t* tgt_off_or_null,
                                    int64_t tgt_stride, int64_t tgt_numel, const int64_t* in_len, const int64_t* tgt_len,
                                    int B, int T, int V, int Lmax, int blank, float* workspace, float* loss,
                                    cfm_stream_t stream) {
    CtcArgs a;
    const int rc = fill_args(a, logits, targets, tgt_off_or_null, tgt_stride, tgt_numel, in_len, tgt_len, B, T, V, Lmax, blank,
                             workspace);
    if (rc != CFM_OK) return rc;
    CFM_REQUIRE(loss, CFM_ERR_NULL);
    a.loss = loss;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t frames = (int64_t)B * T;
    hipLaunchKernelGGL(ctc_prepare_kernel, dim3((unsigned)((frames + 3) / 4)), dim3(256), 0, s, a);
    launch_chain<false>(a, s);
    hipLaunchKernelGGL(ctc_mean_kernel, dim3(1), dim3(64), 0, s, a);
    return cfm_launch_status();
}

extern "C" int cfm_ctc_loss_bwd_f32(const float* logits, const int64_t* targets, const int64_t* tgt_off_or_null,
                                    int64_t tgt_stride, int64_t tgt_numel, const int64_t* in_len, const int64_t* tgt_len,
                                    int B, int T, int V, int Lmax, int blank, float* workspace, const float* grad_out,
                                    float* dlogits, cfm_stream_t stream) {
    CtcArgs a;
    const int rc = fill_args(a, logits, targets, tgt_off_or_null, tgt_stride, tgt_numel, in_len, tgt_len, B, T, V, Lmax, blank,
                             workspace);
    if (rc != CFM_OK) return rc;
    CFM_REQUIRE(grad_out && dlogits, CFM_ERR_NULL);
    a.grad_out = grad_out; a.dlogits = dlogits;
    const size_t per_wave = ((size_t)V + 2 * 64 * (size_t)a.P) * sizeof(float);
    CFM_REQUIRE(per_wave <= 64 * 1024, CFM_ERR_UNSUPPORTED);
    const int waves = 4 * per_wave <= 64 * 1024 ? 4 : (2 * per_wave <= 64 * 1024 ? 2 : 1);
    hipStream_t s = static_cast<hipStream_t>(stream);
    launch_chain<true>(a, s);
    const int64_t frames = (int64_t)B * T;
    hipLaunchKernelGGL(ctc_grad_kernel, dim3((unsigned)((frames + waves - 1) / waves)), dim3(64 * waves), waves * per_wave, s,
                       a, waves);
    return cfm_launch_status();
}
