// Fused (flash-style) backward of the relative-position attention core, fp32 -- ONE kernel, no (B,H,T,T) or
// (H,B,T,2T-1) tensor anywhere (reference intermediates: model/utils/attention.py:49-70).
//
//   s[i,k] = ((q_i+u).k_k + (q_i+v).p_{i-k}) * scale ;  P = exp(s - lse_i) ;  W = P o M (dropout) ;  O = W.V
//   D_i = dO_i.O_i ; dW = dO.V^T ; dS = P o (dW o M - D_i) * scale
//   dV = W^T.dO ; dK = dS^T.(Q+u) ; d(Q+u) = dS.K ; d(Q+v)_i = sum_k dS[i,k] p_{i-k} ; dp_r = sum_{b, i-k=r} dS[i,k] (q_i+v)
//   dq = d(Q+u) + d(Q+v) ; du = sum_{b,i} d(Q+u)_i ; dv = sum_{b,i} d(Q+v)_i
//
// Decomposition: workgroup = 4 waves = 128 keys of one (batch, head); a wave owns 32 keys and keeps dK^T / dV^T of
// them in 64 accumulator registers while the workgroup sweeps the queries in tiles of 32 (so dK and dV need no sum
// across workgroups).  One workgroup per CU (120 KB of LDS, one wave per SIMD with the whole register file).
// Every product runs on v_mfma_f32_32x32x2_f32 with the KEY ON THE LANE (scores S[query][key]: query rows in the 16
// accumulator registers, key = lane), so the recomputed W and dS tiles are, as they stand in registers, the B operands
// of dV^T += dO^T.W and dK^T += (Q+u)^T.dS (contraction over the register axis: no lane movement).  dS crosses LDS
// once (a per-wave 32x33 tile) and is read back three ways: transposed for d(Q+u) = dS.K, and along the diagonals
// (the inverse of the "relative shift") for d(Q+v) = dG.Pband and dPband = dG^T.(Q+v).
// The positional band of a (32 query x 32 key) tile is 63 table rows: G = (Q+v).Pband^T is computed at 64 rows and
// skewed through the same per-wave LDS tile (row il, column (il - kl + 31) & 31: conflict-free both ways).
// Sums across workgroups go through fp32 global atomics: dq (the key blocks of one (b,h)) after the four waves'
// contributions have been summed through per-wave LDS slabs (plain stores; 33 MB of atomics per layer at cfg-2), dpos (all
// batches, all key blocks) straight from the accumulators once a band tile is complete for its wave (a register carry
// joins the two query tiles that touch the same 32 table rows; 134 MB per layer at cfg-2, issued behind the MFMAs).
// D_i = dO_i.O_i is computed while the query tile is staged (no separate row-dot kernel, no qbias kernel).
#include "cfm_common.h"
#include <math.h>

namespace {

constexpr int QROW = 68;     // padded LDS row (floats) of the staged query-tile operands and of the P ring: conflict-free ds_read_b128
constexpr int RING = 160;    // table rows resident per workgroup: 128 keys + 32 queries - 1, rounded to 5 x 32
constexpr int SROW = 33;     // padded row of the per-wave 32x32 skew / dS tile

struct AttnBwdArgs {
    const float* q; const float* k; const float* v; int64_t ld;
    const float* pos; int64_t ldp; const float* u; const float* vb; const int64_t* lengths;
    const float* o; const float* dout; int64_t ldo; const float* lse;
    float* dq; float* dk; float* dv; int64_t ldg;
    float* dpos; int64_t lddp; float* du; float* dvb;
    int B, T, H, dh; float scale; float drop_p; unsigned long long drop_seed;
    int prec;                             // CFM_PREC_*: under autocast the SCORE operands (q+u, q+v, K, table rows) and V are rounded to the
                                          // forward's 16-bit type, so the recomputed P matches the log-sum-exp the 16-bit forward kernel saved
    unsigned long long* trace;            // diagnostics: s_memrealtime stamps of wave 0 of workgroup (0,0), 16 per query tile
};

__device__ __forceinline__ int rho(int r, int hf) { return (r & 3) + 8 * (r >> 2) + 4 * hf; }   // accumulator register -> tile row
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ void glob_add(float* p, float v) { atomicAdd(p, v); }     // global_atomic_add_f32 (no return)
__device__ __forceinline__ float round16(float x, int prec) {                          // RNE to the autocast type and back
    return prec == CFM_PREC_BF16 ? (float)(__bf16)x : (prec == CFM_PREC_FP16 ? (float)(_Float16)x : x);
}
__device__ __forceinline__ f32x4 round16(f32x4 x, int prec) {
    if (prec == 0) return x;
    return f32x4{round16(x[0], prec), round16(x[1], prec), round16(x[2], prec), round16(x[3], prec)};
}

// Re-materialise the lane coordinates inside the query-tile loop: every LDS address below is a function of (li, hf) only,
// so loop-invariant code motion would otherwise hoist ~200 address / mask registers out of the loop and keep them live
// across all phases (measured: 256 + 256 registers and 800 bytes of scratch per lane without this).
#define ATB_FRESH_LANE() int li = li_; int hf = hf_; asm volatile("" : "+v"(li), "+v"(hf))

template <int NC, int ND>
__global__ __launch_bounds__(256, 1) void relpos_attn_bwd_kernel(const AttnBwdArgs a) {
    __shared__ __attribute__((aligned(16))) float smem[3 * 32 * QROW + 64 + RING * QROW + 4 * 32 * SROW + 4 * 32 * 64];
    float* Qu = smem;                       // [32][QROW]  q + u      (this query tile)
    float* Qv = Qu + 32 * QROW;             // [32][QROW]  q + v
    float* dOs = Qv + 32 * QROW;            // [32][QROW]  dO
    float* lseS = dOs + 32 * QROW;          // [32]
    float* DS = lseS + 32;                  // [32]        D_i = dO_i . O_i
    float* Pr = DS + 32;                    // [RING][QROW] projected-position rows (operand ring)
    float* gsAll = Pr + RING * QROW;        // [4][32][SROW] per-wave skew / dS tile
    float* dQs = gsAll + 4 * 32 * SROW;     // [4][32][64]   per-wave dq contribution of this query tile (summed when flushed)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, hf = lane >> 5;
    const int li_ = li, hf_ = hf;
    float* gs = gsAll + wave * 32 * SROW;
    const int bh = blockIdx.y, b = bh / a.H, h = bh % a.H;
    const int T = a.T, dh = a.dh;
    const int kb = blockIdx.x * 128, k0 = kb + 32 * wave;
    const bool wactive = k0 < T;                                   // wave-uniform; idle waves still stage + barrier

    int klen = T;
    bool uniform = false;
    if (a.lengths) {
        const int64_t L = a.lengths[b];
        if (L <= 0) uniform = true;                                // every key masked: uniform weights, no score gradient
        else if (L < T) klen = (int)L;
    }
    const int jmax = 2 * T - 2;
    const int ring_bias = RING * ((T + 512) / RING + 3);           // (j + ring_bias) >= 0 for every j touched
    const int nq = (T + 31) / 32;

    // ---- this wave's keys: K and V as MFMA B operands (lane = key, registers = head dims 8c+4hf+e), and K once more
    //      with the head dim on the lane (B operand of d(Q+u) = dS.K)
    float kreg[4 * NC], vreg[4 * NC], k2reg[ND][16];
    {
        const int key = min(k0 + li, T - 1);
        const float* krow = a.k + ((int64_t)b * T + key) * a.ld + h * dh;
        const float* vrow = a.v + ((int64_t)b * T + key) * a.ld + h * dh;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int dd = 8 * c + 4 * hf;
            f32x4 kk = {0.f, 0.f, 0.f, 0.f}, vv = kk;
            if (dd < dh) { kk = *reinterpret_cast<const f32x4*>(krow + dd); vv = *reinterpret_cast<const f32x4*>(vrow + dd); }
#pragma unroll
            for (int e = 0; e < 4; ++e) { kreg[4 * c + e] = round16(kk[e], a.prec); vreg[4 * c + e] = round16(vv[e], a.prec); }
        }
#pragma unroll
        for (int nt = 0; nt < ND; ++nt)
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int key2 = min(k0 + rho(s, hf), T - 1), cd = 32 * nt + li;
                k2reg[nt][s] = cd < dh ? a.k[((int64_t)b * T + key2) * a.ld + h * dh + cd] : 0.f;
            }
    }

    // ---- cooperative staging: thread -> (row srow + 16*pass, 16-byte chunk sch) of a 32-row tile
    const int srow = tid >> 4, sch = tid & 15;
    const bool sok = sch * 4 < dh;
    f32x4 ubias = {0.f, 0.f, 0.f, 0.f}, vbias = ubias;
    if (sok) {
        ubias = *reinterpret_cast<const f32x4*>(a.u + h * dh + sch * 4);
        vbias = *reinterpret_cast<const f32x4*>(a.vb + h * dh + sch * 4);
    }
    f32x4 pq[2], po[2], pdo[2], ppr[2];       // raw prefetched rows: nothing below touches them until commit(), so the loads
    float plse[2];                            // stay in flight behind a whole query tile of MFMAs
    auto jlo_of = [&](int i0) { return T - 32 - i0 + kb; };          // lowest table row of the workgroup's window
    auto prefetch = [&](int it) {
        const int i0 = 32 * it;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int r = srow + 16 * p, i = i0 + r;
            const int64_t row = (int64_t)b * T + min(i, T - 1);
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            pq[p] = z; po[p] = z; pdo[p] = z; ppr[p] = z;
            if (sok) {
                pq[p] = *reinterpret_cast<const f32x4*>(a.q + row * a.ld + h * dh + sch * 4);
                po[p] = *reinterpret_cast<const f32x4*>(a.o + row * a.ldo + h * dh + sch * 4);
                pdo[p] = *reinterpret_cast<const f32x4*>(a.dout + row * a.ldo + h * dh + sch * 4);
                const int j = max(0, min(jlo_of(i0) + r, jmax));                   // (used from the second tile on)
                ppr[p] = *reinterpret_cast<const f32x4*>(a.pos + (int64_t)j * a.ldp + h * dh + sch * 4);
            }
            plse[p] = a.lse[(int64_t)bh * T + min(i, T - 1)];
        }
    };
    auto commit = [&](int it, bool ring_rows) {
        const int i0 = 32 * it;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int r = srow + 16 * p;
            const bool use = i0 + r < T && sok;                                    // rows past T: zero operands, P = exp(s - inf) = 0
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(Qu + r * QROW + sch * 4) = use ? round16(pq[p] + ubias, a.prec) : z;
            *reinterpret_cast<f32x4*>(Qv + r * QROW + sch * 4) = use ? round16(pq[p] + vbias, a.prec) : z;
            *reinterpret_cast<f32x4*>(dOs + r * QROW + sch * 4) = use ? round16(pdo[p], a.prec) : z;
            // D_i = dO_i . O_i.  Under autocast (prec != 0: the small-head fallback of the 16-bit kernel) dO is rounded here and in
            // dW = dO.V^T alike, so that sum_k P (dW - D) = 0 holds to the accuracy of P (see attention_bwd_flash_mfma16.hip)
            const f32x4 dor = round16(pdo[p], a.prec);
            float dot = use ? po[p][0] * dor[0] + po[p][1] * dor[1] + po[p][2] * dor[2] + po[p][3] * dor[3] : 0.f;
            dot += __shfl_xor(dot, 8, 64); dot += __shfl_xor(dot, 4, 64);
            dot += __shfl_xor(dot, 2, 64); dot += __shfl_xor(dot, 1, 64);       // the 16 chunks of a row sit in 16 adjacent lanes
            if (sch == 0) { lseS[r] = i0 + r < T ? plse[p] : INFINITY; DS[r] = dot; }
            if (ring_rows) {
                const int slot = (jlo_of(i0) + r + ring_bias) % RING;
                *reinterpret_cast<f32x4*>(Pr + slot * QROW + sch * 4) = round16(ppr[p], a.prec);
            }
        }
    };
    // LDS-only workgroup barrier: __syncthreads() also drains vmcnt, i.e. it would wait for every outstanding global atomic
    // (2-4 us per query tile in the phase trace); LDS visibility only needs lgkmcnt(0).
    // (No __builtin_amdgcn_fence here: a workgroup-scope release also waits for vmcnt.  The asm memory clobbers keep the
    // compiler from moving or caching LDS accesses across the barrier.)
    auto lds_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    // ---- prologue: the 160 table rows of the first query tile, zeroed accumulation ring + dq tile, query tile 0
    {
        const int jlo = jlo_of(0);
        for (int idx = tid; idx < RING * 16; idx += 256) {
            const int r = idx >> 4, ch = idx & 15;
            const int j = max(0, min(jlo + r, jmax));
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            const f32x4 val = ch * 4 < dh ? *reinterpret_cast<const f32x4*>(a.pos + (int64_t)j * a.ldp + h * dh + ch * 4) : z;
            *reinterpret_cast<f32x4*>(Pr + ((jlo + r + ring_bias) % RING) * QROW + ch * 4) = round16(val, a.prec);
        }
    }
    for (int idx = tid; idx < 4 * 32 * 16; idx += 256) *reinterpret_cast<f32x4*>(dQs + idx * 4) = f32x4{0.f, 0.f, 0.f, 0.f};   // (idle waves never write theirs)
    prefetch(0);
    commit(0, false);
    __syncthreads();
    if (nq > 1) prefetch(1);

    f32x16 dKacc[ND], dVacc[ND];
#pragma unroll
    for (int n = 0; n < ND; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dKacc[n][r] = 0.f; dVacc[n][r] = 0.f; }
    f32x16 dPcarry[ND];                     // this wave's dPband rows below the completed tile: next query tile's upper band tile
#pragma unroll
    for (int n = 0; n < ND; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) dPcarry[n][r] = 0.f;
    float du_acc[ND], dv_acc[ND];
#pragma unroll
    for (int n = 0; n < ND; ++n) { du_acc[n] = 0.f; dv_acc[n] = 0.f; }
    const float inv_keep = a.drop_p > 0.f ? 1.0f / (1.0f - a.drop_p) : 1.0f;
    const float inv_T = 1.0f / (float)T;

    const bool tracer = a.trace && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0;
#define ATB_STAMP(i) do { if (tracer) a.trace[16 * it + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
    for (int it = 0; it < nq; ++it) {
        const int i0 = 32 * it;
        ATB_STAMP(0);
        if (wactive) {
            const int jtop = T + 30 - i0 + k0;                       // table row of band row jj: j = jtop - jj
            const int slot0 = (jtop + ring_bias) % RING;             // slot(jtop - jj) = slot0 - jj (+RING if negative)
            f32x16 S, dW;
            float sk[16], w[16], ds[16];
            // ---- (1) content scores S[il][kl] = (Q+u).K^T : A = staged rows (lane = query), B = key registers
            {
                ATB_FRESH_LANE();
#pragma unroll
                for (int r = 0; r < 16; ++r) S[r] = 0.f;
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const f32x4 aq = *reinterpret_cast<const f32x4*>(Qu + li * QROW + 8 * c + 4 * hf);
#pragma unroll
                    for (int e = 0; e < 4; ++e) S = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[e], kreg[4 * c + e], S, 0, 0, 0);
                }
            }
            ATB_STAMP(1);
            // ---- (3) positional band G[il][jj] = (Q+v).Pband^T, jj = il - kl + 31; then the "relative shift":
            //      S[il][kl] += G[il][il - kl + 31], a per-row rotation through the per-wave LDS tile.  The LDS round trip of
            //      band tile 0 runs under the MFMAs of band tile 1, that of tile 1 under (2) dW = dO.V^T.
            f32x16 G0, G1;
            auto band = [&](f32x16& G, int mt) {
                ATB_FRESH_LANE();
                int slot = slot0 - (32 * mt + li);
                slot += slot < 0 ? RING : 0;
                const float* prow = Pr + slot * QROW + 4 * hf;
#pragma unroll
                for (int r = 0; r < 16; ++r) G[r] = 0.f;
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const f32x4 av = *reinterpret_cast<const f32x4*>(Qv + li * QROW + 8 * c + 4 * hf);
                    const f32x4 bp = *reinterpret_cast<const f32x4*>(prow + 8 * c);
#pragma unroll
                    for (int e = 0; e < 4; ++e) G = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bp[e], G, 0, 0, 0);
                }
            };
            // The "relative shift" S[il][kl] += G[il][il - kl + 31]: with the key on the lane it is, per accumulator register (one
            // query row per lane half), a ROTATION of the 32 lanes of that half -- one ds_bpermute_b32 per register and band
            // tile through the LDS crossbar, no LDS memory, no wave fence (the first version spilled G to a per-wave LDS
            // tile and read it back skewed: 1.0-1.4 us of write / fence / read latency per query tile).
            auto unskew = [&](const f32x16& G, int mt) {
                ATB_FRESH_LANE();
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int il = rho(r, hf);
                    const int src = ((il - li + 31) & 31) + 32 * hf;          // lane holding column jj & 31 of this half's row
                    const float gr = G[r];                                   // (bit_cast straight from the vector element picks element 0)
                    const float val = __int_as_float(__builtin_amdgcn_ds_bpermute(src << 2, __float_as_int(gr)));
                    sk[r] = mt == 0 ? val : (il > li ? val : sk[r]);         // jj >= 32  <=>  il > kl : second band tile
                }
            };
            band(G0, 0);
            band(G1, 1);                                            // MFMAs in flight while the rotated reads of tile 0 return
            unskew(G0, 0);
            // ---- (2) dW = dO.V^T
            {
                ATB_FRESH_LANE();
#pragma unroll
                for (int r = 0; r < 16; ++r) dW[r] = 0.f;
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const f32x4 ad = *reinterpret_cast<const f32x4*>(dOs + li * QROW + 8 * c + 4 * hf);
#pragma unroll
                    for (int e = 0; e < 4; ++e) dW = __builtin_amdgcn_mfma_f32_32x32x2f32(ad[e], vreg[4 * c + e], dW, 0, 0, 0);
                }
            }
            unskew(G1, 1);
            __builtin_amdgcn_sched_barrier(0);
            ATB_STAMP(2);
            // ---- (4) probabilities and (5) score gradient: lane = key, register r = query row rho(r, hf)
            {
                ATB_FRESH_LANE();
                const bool kvalid = k0 + li < klen;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const f32x4 l4 = *reinterpret_cast<const f32x4*>(lseS + 8 * gq + 4 * hf);
                    const f32x4 d4 = *reinterpret_cast<const f32x4*>(DS + 8 * gq + 4 * hf);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 4 * gq + e;
                        const float sc = (S[r] + sk[r]) * a.scale;
                        float p = kvalid ? exp_fast(sc - l4[e]) : 0.f;
                        if (uniform) p = (k0 + li < T && l4[e] < INFINITY) ? inv_T : 0.f;
                        w[r] = p;
                        ds[r] = uniform ? 0.f : p * (dW[r] - d4[e]) * a.scale;
                    }
                }
                if (a.drop_p > 0.f) {                                  // weights were dropped after normalisation (attention.py:67)
                    const unsigned long long rowbase = ((unsigned long long)bh * T + (unsigned)(i0 + 4 * hf)) * (unsigned long long)T +
                                                       (unsigned)(k0 + li);
#pragma unroll 1
                    for (int gq = 0; gq < 4; ++gq)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float m = dropout_keep(a.drop_seed, rowbase + (unsigned long long)(8 * gq + e) * (unsigned long long)T,
                                                         a.drop_p, inv_keep);
                            // W = P o M ; dS = P o (dW o M - D) * scale = ds + P * dW * (m - 1) * scale
#pragma unroll
                            for (int g2 = 0; g2 < 4; ++g2)
                                if (g2 == gq) {
                                    if (!uniform) ds[4 * g2 + e] += w[4 * g2 + e] * dW[4 * g2 + e] * (m - 1.0f) * a.scale;
                                    w[4 * g2 + e] *= m;
                                }
                        }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            ATB_STAMP(3);
            // ---- (6) dV^T[c][kl] += dO^T.W and (7) dK^T[c][kl] += (Q+u)^T.dS : contraction over the query row = register axis
            //      (groups of 4 steps fenced with sched_barrier: left alone, the scheduler hoists all 64 operand reads of the
            //      phase to its top and spills hundreds of registers)
#pragma unroll
            for (int sg = 0; sg < 4; ++sg) {
                ATB_FRESH_LANE();
#pragma unroll
                for (int nt = 0; nt < ND; ++nt)
#pragma unroll
                    for (int s = 4 * sg; s < 4 * sg + 4; ++s) {
                        const int off = rho(s, hf) * QROW + 32 * nt + li;
                        dVacc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(dOs[off], w[s], dVacc[nt], 0, 0, 0);
                        dKacc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(Qu[off], ds[s], dKacc[nt], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            ATB_STAMP(4);
            // ---- (8) dS -> per-wave LDS tile; d(Q+u)[il][c] = dS.K : A = dS^T read (lane = query), B = K with the dim on the lane
            f32x16 dQ[ND];
            {
                ATB_FRESH_LANE();
#pragma unroll
                for (int r = 0; r < 16; ++r) gs[rho(r, hf) * SROW + li] = ds[r];
                wave_lds_fence();
#pragma unroll
                for (int n = 0; n < ND; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) dQ[n][r] = 0.f;
                float av[16];                                       // operand block first, then the MFMA block: left to itself the
#pragma unroll                                                      // compiler reads one operand, waits lgkmcnt(0), issues two MFMAs ...
                for (int s = 0; s < 16; ++s) av[s] = gs[li * SROW + rho(s, hf)];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < 16; ++s)
#pragma unroll
                    for (int nt = 0; nt < ND; ++nt) dQ[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], k2reg[nt][s], dQ[nt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nt = 0; nt < ND; ++nt) {
                    float cs = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) cs += dQ[nt][r];
                    du_acc[nt] += cs;
                }
            }
            ATB_STAMP(5);
            // ---- (9) d(Q+v)[il][c] = sum_jj dG[il][jj] Pband[jj][c], dG[il][jj] = dS[il][il - jj + 31] (0 outside the tile).
            //      Per band tile: an operand block (16 diagonal reads of the dS tile + 32 table reads, issued back to back), then
            //      the MFMA block.  (Left to itself the compiler reads one operand, waits lgkmcnt(0), issues two MFMAs ...; issuing
            //      the NEXT unit's operand block inside the current MFMA block -- a second register set -- spilled and was slower.)
            {
                f32x16 dQp[ND];
#pragma unroll
                for (int n = 0; n < ND; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) dQp[n][r] = 0.f;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    float av[16], bv[16][ND];
                    {
                        ATB_FRESH_LANE();
#pragma unroll
                        for (int s = 0; s < 16; ++s) {
                            const int jj = 32 * mt + rho(s, hf);
                            const int kl = li - jj + 31;
                            const float raw = gs[li * SROW + (kl & 31)];
                            av[s] = (unsigned)kl < 32u ? raw : 0.f;
                            int slot = slot0 - jj;
                            slot += slot < 0 ? RING : 0;
#pragma unroll
                            for (int nt = 0; nt < ND; ++nt) bv[s][nt] = Pr[slot * QROW + 32 * nt + li];
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int s = 0; s < 16; ++s)
#pragma unroll
                        for (int nt = 0; nt < ND; ++nt)
                            dQp[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[s][nt], dQp[nt], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                ATB_FRESH_LANE();
#pragma unroll
                for (int nt = 0; nt < ND; ++nt) {                   // dq = d(Q+u) + d(Q+v) -> this wave's slab; column sums for dv
                    float cs = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        cs += dQp[nt][r];
                        dQs[(wave * 32 + rho(r, hf)) * 64 + 32 * nt + li] = dQ[nt][r] + dQp[nt][r];
                    }
                    dv_acc[nt] += cs;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            ATB_STAMP(6);
            // ---- (10) dPband[jj][c] = sum_il dG[il][jj] (Q+v)[il][c].  The upper band tile (jj < 32) starts from the carry --
            //      the lower tile of the previous query tile covers the same table rows -- and is then complete for this wave:
            //      it goes straight from the accumulators to dpos (fp32 atomics, two 128-byte row segments per instruction);
            //      the lower tile becomes the new carry.  (ds_add_f32 into a shared LDS ring was tried first: LDS float atomics
            //      retire about one lane per 2.5 cycles and cost 458 of 843 us.)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                f32x16 dPb[ND];
#pragma unroll
                for (int n = 0; n < ND; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) dPb[n][r] = mt == 0 ? dPcarry[n][r] : 0.f;
                float av[16], bv[16][ND];
                {
                    ATB_FRESH_LANE();
#pragma unroll
                    for (int s = 0; s < 16; ++s) {
                        const int il = rho(s, hf);
                        const int kl = il - (32 * mt + li) + 31;
                        const float raw = gs[il * SROW + (kl & 31)];
                        av[s] = (unsigned)kl < 32u ? raw : 0.f;
#pragma unroll
                        for (int nt = 0; nt < ND; ++nt) bv[s][nt] = Qv[il * QROW + 32 * nt + li];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < 16; ++s)
#pragma unroll
                    for (int nt = 0; nt < ND; ++nt)
                        dPb[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[s][nt], dPb[nt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (mt == 0) {
                    // UNCONDITIONAL atomics (clamped address, value 0 where the row / column does not exist): a conditional one
                    // makes the count of outstanding memory operations unknown to the compiler, and the next commit() of
                    // prefetched rows then waits vmcnt(0) -- for these atomics -- instead of a counted vmcnt (1-3 us per tile).
                    ATB_FRESH_LANE();
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int j = jtop - rho(r, hf);
                        const bool jok = j >= 0 && j <= jmax;
                        float* prow = a.dpos + (int64_t)(jok ? j : 0) * a.lddp + h * dh;
#pragma unroll
                        for (int nt = 0; nt < ND; ++nt) {
                            const int cd = 32 * nt + li;
                            const bool ok = jok && cd < dh;
                            glob_add(prow + (cd < dh ? cd : 0), ok ? dPb[nt][r] : 0.f);
                        }
                    }
                } else {
#pragma unroll
                    for (int n = 0; n < ND; ++n) dPcarry[n] = dPb[n];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            wave_lds_fence();
        }
        ATB_STAMP(7);
        lds_barrier();                                         // every wave is done with this query tile; its dq slab is written
        ATB_STAMP(8);
        if (it + 1 < nq) commit(it + 1, true);                 // next query tile + its 32 new table rows (the freed ring slots)
        ATB_STAMP(9);
        // ---- flush: the four waves' dq contributions summed -> global (atomics: the other key blocks of this (b,h) add to
        //      the same rows), one full 256-byte row per wave-instruction
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int idx = p * 256 + tid, r = idx >> 6, c = idx & 63;
            const float val = (dQs[r * 64 + c] + dQs[(32 + r) * 64 + c]) + (dQs[(64 + r) * 64 + c] + dQs[(96 + r) * 64 + c]);
            const bool ok = i0 + r < T && c < dh;                    // (unconditional, as for dpos above)
            glob_add(a.dq + ((int64_t)b * T + min(i0 + r, T - 1)) * a.ldg + h * dh + (c < dh ? c : 0), ok ? val : 0.f);
        }
        lds_barrier();
        ATB_STAMP(10);
        if (it + 2 < nq) prefetch(it + 2);
        ATB_STAMP(11);
    }
#undef ATB_STAMP

    // ---- epilogue: the carried dPband rows (below the last query tile's band), dK / dV of this wave's keys, du / dv
    if (wactive) {
        const int jtop = T + 30 - 32 * (nq - 1) + k0 - 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = jtop - rho(r, hf);
#pragma unroll
            for (int nt = 0; nt < ND; ++nt) {
                const int cd = 32 * nt + li;
                if (j >= 0 && j <= jmax && cd < dh && dPcarry[nt][r] != 0.f)
                    glob_add(a.dpos + (int64_t)j * a.lddp + h * dh + cd, dPcarry[nt][r]);
            }
        }
    }
    if (wactive && k0 + li < T) {
        float* dkrow = a.dk + ((int64_t)b * T + k0 + li) * a.ldg + h * dh;
        float* dvrow = a.dv + ((int64_t)b * T + k0 + li) * a.ldg + h * dh;
#pragma unroll
        for (int nt = 0; nt < ND; ++nt)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int dd = 32 * nt + 8 * gq + 4 * hf;
                if (dd < dh) {
                    *reinterpret_cast<f32x4*>(dkrow + dd) = f32x4{dKacc[nt][4 * gq], dKacc[nt][4 * gq + 1], dKacc[nt][4 * gq + 2], dKacc[nt][4 * gq + 3]};
                    *reinterpret_cast<f32x4*>(dvrow + dd) = f32x4{dVacc[nt][4 * gq], dVacc[nt][4 * gq + 1], dVacc[nt][4 * gq + 2], dVacc[nt][4 * gq + 3]};
                }
            }
    }
    if (wactive) {
#pragma unroll
        for (int nt = 0; nt < ND; ++nt) {
            const float su = du_acc[nt] + __shfl_xor(du_acc[nt], 32, 64);
            const float sv = dv_acc[nt] + __shfl_xor(dv_acc[nt], 32, 64);
            const int cd = 32 * nt + li;
            if (hf == 0 && cd < dh) {
                glob_add(a.du + h * dh + cd, su);
                glob_add(a.dvb + h * dh + cd, sv);
            }
        }
    }
}

}  // namespace

static unsigned long long* g_atb_trace = nullptr;     // diagnostics only, set by cfm_debug_attention_bwd_trace_f32

// Backward of cfm_relpos_attention_train_f32 (same q/k/v/pos/u/vbias/lengths/drop arguments), given the forward's
// context `ctx` (B,T,H*dh; row stride ldo), its log-sum-exp `lse` (B,H,T) and the context gradient `dctx` (layout of ctx).
// dq / dk / dv: row stride ldg (e.g. the three column slots of one (B*T, 3d) buffer).  dq, dpos (2T-1 rows, stride lddp),
// du and dvbias (H*dh each) are ACCUMULATED INTO with fp32 atomics: the caller zero-fills them; dk / dv are written.
// prec: CFM_PREC_F32, or the 16-bit type the forward ran in (cfm_relpos_attention_mfma16_f32): q+u, q+v, K, V and the table
// rows are then rounded to it before the score / dW recompute (fp32 MFMAs throughout; everything else stays unrounded).
extern "C" int cfm_relpos_attention_bwd_f32(const float* q, const float* k, const float* v, int64_t ld, const float* pos,
                                            int64_t ldp, const float* u, const float* vbias,
                                            const int64_t* lengths_or_null, const float* ctx, const float* dctx, int64_t ldo,
                                            const float* lse, float* dq, float* dk, float* dv, int64_t ldg, float* dpos,
                                            int64_t lddp, float* du, float* dvbias, int B, int T, int H, int dh,
                                            float drop_p, uint64_t drop_seed, int prec, cfm_stream_t stream) {
    CFM_REQUIRE(q && k && v && pos && u && vbias && ctx && dctx && lse && dq && dk && dv && dpos && du && dvbias, CFM_ERR_NULL);
    CFM_REQUIRE(B > 0 && T > 0 && H > 0 && dh > 0 && (dh & 3) == 0, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(dh <= 64, CFM_ERR_UNSUPPORTED);
    CFM_REQUIRE((ld & 3) == 0 && (ldp & 3) == 0 && (ldo & 3) == 0 && (ldg & 3) == 0 && (lddp & 3) == 0, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(ld >= dh && ldp >= dh && ldo >= dh && ldg >= dh && lddp >= dh, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(drop_p >= 0.f && drop_p < 1.f, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(prec == CFM_PREC_F32 || prec == CFM_PREC_BF16 || prec == CFM_PREC_FP16, CFM_ERR_UNSUPPORTED);
    CFM_REQUIRE(CFM_ALIGNED16(q) && CFM_ALIGNED16(k) && CFM_ALIGNED16(v) && CFM_ALIGNED16(pos) && CFM_ALIGNED16(u) &&
                CFM_ALIGNED16(vbias) && CFM_ALIGNED16(ctx) && CFM_ALIGNED16(dctx) && CFM_ALIGNED16(dq) && CFM_ALIGNED16(dk) &&
                CFM_ALIGNED16(dv) && CFM_ALIGNED16(dpos), CFM_ERR_ALIGN);
    CFM_REQUIRE((int64_t)B * H <= 65535 && T < (1 << 24), CFM_ERR_UNSUPPORTED);
    AttnBwdArgs a{q, k, v, ld, pos, ldp, u, vbias, lengths_or_null, ctx, dctx, ldo, lse, dq, dk, dv, ldg, dpos, lddp, du, dvbias,
                  B, T, H, dh, 1.0f / sqrtf((float)dh), drop_p, drop_seed, prec, g_atb_trace};
    const dim3 grid((unsigned)((T + 127) / 128), (unsigned)(B * H)), block(256);
    hipStream_t s = static_cast<hipStream_t>(stream);
#define ATB_LAUNCH(NC, ND) hipLaunchKernelGGL((relpos_attn_bwd_kernel<NC, ND>), grid, block, 0, s, a)
    if (dh <= 8) ATB_LAUNCH(1, 1);
    else if (dh <= 16) ATB_LAUNCH(2, 1);
    else if (dh <= 32) ATB_LAUNCH(4, 1);
    else if (dh <= 40) ATB_LAUNCH(5, 2);
    else ATB_LAUNCH(8, 2);
#undef ATB_LAUNCH
    return cfm_launch_status();
}

// diagnostics only (tools/attn_bwd_bench.py trace): the NEXT cfm_relpos_attention_bwd_f32 launches record s_memrealtime
// (100 MHz) stamps of wave 0 of workgroup (0,0) at 12 phase boundaries per query tile into trace[16*tile + phase]
// (16*ceil(T/32) uint64); pass NULL to switch it off again.  Not thread-safe, not for production use.
extern "C" int cfm_debug_attention_bwd_trace_f32(void* trace_or_null) {
    g_atb_trace = static_cast<unsigned long long*>(trace_or_null);
    return CFM_OK;
}
