// Fused (flash-style) backward of the relative-position attention core under torch.autocast: the algorithm, work split,
// band / carry logic and atomics of attention_bwd_flash_f32.hip (read that header first), with EVERY product on
// v_mfma_f32_32x32x16_{bf16,f16}: 44 matrix instructions per (32 query x 32 key) tile and wave instead of 352 fp32 ones.
// With the matrix work gone the kernel is bound by operand movement, so every operand is laid out in LDS for ONE aligned
// 16-byte (or two 8-byte) fragment read per MFMA:
//   * query-tile rows  Qu / Qv / dO     [il][c]  (144-byte rows)   A operands of S, G, dW            (contraction over c)
//   * the same, transposed             [c][il]                    A of dV^T, dK^T (two ds_read_b64 in the k-order of an
//                                                                  accumulator used as B operand), B of dPband
//   * table-row ring   Pband           [slot][c]                  B of G;      transposed [c][slot'] : B of d(Q+v), slot'
//                                                                  numbered so that a band tile is 32 consecutive slots
//   * dS of the tile (per wave)         [il][kl]                   A of d(Q+u) = dS.K
//       pre-skewed  dG[il][jj = il-kl+31] and its transpose [jj][il]: A of d(Q+v) = dG.Pband and of dPband = dG^T.(Q+v) --
//       the inverse "relative shift" is paid once, as the 2-byte scatter that writes these tiles (cells outside the band
//       parallelogram are never written and stay zero), not as masked diagonal reads per MFMA step.
//   K (both orientations) and V live in registers as 16-bit fragments for the whole kernel.
// Rounding = torch.autocast's: Q+u, Q+v, K, V, the table rows, dO, P (as W) and dS are rounded to the 16-bit type where they
// enter a product; scores, softmax, D_i, every accumulation and every output are fp32.  D_i = dO_i.O_i uses the ROUNDED dO so
// that sum_k P (dW - D) = 0 holds to the accuracy of P (the key / position projection bias gradients are that sum).
#include "cfm_common.h"
#include <math.h>

namespace {

constexpr int RING = 160;    // table rows resident per workgroup (see the fp32 kernel)
constexpr int P16 = 72;      // pitch (elements) of row-major 16-bit tiles: 144-byte rows, conflict-free ds_read_b128
constexpr int PT40 = 40;     // pitch of [c][il] tiles read with ds_read_b128 (80-byte rows)
constexpr int PT36 = 36;     // pitch of [c][il] tiles read with ds_read_b64 only (72-byte rows: conflict-free for b64)
constexpr int PRT = 168;     // pitch of the transposed ring [c][slot']
constexpr int SROW = 33;     // fp32 skew tile row

struct AttnBwd16Args {
    const float* q; const float* k; const float* v; int64_t ld;
    const float* pos; int64_t ldp; const float* u; const float* vb; const int64_t* lengths;
    const float* o; const float* dout; int64_t ldo; const float* lse;
    float* dq; float* dk; float* dv; int64_t ldg;
    float* dpos; int64_t lddp; float* du; float* dvb;
    int B, T, H, dh; float scale; float drop_p; unsigned long long drop_seed;
    unsigned long long* trace;            // diagnostics: s_memrealtime stamps of wave 0 of workgroup (0,0), 16 per query tile
};

__device__ __forceinline__ int rho(int r, int hf) { return (r & 3) + 8 * (r >> 2) + 4 * hf; }
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ void glob_add(float* p, float v) { atomicAdd(p, v); }

#define ATB_FRESH_LANE() int li = li_; int hf = hf_; asm volatile("" : "+v"(li), "+v"(hf))

// per-wave LDS region (bytes): fp32 skew tile | dS [il][kl] | dG [il][jj] | dG^T [jj][il].  The wave's fp32 dq slab (8 KB)
// aliases the first three: it is written when nothing of them is needed any more; the part of dG it clobbers is zeroed again
// after the flush.
constexpr int WV_GS = 0, WV_T1 = 32 * SROW * 4, WV_T2 = WV_T1 + 32 * PT40 * 2, WV_T3 = WV_T2 + 32 * P16 * 2,
              WV_BYTES = WV_T3 + 64 * PT40 * 2, WV_SLAB = 32 * 64 * 4;
static_assert(WV_SLAB > WV_T2 && WV_SLAB < WV_T3 && (WV_SLAB - WV_T2) % 16 == 0, "the dq slab must end inside the dG tile");
// workgroup LDS (bytes)
constexpr int O_QU = 0, O_QV = O_QU + 32 * P16 * 2, O_DO = O_QV + 32 * P16 * 2, O_PR = O_DO + 32 * P16 * 2,
              O_QUT = O_PR + RING * P16 * 2, O_DOT = O_QUT + 64 * PT36 * 2, O_QVT = O_DOT + 64 * PT36 * 2,
              O_PRT = O_QVT + 64 * PT40 * 2, O_LSE = O_PRT + 64 * PRT * 2, O_WV = O_LSE + 256, O_CH = O_WV + 4 * WV_BYTES,
              LDS_BYTES = O_CH + 3 * WV_SLAB;                       // 3 chain slabs: wave w -> wave w+1
static_assert(O_PR % 16 == 0 && O_QUT % 16 == 0 && O_DOT % 16 == 0 && O_QVT % 16 == 0 && O_PRT % 16 == 0 && O_WV % 16 == 0 &&
              WV_BYTES % 16 == 0 && WV_T1 % 16 == 0 && WV_T3 % 16 == 0, "16-byte alignment of every LDS array");
static_assert(LDS_BYTES <= 160 * 1024 && O_CH % 16 == 0 && WV_T2 % 16 == 0, "LDS budget / alignment");

template <typename T16, int NS, int ND>
__global__ __launch_bounds__(256, 1) void relpos_attn_bwd16_kernel(const AttnBwd16Args a) {
    typedef typename Lowp<T16>::x8 x8;
    typedef typename Lowp<T16>::x4 x4;
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
    T16* Qu16 = reinterpret_cast<T16*>(smem + O_QU);       // [32][P16]
    T16* Qv16 = reinterpret_cast<T16*>(smem + O_QV);
    T16* dO16 = reinterpret_cast<T16*>(smem + O_DO);
    T16* Pr16 = reinterpret_cast<T16*>(smem + O_PR);       // [RING][P16]      slot  = (j + ring_bias) % RING
    T16* QuT = reinterpret_cast<T16*>(smem + O_QUT);       // [64][PT36]
    T16* dOT = reinterpret_cast<T16*>(smem + O_DOT);       // [64][PT36]
    T16* QvT = reinterpret_cast<T16*>(smem + O_QVT);       // [64][PT40]
    T16* PrT = reinterpret_cast<T16*>(smem + O_PRT);       // [64][PRT]        slot' = (biasT - j) % RING
    float* lseS = reinterpret_cast<float*>(smem + O_LSE);  // [32]
    float* DS = lseS + 32;                                 // [32]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, hf = lane >> 5;
    const int li_ = li, hf_ = hf;
    unsigned char* wv = smem + O_WV + wave * WV_BYTES;
    T16* T1 = reinterpret_cast<T16*>(wv + WV_T1);          // [32][PT40]  dS[il][kl]
    T16* T2 = reinterpret_cast<T16*>(wv + WV_T2);          // [32][P16]   dG[il][jj]
    T16* T3 = reinterpret_cast<T16*>(wv + WV_T3);          // [64][PT40]  dG^T[jj][il]
    float* dQs = reinterpret_cast<float*>(wv);             // [32][64] fp32 dq slab of this wave (aliases gs | T1 | head of T2)
    float* chain_out = reinterpret_cast<float*>(smem + O_CH + (wave < 3 ? wave : 0) * WV_SLAB);       // [32][64] to wave + 1
    const float* chain_in = reinterpret_cast<const float*>(smem + O_CH + (wave > 0 ? wave - 1 : 0) * WV_SLAB);
    const int bh = blockIdx.y, b = bh / a.H, h = bh % a.H;
    const int T = a.T, dh = a.dh;
    const int kb = blockIdx.x * 128, k0 = kb + 32 * wave;
    const bool wactive = k0 < T;

    int klen = T;
    bool uniform = false;
    if (a.lengths) {
        const int64_t L = a.lengths[b];
        if (L <= 0) uniform = true;
        else if (L < T) klen = (int)L;
    }
    const int jmax = 2 * T - 2;
    const int ring_bias = RING * ((T + 512) / RING + 3);
    const int biasT = T + 30 + ring_bias;                          // slot'(j) = (biasT - j) % RING;  biasT - j > 0 always
    const int nq = (T + 31) / 32;

    // ---- zero this wave's dG tiles once (cells outside the band parallelogram are never written) and the chain slabs
    for (int off = lane * 16; off < WV_BYTES; off += 64 * 16) *reinterpret_cast<f32x4*>(wv + off) = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int off = tid * 16; off < 3 * WV_SLAB; off += 256 * 16) *reinterpret_cast<f32x4*>(smem + O_CH + off) = f32x4{0.f, 0.f, 0.f, 0.f};
    // dpos rows: wave w's completed band tile at query tile t and wave w+1's at t+1 are the SAME 32 table rows, so the partial
    // sums travel wave 0 -> 1 -> 2 -> 3 through the chain slabs and only the last live wave issues atomics: 4x fewer than one
    // tile per wave (134 MB per layer at cfg-2, which kept the chip's fp32 atomic units ~60 % busy and stalled the waves at
    // their 16 outstanding atomics).
    const bool chain_tail = wave == 3 || k0 + 32 >= T;             // nobody downstream: this wave's tiles go to memory

    // ---- this wave's keys as 16-bit MFMA B fragments: K, V with the key on the lane (scores, dW) and K with the head dim on
    //      the lane (d(Q+u) = dS.K)
    x8 k16[NS], v16[NS], k2[ND][2];
    {
        const int key = min(k0 + li, T - 1);
        const float* krow = a.k + ((int64_t)b * T + key) * a.ld + h * dh;
        const float* vrow = a.v + ((int64_t)b * T + key) * a.ld + h * dh;
#pragma unroll
        for (int st = 0; st < NS; ++st)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int dd = 16 * st + 8 * hf + 4 * q;
                f32x4 kk = {0.f, 0.f, 0.f, 0.f}, vv = kk;
                if (dd < dh) { kk = *reinterpret_cast<const f32x4*>(krow + dd); vv = *reinterpret_cast<const f32x4*>(vrow + dd); }
#pragma unroll
                for (int e = 0; e < 4; ++e) { k16[st][4 * q + e] = (T16)kk[e]; v16[st][4 * q + e] = (T16)vv[e]; }
            }
#pragma unroll
        for (int nt = 0; nt < ND; ++nt)
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int key2 = min(k0 + 16 * st + 8 * hf + j, T - 1), cd = 32 * nt + li;
                    k2[nt][st][j] = (T16)(cd < dh ? a.k[((int64_t)b * T + key2) * a.ld + h * dh + cd] : 0.f);
                }
    }

    // ---- cooperative staging: thread -> (row srow + 16*pass, 4-dim chunk sch) of a 32-row tile
    const int srow = tid >> 4, sch = tid & 15;
    const bool sok = sch * 4 < dh;
    f32x4 ubias = {0.f, 0.f, 0.f, 0.f}, vbias = ubias;
    if (sok) {
        ubias = *reinterpret_cast<const f32x4*>(a.u + h * dh + sch * 4);
        vbias = *reinterpret_cast<const f32x4*>(a.vb + h * dh + sch * 4);
    }
    f32x4 pq[2], po[2], pdo[2], ppr[2];
    float plse[2];
    auto jlo_of = [&](int i0) { return T - 32 - i0 + kb; };
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
                const int j = max(0, min(jlo_of(i0) + r, jmax));
                ppr[p] = *reinterpret_cast<const f32x4*>(a.pos + (int64_t)j * a.ldp + h * dh + sch * 4);
            }
            plse[p] = a.lse[(int64_t)bh * T + min(i, T - 1)];
        }
    };
    auto put_ring_row = [&](int jraw, const f32x4 val, int ch) {        // one table row chunk -> both ring images
        const x4 v16b = Lowp<T16>::cvt4(val);
        *reinterpret_cast<x4*>(Pr16 + ((jraw + ring_bias) % RING) * P16 + ch * 4) = v16b;
        const int st = (biasT - jraw) % RING;
#pragma unroll
        for (int e = 0; e < 4; ++e) PrT[(ch * 4 + e) * PRT + st] = v16b[e];
    };
    // commit = convert (registers only: touches the prefetched rows, i.e. it is where the wait for their loads lands) + store
    // (LDS only).  convert() runs BEFORE the tile's dpos atomics are issued: vmcnt is in-order, so a wait for the prefetched
    // rows placed after the atomics waits for the atomics as well (1-3 us per tile in the phase trace).
    x4 cqu[2], cqv[2], cdo[2], cpr[2];
    float cD[2], clse[2];
    auto convert = [&](int it) {
        const int i0 = 32 * it;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int r = srow + 16 * p;
            const bool use = i0 + r < T && sok;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            cqu[p] = Lowp<T16>::cvt4(use ? pq[p] + ubias : z);
            cqv[p] = Lowp<T16>::cvt4(use ? pq[p] + vbias : z);
            cdo[p] = Lowp<T16>::cvt4(use ? pdo[p] : z);
            cpr[p] = Lowp<T16>::cvt4(ppr[p]);
            float dot = 0.f;                                          // D_i from the ROUNDED dO (see the file header)
#pragma unroll
            for (int e = 0; e < 4; ++e) dot += use ? po[p][e] * (float)cdo[p][e] : 0.f;
            dot += __shfl_xor(dot, 8, 64); dot += __shfl_xor(dot, 4, 64);
            dot += __shfl_xor(dot, 2, 64); dot += __shfl_xor(dot, 1, 64);
            cD[p] = dot;
            clse[p] = i0 + r < T ? plse[p] : INFINITY;
        }
    };
    auto put_ring_row16 = [&](int jraw, const x4 v16b, int ch) {
        *reinterpret_cast<x4*>(Pr16 + ((jraw + ring_bias) % RING) * P16 + ch * 4) = v16b;
        const int st = (biasT - jraw) % RING;
#pragma unroll
        for (int e = 0; e < 4; ++e) PrT[(ch * 4 + e) * PRT + st] = v16b[e];
    };
    auto store = [&](int it, bool ring_rows) {
        const int i0 = 32 * it;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int r = srow + 16 * p;
            *reinterpret_cast<x4*>(Qu16 + r * P16 + sch * 4) = cqu[p];
            *reinterpret_cast<x4*>(Qv16 + r * P16 + sch * 4) = cqv[p];
            *reinterpret_cast<x4*>(dO16 + r * P16 + sch * 4) = cdo[p];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                QuT[(sch * 4 + e) * PT36 + r] = cqu[p][e];
                dOT[(sch * 4 + e) * PT36 + r] = cdo[p][e];
                QvT[(sch * 4 + e) * PT40 + r] = cqv[p][e];
            }
            if (sch == 0) { lseS[r] = clse[p]; DS[r] = cD[p]; }
            if (ring_rows) put_ring_row16(jlo_of(i0) + r, cpr[p], sch);
        }
    };
    auto lds_barrier = [&]() {                                       // LDS-only: __syncthreads() would also drain the atomics
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    // ---- prologue: the 160 table rows of the first query tile, query tile 0
    {
        const int jlo = jlo_of(0);
        for (int idx = tid; idx < RING * 16; idx += 256) {
            const int r = idx >> 4, ch = idx & 15;
            const int j = max(0, min(jlo + r, jmax));
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            put_ring_row(jlo + r, ch * 4 < dh ? *reinterpret_cast<const f32x4*>(a.pos + (int64_t)j * a.ldp + h * dh + ch * 4) : z, ch);
        }
    }
    prefetch(0);
    convert(0);
    store(0, false);
    __syncthreads();
    if (nq > 1) prefetch(1);

    f32x16 dKacc[ND], dVacc[ND], dPcarry[ND];
#pragma unroll
    for (int n = 0; n < ND; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dKacc[n][r] = 0.f; dVacc[n][r] = 0.f; dPcarry[n][r] = 0.f; }
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
        const int jtop = T + 30 - i0 + k0;                           // table row of band row jj: j = jtop - jj
        f32x16 dQ[ND], dPb[ND];                                      // (live across the mid-tile barrier)
        if (wactive) {
            const int slot0 = (jtop + ring_bias) % RING;             // row-major ring: slot(jtop - jj) = slot0 - jj (+RING)
            const int slotT0 = (((i0 - k0) % RING) + RING) % RING;   // transposed ring: slot'(jtop - jj) = (slotT0 + jj) % RING,
                                                                     // slotT0 a multiple of 32: a band tile never wraps
            f32x16 S, dW, G0, G1;
            float sk[16], w[16], ds[16];
            // ---- (1) S[il][kl] = (Q+u).K^T
            {
                ATB_FRESH_LANE();
#pragma unroll
                for (int r = 0; r < 16; ++r) S[r] = 0.f;
#pragma unroll
                for (int st = 0; st < NS; ++st)
                    S = Lowp<T16>::mfma(*reinterpret_cast<const x8*>(Qu16 + li * P16 + 16 * st + 8 * hf), k16[st], S);
            }
            ATB_STAMP(1);
            // ---- (3) band G[il][jj] = (Q+v).Pband^T, jj = il - kl + 31, and the "relative shift" through the fp32 skew tile;
            //      (2) dW = dO.V^T runs while the second tile's LDS round trip is in flight
            auto band = [&](f32x16& G, int mt) {
                ATB_FRESH_LANE();
                int slot = slot0 - (32 * mt + li);
                slot += slot < 0 ? RING : 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) G[r] = 0.f;
#pragma unroll
                for (int st = 0; st < NS; ++st)
                    G = Lowp<T16>::mfma(*reinterpret_cast<const x8*>(Qv16 + li * P16 + 16 * st + 8 * hf),
                                        *reinterpret_cast<const x8*>(Pr16 + slot * P16 + 16 * st + 8 * hf), G);
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
            band(G1, 1);
            unskew(G0, 0);
            {
                ATB_FRESH_LANE();
#pragma unroll
                for (int r = 0; r < 16; ++r) dW[r] = 0.f;
#pragma unroll
                for (int st = 0; st < NS; ++st)
                    dW = Lowp<T16>::mfma(*reinterpret_cast<const x8*>(dO16 + li * P16 + 16 * st + 8 * hf), v16[st], dW);
            }
            unskew(G1, 1);
            __builtin_amdgcn_sched_barrier(0);
            ATB_STAMP(2);
            // ---- (4) probabilities, (5) score gradient: lane = key, register r = query row rho(r, hf)
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
                if (a.drop_p > 0.f) {
                    const unsigned long long rowbase = ((unsigned long long)bh * T + (unsigned)(i0 + 4 * hf)) * (unsigned long long)T +
                                                       (unsigned)(k0 + li);
#pragma unroll 1
                    for (int gq = 0; gq < 4; ++gq)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float m = dropout_keep(a.drop_seed, rowbase + (unsigned long long)(8 * gq + e) * (unsigned long long)T,
                                                         a.drop_p, inv_keep);
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
            // ---- dS -> the three per-wave 16-bit tiles (2-byte scatter; the inverse relative shift happens HERE)
            x8 wb[2], dsb[2];                                        // W and dS as B fragments: registers 8s..8s+7 = k-step s
            {
                ATB_FRESH_LANE();
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int il = rho(r, hf), jj = il - li + 31;
                    const T16 d16 = (T16)ds[r];
                    wb[r >> 3][r & 7] = (T16)w[r];
                    dsb[r >> 3][r & 7] = d16;
                    T1[il * PT40 + li] = d16;
                    T2[il * P16 + jj] = d16;
                    T3[jj * PT40 + il] = d16;
                }
                wave_lds_fence();
            }
            ATB_STAMP(4);
            // ---- (6) dV^T[c][kl] += dO^T.W, (7) dK^T[c][kl] += (Q+u)^T.dS : the accumulator-as-B-operand k-order
            //      (element j of lane half hf <-> query row 16s + 8(j>>2) + 4hf + (j&3)) is met by two 8-byte reads of the
            //      transposed query tile
            {
                ATB_FRESH_LANE();
#pragma unroll
                for (int nt = 0; nt < ND; ++nt)
#pragma unroll
                    for (int st = 0; st < 2; ++st) {
                        const int c0 = (32 * nt + li) * PT36 + 16 * st + 4 * hf;
                        const x4 dlo = *reinterpret_cast<const x4*>(dOT + c0), dhi = *reinterpret_cast<const x4*>(dOT + c0 + 8);
                        const x4 qlo = *reinterpret_cast<const x4*>(QuT + c0), qhi = *reinterpret_cast<const x4*>(QuT + c0 + 8);
                        dVacc[nt] = Lowp<T16>::mfma(__builtin_shufflevector(dlo, dhi, 0, 1, 2, 3, 4, 5, 6, 7), wb[st], dVacc[nt]);
                        dKacc[nt] = Lowp<T16>::mfma(__builtin_shufflevector(qlo, qhi, 0, 1, 2, 3, 4, 5, 6, 7), dsb[st], dKacc[nt]);
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
            ATB_STAMP(5);
            // ---- (8) d(Q+u)[il][c] = dS.K ; (9) d(Q+v)[il][c] = dG.Pband
            {
                f32x16 dQp[ND];
                ATB_FRESH_LANE();
#pragma unroll
                for (int n = 0; n < ND; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { dQ[n][r] = 0.f; dQp[n][r] = 0.f; }
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    const x8 av = *reinterpret_cast<const x8*>(T1 + li * PT40 + 16 * st + 8 * hf);
#pragma unroll
                    for (int nt = 0; nt < ND; ++nt) dQ[nt] = Lowp<T16>::mfma(av, k2[nt][st], dQ[nt]);
                }
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    int base = slotT0 + 32 * mt;
                    base -= base >= RING ? RING : 0;
#pragma unroll
                    for (int st = 0; st < 2; ++st) {
                        const x8 av = *reinterpret_cast<const x8*>(T2 + li * P16 + 32 * mt + 16 * st + 8 * hf);
#pragma unroll
                        for (int nt = 0; nt < ND; ++nt)
                            dQp[nt] = Lowp<T16>::mfma(av, *reinterpret_cast<const x8*>(PrT + (32 * nt + li) * PRT + base + 16 * st + 8 * hf),
                                                      dQp[nt]);
                    }
                }
#pragma unroll
                for (int nt = 0; nt < ND; ++nt) {
                    float cu = 0.f, cv = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) { cu += dQ[nt][r]; cv += dQp[nt][r]; dQ[nt][r] += dQp[nt][r]; }
                    du_acc[nt] += cu;
                    dv_acc[nt] += cv;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            ATB_STAMP(6);
            // ---- (10a) upper dPband tile starts from the carry (own lower tile of the previous query tile) plus the chain slab
            //      of wave - 1 (its upper tile of the previous query tile): same 32 table rows
            {
                ATB_FRESH_LANE();
#pragma unroll
                for (int n = 0; n < ND; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        dPb[n][r] = dPcarry[n][r] + (wave > 0 ? chain_in[rho(r, hf) * 64 + 32 * n + li] : 0.f);
            }
        }
        if (it + 1 < nq) convert(it + 1);                           // (before the atomics below: see convert())
        lds_barrier();                                             // chain slabs: every read above precedes every write below
        if (wactive) {
            // ---- (10b) dPband[jj][c] += dG^T.(Q+v); upper tile -> chain slab of the next wave (or dpos atomics at the chain's
            //      tail / on the last query tile); lower tile -> carry
            {
                ATB_FRESH_LANE();
#pragma unroll
                for (int n = 0; n < ND; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) dPcarry[n][r] = 0.f;
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    const x8 a0 = *reinterpret_cast<const x8*>(T3 + li * PT40 + 16 * st + 8 * hf);
                    const x8 a1 = *reinterpret_cast<const x8*>(T3 + (32 + li) * PT40 + 16 * st + 8 * hf);
#pragma unroll
                    for (int nt = 0; nt < ND; ++nt) {
                        const x8 bq = *reinterpret_cast<const x8*>(QvT + (32 * nt + li) * PT40 + 16 * st + 8 * hf);
                        dPb[nt] = Lowp<T16>::mfma(a0, bq, dPb[nt]);
                        dPcarry[nt] = Lowp<T16>::mfma(a1, bq, dPcarry[nt]);
                    }
                }
            }
            wave_lds_fence();                                        // every read of gs | T1 | T2 is done: the dq slab may overwrite them
            {
                ATB_FRESH_LANE();
#pragma unroll
                for (int nt = 0; nt < ND; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) dQs[rho(r, hf) * 64 + 32 * nt + li] = dQ[nt][r];
                if (chain_tail || it + 1 == nq) {                    // wave-uniform
#pragma unroll
                    for (int r = 0; r < 16; ++r) {                   // unconditional atomics: see the fp32 kernel
                        const int j = jtop - rho(r, hf);
                        const bool jok = j >= 0 && j <= jmax;
                        float* prow = a.dpos + (int64_t)(jok ? j : 0) * a.lddp + h * dh;
#pragma unroll
                        for (int nt = 0; nt < ND; ++nt) {
                            const int cd = 32 * nt + li;
                            glob_add(prow + (cd < dh ? cd : 0), (jok && cd < dh) ? dPb[nt][r] : 0.f);
                        }
                    }
                } else {
#pragma unroll
                    for (int nt = 0; nt < ND; ++nt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) chain_out[rho(r, hf) * 64 + 32 * nt + li] = dPb[nt][r];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        ATB_STAMP(7);
        lds_barrier();                                             // every wave is done with this query tile; its dq slab is written
        ATB_STAMP(8);
        if (it + 1 < nq) store(it + 1, true);
        ATB_STAMP(9);
        // ---- flush: the four waves' dq slabs summed -> global (atomics: the other key blocks of this (b,h) add too)
        //      (excusing the chain's tail wave, which has just issued 32 dpos atomics, and sharing its rows among the other three
        //      measured 2-4 % slower: dropped)
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int idx = p * 256 + tid, r = idx >> 6, c = idx & 63;
            float val = 0.f;
#pragma unroll
            for (int wq = 0; wq < 4; ++wq)                            // (an idle wave's region still holds its start-up zeros)
                val += reinterpret_cast<const float*>(smem + O_WV + wq * WV_BYTES)[r * 64 + c];
            const bool ok = i0 + r < T && c < dh;
            glob_add(a.dq + ((int64_t)b * T + min(i0 + r, T - 1)) * a.ldg + h * dh + (c < dh ? c : 0), ok ? val : 0.f);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // this thread's slab reads have returned ...
        __builtin_amdgcn_s_barrier();                                 // ... and everybody else's
        for (int off = WV_T2 + lane * 16; off < WV_SLAB; off += 64 * 16)     // the dG cells the dq slab clobbered: zero again
            *reinterpret_cast<f32x4*>(wv + off) = f32x4{0.f, 0.f, 0.f, 0.f};
        lds_barrier();
        ATB_STAMP(10);
        if (it + 2 < nq) prefetch(it + 2);
        ATB_STAMP(11);
    }
#undef ATB_STAMP

    // ---- epilogue: carried dPband rows, dK / dV of this wave's keys, du / dv
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
        if (k0 + li < T) {
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

template <typename T16>
int launch_bwd16(const AttnBwd16Args& a, hipStream_t s) {
    const dim3 grid((unsigned)((a.T + 127) / 128), (unsigned)(a.B * a.H)), block(256);
    if (a.dh <= 16) hipLaunchKernelGGL((relpos_attn_bwd16_kernel<T16, 1, 1>), grid, block, 0, s, a);
    else if (a.dh <= 32) hipLaunchKernelGGL((relpos_attn_bwd16_kernel<T16, 2, 1>), grid, block, 0, s, a);
    else if (a.dh <= 48) hipLaunchKernelGGL((relpos_attn_bwd16_kernel<T16, 3, 2>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((relpos_attn_bwd16_kernel<T16, 4, 2>), grid, block, 0, s, a);
    return cfm_launch_status();
}

}  // namespace

static unsigned long long* g_atb16_trace = nullptr;    // diagnostics only, set by cfm_debug_attention_bwd_trace_mfma16

// The autocast form of cfm_relpos_attention_bwd_f32 (same arguments and accumulate-into convention; tensors stay fp32 in HBM):
// prec = CFM_PREC_BF16 / CFM_PREC_FP16, the type the forward kernel cfm_relpos_attention_mfma16_f32 ran in.
extern "C" int cfm_relpos_attention_bwd_mfma16_f32(int prec, const float* q, const float* k, const float* v, int64_t ld,
                                                   const float* pos, int64_t ldp, const float* u, const float* vbias,
                                                   const int64_t* lengths_or_null, const float* ctx, const float* dctx,
                                                   int64_t ldo, const float* lse, float* dq, float* dk, float* dv, int64_t ldg,
                                                   float* dpos, int64_t lddp, float* du, float* dvbias, int B, int T, int H,
                                                   int dh, float drop_p, uint64_t drop_seed, cfm_stream_t stream) {
    CFM_REQUIRE(q && k && v && pos && u && vbias && ctx && dctx && lse && dq && dk && dv && dpos && du && dvbias, CFM_ERR_NULL);
    CFM_REQUIRE(prec == CFM_PREC_BF16 || prec == CFM_PREC_FP16, CFM_ERR_UNSUPPORTED);
    CFM_REQUIRE(B > 0 && T > 0 && H > 0 && dh > 0 && (dh & 3) == 0, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(dh <= 64, CFM_ERR_UNSUPPORTED);
    CFM_REQUIRE((ld & 3) == 0 && (ldp & 3) == 0 && (ldo & 3) == 0 && (ldg & 3) == 0 && (lddp & 3) == 0, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(ld >= dh && ldp >= dh && ldo >= dh && ldg >= dh && lddp >= dh, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(drop_p >= 0.f && drop_p < 1.f, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(CFM_ALIGNED16(q) && CFM_ALIGNED16(k) && CFM_ALIGNED16(v) && CFM_ALIGNED16(pos) && CFM_ALIGNED16(u) &&
                CFM_ALIGNED16(vbias) && CFM_ALIGNED16(ctx) && CFM_ALIGNED16(dctx) && CFM_ALIGNED16(dq) && CFM_ALIGNED16(dk) &&
                CFM_ALIGNED16(dv) && CFM_ALIGNED16(dpos), CFM_ERR_ALIGN);
    CFM_REQUIRE((int64_t)B * H <= 65535 && T < (1 << 24), CFM_ERR_UNSUPPORTED);
    AttnBwd16Args a{q, k, v, ld, pos, ldp, u, vbias, lengths_or_null, ctx, dctx, ldo, lse, dq, dk, dv, ldg, dpos, lddp, du, dvbias,
                    B, T, H, dh, 1.0f / sqrtf((float)dh), drop_p, drop_seed, g_atb16_trace};
    hipStream_t s = static_cast<hipStream_t>(stream);
    return prec == CFM_PREC_BF16 ? launch_bwd16<__bf16>(a, s) : launch_bwd16<_Float16>(a, s);
}

// diagnostics only: as cfm_debug_attention_bwd_trace_f32, for the 16-bit kernel (12 stamps per query tile, wave 0 of (0,0))
extern "C" int cfm_debug_attention_bwd_trace_mfma16(void* trace_or_null) {
    g_atb16_trace = static_cast<unsigned long long*>(trace_or_null);
    return CFM_OK;
}
