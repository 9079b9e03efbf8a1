// Fused relative-position multi-head self-attention forward, fp32 (Transformer-XL style scores).
//
//   score[i,k] = ((q_i+u).k_k + (q_i+v).p_{r=i-k}) / sqrt(dh);  P = softmax_k(score);  ctx = P.V
//
// Flash-style: the (B,H,T,T) / (B,H,T,2T-1) score tensors of the reference never exist.  A workgroup = 4 waves =
// 128 query rows of one (batch, head); each wave owns 32 queries and the workgroup sweeps the keys in tiles of 32
// with an online softmax.  Every product runs on v_mfma_f32_32x32x2_f32 in the TRANSPOSED orientation (keys / band
// rows / head dims on the MFMA row axis, queries on the lane axis):
//   S^T = K . (Q+u)^T                       keys x queries        4*NC MFMAs
//   G^T = Pband . (Q+v)^T                    64-row band x queries 8*NC MFMAs; band row jj <-> r = i0-k0-31+jj
//   S^T[kk][i] += G^T[i-kk+31][i]            the "relative shift" = a per-lane column skew through a 4 KB per-wave
//                                            LDS tile in which every lane only ever touches its own bank
//   O^T += V^T . P^T                         P^T is consumed straight from the accumulator registers (the softmax
//                                            row reductions are in-lane + one cross-half shuffle)
// Staging (round 2): the K and V tiles (shared by the 4 waves) are DOUBLE-BUFFERED in LDS -- tile kt+1 is written into the
// other buffer at the start of tile kt from registers loaded a tile earlier, so a key tile costs ONE workgroup barrier (the
// single-buffered form had two around the rotation and the phase trace showed 3.9-4.2 of 9 us per key tile in them).  The LDS
// for the second buffers comes from the positional rows: a wave's band tile is 32 consecutive table rows, read straight from
// global memory (L2-resident: 1 MB per layer) into the MFMA A-operand layout at the start of the key tile and consumed after the
// content MFMAs; the 160-row LDS ring is gone.  50 KB / workgroup, 2 workgroups per CU.  K rows are padded to 272 B:
// conflict-free ds_read_b128 fragment reads.
// Key-padding mask: keys >= lengths[b] are skipped entirely (identical to the reference's finfo.min fill because exp
// underflows to exactly 0; lengths[b] <= 0 reproduces its uniform-softmax degenerate case).
#include "cfm_common.h"
#include <math.h>

namespace {

constexpr int KROW = 68;                 // padded LDS row (floats) of the K tile

struct AttnArgs {
    const float* q; const float* k; const float* v; int64_t ld;
    const float* pos; int64_t ldp; const float* u; const float* vb;
    const int64_t* lengths; float* ctx; int64_t ldo; float* lse;
    int B, T, H, dh; float inv_sqrt_dh;
    int q_begin, q_end;                             // query rows computed by this launch: [q_begin, q_end) (streaming: the new rows)
    int nsplit;                                     // > 1: the keys of every (b,h,row block) are split over nsplit workgroups, each
    float* part_ctx; float* part_lse;               //      writing a normalised partial context + its log-sum-exp (merged afterwards)
    float drop_p; unsigned long long drop_seed;     // training: dropout on the softmax weights (attention.py:67)
    unsigned long long* trace;                      // diagnostics (cfm_debug_attention_trace_f32): phase stamps of one wave
};

// NW = 4: a workgroup is 4 waves = 128 query rows; a key tile is staged, then every wave runs phase 1 (content + band products)
//         and phase 2 (skew, softmax, P.V), one barrier per key tile (rounds 1-2).
// NW = 8 (round 3): 8 waves = 256 query rows = a whole (batch, head) of the T' = 249 workload, so K / V are staged ONCE per
//         (batch, head) instead of twice, and the two halves of the workgroup run HALF A KEY TILE APART: while waves 0-3 are in
//         phase 1 of tile t (64 MFMAs), waves 4-7 -- their partners on the same SIMDs -- are in phase 2 of tile t-1 (the VALU /
//         LDS section + 32 MFMAs), and vice versa in the next interval; one barrier per interval.  The round-2 kernel put two
//         independent 4-wave workgroups on a CU: both ran the same program in the same phase (MFMA pipe contended, then idle
//         under the softmax): matrix pipe 0.53 busy (profiles/r03_pmc_mfma.json).
template <int NC, int ND, int NW>
__global__ __launch_bounds__(64 * NW, 2) void relpos_attn_fwd_kernel(const AttnArgs a) {
    constexpr int RING = NW == 8 ? 9 : 0;        // NW = 8: 32-row blocks of the positional table kept in LDS (see below)
    __shared__ __attribute__((aligned(16))) float smem[2 * 32 * KROW + 2 * 32 * 64 + NW * 32 * 32 + RING * 32 * KROW];
    float* Ksb = smem;                           // [2][32][KROW]
    float* Vsb = Ksb + 2 * 32 * KROW;            // [2][32][64]
    float* gs = Vsb + 2 * 32 * 64 + (threadIdx.x >> 6) * 1024;   // per-wave skew tile [32][32]
    [[maybe_unused]] float* ring = Vsb + 2 * 32 * 64 + NW * 32 * 32;   // [RING][32][KROW]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, hf = lane >> 5;
    const int bh = blockIdx.y, b = bh / a.H, h = bh % a.H;
    const int T = a.T, dh = a.dh;
    constexpr int QB = 32 * NW;                                          // query rows per workgroup
    const int split = a.nsplit > 1 ? (int)(blockIdx.x % (unsigned)a.nsplit) : 0;
    const int q0 = a.q_begin + (int)(a.nsplit > 1 ? blockIdx.x / (unsigned)a.nsplit : blockIdx.x) * QB;
    const int i0 = q0 + wave * 32;
    const bool active = i0 < a.q_end;                                    // wave-uniform; idle waves still stage + barrier
    const bool tracer = NW == 4 && a.trace && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0;
    if (tracer) a.trace[9] = __builtin_amdgcn_s_memrealtime();           // kernel entry

    int klen = T;
    bool uniform = false;
    if (a.lengths) {
        const int64_t L = a.lengths[b];
        if (L <= 0) uniform = true;                                // every key masked -> uniform weights
        else if (L < T) klen = (int)L;
    }
    const int ntiles_all = (klen + 31) / 32;
    const int tiles_per_split = (ntiles_all + a.nsplit - 1) / a.nsplit;
    const int kt_begin = split * tiles_per_split;                  // this workgroup's key tiles: [kt_begin, ntiles)
    const int ntiles = min(ntiles_all, kt_begin + tiles_per_split);
    if (kt_begin >= ntiles) return;                                // an empty split (short cache): uniform for the workgroup

    const float* kbase = a.k + (int64_t)b * T * a.ld + h * dh;
    const float* vbase = a.v + (int64_t)b * T * a.ld + h * dh;
    const float* pbase = a.pos + h * dh;
    const int jmax = 2 * T - 2;

    // ---- cooperative staging: thread -> (row srow + RPP*pass, 16-byte chunk sch) of a 32-row x 256-byte tile
    constexpr int RPP = 4 * NW, SP = 32 / RPP;                     // rows per pass (16 | 32), passes (2 | 1)
    const int srow = tid >> 4, sch = tid & 15;
    const bool sok = sch * 4 < dh;
    f32x4 pk[SP], pv[SP];
    auto prefetch = [&](int kt) {                                  // K/V tile kt -> registers
        const int k0 = kt * 32;
#pragma unroll
        for (int p = 0; p < SP; ++p) {
            const int r = srow + RPP * p;
            const int key = min(k0 + r, T - 1);
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            pk[p] = sok ? *reinterpret_cast<const f32x4*>(kbase + (int64_t)key * a.ld + sch * 4) : z;
            pv[p] = sok ? *reinterpret_cast<const f32x4*>(vbase + (int64_t)key * a.ld + sch * 4) : z;
        }
    };
    auto commit = [&](int buf) {                                   // registers -> LDS buffer `buf`
#pragma unroll
        for (int p = 0; p < SP; ++p) {
            const int r = srow + RPP * p;
            *reinterpret_cast<f32x4*>(Ksb + buf * 32 * KROW + r * KROW + sch * 4) = pk[p];
            *reinterpret_cast<f32x4*>(Vsb + buf * 32 * 64 + r * 64 + sch * 4) = pv[p];
        }
    };
    // NW = 8: the positional band as a RING of 32-row table blocks in LDS.  Wave w (queries q0 + 32w ..) needs, for key tile t,
    // table rows R0 + 32 (t - w) + (31 - li), R0 = T - 1 - q0: "block" t - w.  The 8 waves of a workgroup therefore read 8
    // CONSECUTIVE blocks, and a key tile brings in exactly ONE new block (32 rows x 256 B, one coalesced 16-byte load per thread,
    // requested a key tile ahead like K / V); the 4-wave form's per-wave global reads of its band rows -- 8 loads of 64 different
    // cache lines each, per wave and key tile, waited for in front of the band MFMAs -- were 1.3-1.7 us of a 7.5 us key tile
    // (profiles/r03_attn_fwd_trace.txt).  Block b lives in slot b mod 9 from the interval it is committed in until wave 7 of
    // the lagging half has read it (tile b + 7): blocks -8 .. 0 are loaded in the prologue (block -w - 1 = wave w's band tile 1 of
    // the first key tile).
    const int R0 = T - 1 - q0;
    [[maybe_unused]] f32x4 pr;
    [[maybe_unused]] auto ring_slot = [](int blk) { return ((blk % 9) + 9) % 9; };
    [[maybe_unused]] auto ring_load = [&](int blk) {              // thread (row srow, chunk sch) of block blk -> register
        const int j = max(0, min(R0 + 32 * blk + srow, jmax));     // (rows outside the table belong to pairs that do not exist)
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        pr = sok ? *reinterpret_cast<const f32x4*>(pbase + (int64_t)j * a.ldp + sch * 4) : z;
    };
    [[maybe_unused]] auto ring_commit = [&](int blk) {
        *reinterpret_cast<f32x4*>(ring + (ring_slot(blk) * 32 + srow) * KROW + sch * 4) = pr;
    };
    prefetch(kt_begin);
    commit(0);
    if constexpr (NW == 8) {
        for (int blk = -8; blk <= 0; ++blk) { ring_load(blk); ring_commit(blk); }
    }
    if (kt_begin + 1 < ntiles) {
        prefetch(kt_begin + 1);
        if constexpr (NW == 8) ring_load(1);
    }
    __syncthreads();

    // ---- (Q+u)^T and (Q+v)^T as MFMA B operands: lane (query li, half hf) holds dims 8c+4hf+e at step 4c+e
    float qu[4 * NC], qv[4 * NC];
    {
        const int qi = min(i0 + li, T - 1);
        const float* qrow = a.q + ((int64_t)b * T + qi) * a.ld + h * dh;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int dd = 8 * c + 4 * hf;
            f32x4 x = {0.f, 0.f, 0.f, 0.f}, uu = x, vv = x;
            if (dd < dh) {
                x = *reinterpret_cast<const f32x4*>(qrow + dd);
                uu = *reinterpret_cast<const f32x4*>(a.u + h * dh + dd);
                vv = *reinterpret_cast<const f32x4*>(a.vb + h * dh + dd);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) { qu[4 * c + e] = x[e] + uu[e]; qv[4 * c + e] = x[e] + vv[e]; }
        }
    }

    f32x16 o[ND];
#pragma unroll
    for (int n = 0; n < ND; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[n][r] = 0.f;
    float mrow = -INFINITY, lrow = 0.f;                              // running maximum in the log2 domain (scores * scale2), row sum
    const float scale2 = a.inv_sqrt_dh * 1.44269504088896340736f;

#define ATT_STAMP(i) do { if (tracer) a.trace[16 * kt + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
    // ---- positional band G^T[jj][query], jj = 32*mt + row <-> table row j = jbase - jj (jbase = T-1-i0+k0+31), and the
    // "relative shift": lane (query li) needs G^T[li - kk + 31][li] for its 16 keys kk -- a per-lane column skew through the
    // per-wave LDS tile (each lane only touches its own bank).  The 64-row band of a key tile is two 32-row MFMA tiles, and
    // band tile 1 of key tile kt+1 covers exactly the table rows of band tile 0 of key tile kt (jbase moves by 32): only band
    // tile 0 is computed per key tile.  Better: the skew reads address row (jj & 31) of the tile for ALL 16 keys and a
    // select picks the band tile that owns key kk (jj >> 5) -- a per-key `if` compiles to 16 serialized branch + ds_read +
    // wait sequences, measured 3 us per band -- so the 16 values read from band tile 0 of key tile kt already ARE the 16
    // values key tile kt+1 needs from its band tile 1.  They are carried in registers (skp): per key tile one band product
    // (4*NC MFMAs instead of 8*NC), one spill and 16 skew reads instead of two of each.
    // band tile mt of the key tile with base row jbase: table rows j = jbase - (32 mt + li), straight from global memory into the
    // A-operand layout (lane (row li, half hf) holds dims 8c + 4hf + e); rows outside the table belong to (query, key) pairs
    // that do not exist and read a clamped row
    auto band_load = [&](int jbase, int mt, f32x4 (&pf)[NC]) {
        const int j = max(0, min(jbase - (32 * mt + li), jmax));
#pragma unroll
        for (int c = 0; c < NC; ++c) {                               // unconditional (clamped) loads + select: no branches, no drained waits
            const f32x4 v = *reinterpret_cast<const f32x4*>(pbase + (int64_t)j * a.ldp + min(8 * c + 4 * hf, dh - 4));
            pf[c] = 8 * c + 4 * hf < dh ? v : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto band_mma = [&](const f32x4 (&pf)[NC]) {
        f32x16 ga;
#pragma unroll
        for (int r = 0; r < 16; ++r) ga[r] = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) ga = __builtin_amdgcn_mfma_f32_32x32x2f32(pf[c][e], qv[4 * c + e], ga, 0, 0, 0);
        return ga;
    };
    // NW = 8: band tile of ring block `blk`: A-operand row li <-> block row 31 - li, fragments by ds_read_b128 (272-byte rows)
    [[maybe_unused]] auto band_mma_ring = [&](int blk) {
        const float* rp = ring + (ring_slot(blk) * 32 + (31 - li)) * KROW + 4 * hf;
        f32x16 ga, gb;                       // two independent accumulation chains (even / odd 8-dim slices), summed at the end
#pragma unroll
        for (int r = 0; r < 16; ++r) { ga[r] = 0.f; gb[r] = 0.f; }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const f32x4 f = *reinterpret_cast<const f32x4*>(rp + 8 * c);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (e & 1) gb = __builtin_amdgcn_mfma_f32_32x32x2f32(f[e], qv[4 * c + e], gb, 0, 0, 0);
                else ga = __builtin_amdgcn_mfma_f32_32x32x2f32(f[e], qv[4 * c + e], ga, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) ga[r] += gb[r];
        return ga;
    };
    auto spill_band = [&](const f32x16& ga) {
#pragma unroll
        for (int r = 0; r < 16; ++r) gs[((r & 3) + 8 * (r >> 2) + 4 * hf) * 32 + li] = ga[r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto skew_reads = [&](float (&dst)[16]) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int jj = li - ((r & 3) + 8 * (r >> 2) + 4 * hf) + 31;     // 0..62
            dst[r] = gs[(jj & 31) * 32 + li];
        }
    };
    float skp[16];                                                   // band tile 1 of the current key tile, skewed (carried)
    f32x4 pf[NC];                                                    // table rows of a band tile in the A-operand layout
    if (active) {                                                    // first key tile: its band tile 1 is computed explicitly
        f32x16 g1;
        if constexpr (NW == 8) {
            g1 = band_mma_ring(-wave - 1);
        } else {
            band_load(T - 1 - i0 + 32 * kt_begin + 31, 1, pf);
            g1 = band_mma(pf);
        }
        spill_band(g1);
        skew_reads(skp);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // the reads have landed before the tile is rewritten
        __builtin_amdgcn_wave_barrier();
    }
    f32x16 sc;                                                       // content scores of the tile in flight (phase 1 -> phase 2)

    // ---- phase 1 of key tile kt: content scores S^T[key][query] and band tile 0, spilled to the wave's skew tile
    auto phase1 = [&](int kt) {
        const int k0 = kt * 32;
        const float* Ks = Ksb + ((kt - kt_begin) & 1) * 32 * KROW;
        const int jbase = T - 1 - i0 + k0 + 31;
        // the band tile's table rows land behind the content MFMAs.  (Requesting them a key tile ahead -- before P.V or right
        // after the band product -- was measured: the 32 extra live VGPRs spill at dh = 64 and the kernel is slower, 97 vs 80 us.)
        if constexpr (NW == 4) band_load(jbase, 0, pf);
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = 0.f;
        if constexpr (NW == 8) {
            f32x16 sb;                                     // second accumulation chain (odd contraction steps)
#pragma unroll
            for (int r = 0; r < 16; ++r) sb[r] = 0.f;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const f32x4 kf = *reinterpret_cast<const f32x4*>(Ks + li * KROW + 8 * c + 4 * hf);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (e & 1) sb = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qu[4 * c + e], sb, 0, 0, 0);
                    else sc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qu[4 * c + e], sc, 0, 0, 0);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[r] += sb[r];
        } else {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const f32x4 kf = *reinterpret_cast<const f32x4*>(Ks + li * KROW + 8 * c + 4 * hf);
#pragma unroll
            for (int e = 0; e < 4; ++e) sc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qu[4 * c + e], sc, 0, 0, 0);
        }
        }
        ATT_STAMP(2);
        if constexpr (NW == 8) {
            const f32x16 g0 = band_mma_ring((kt - kt_begin) - wave);
            spill_band(g0);
        } else {
            const f32x16 g0 = band_mma(pf);
            spill_band(g0);
        }
        ATT_STAMP(3);
    };
    // ---- phase 2 of key tile kt: relative shift, scale + mask, online softmax, O^T += V^T . P^T
    auto phase2 = [&](int kt) {
        const int k0 = kt * 32;
        const float* Vs = Vsb + ((kt - kt_begin) & 1) * 32 * 64;
        float skn[16];
        skew_reads(skn);
        ATT_STAMP(4);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                        // the tile is rewritten by the next key tile
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int jj = li - ((r & 3) + 8 * (r >> 2) + 4 * hf) + 31;
            sc[r] += (jj >> 5) ? skp[r] : skn[r];
            skp[r] = skn[r];
        }
        ATT_STAMP(5);
        // scale + mask, online softmax (query = lane column; keys = registers x 2 halves)
        // (log2 domain: one multiply by inv_sqrt_dh * log2 e, v_exp_f32 directly; the key-padding selects run only in the tile that
        //  holds the end of the utterance -- the same arithmetic as the pipelined form below: results are bit-identical across forms)
        float p[16];
        float tmax = -INFINITY;
        if (uniform || k0 + 32 > klen) {                        // (wave-uniform)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kk = (r & 3) + 8 * (r >> 2) + 4 * hf;
                float s = sc[r] * scale2;
                if (uniform) s = 0.f;
                if (k0 + kk >= klen) s = -INFINITY;
                p[r] = s;
                tmax = fmaxf(tmax, s);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) { p[r] = sc[r] * scale2; tmax = fmaxf(tmax, p[r]); }
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float mnew = fmaxf(mrow, tmax);                   // finite: key k0 (< klen) is always valid
        const float alpha = __builtin_amdgcn_exp2f(mrow - mnew);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { p[r] = __builtin_amdgcn_exp2f(p[r] - mnew); psum += p[r]; }
        psum += __shfl_xor(psum, 32, 64);
        lrow = lrow * alpha + psum;
        mrow = mnew;
        ATT_STAMP(6);
        if (a.drop_p > 0.f) {                                     // weights are dropped AFTER normalisation: l stays unmasked
            const float inv_keep = 1.0f / (1.0f - a.drop_p);
            const unsigned long long rowbase = ((unsigned long long)bh * T + (unsigned)(i0 + li)) * (unsigned long long)T;
#pragma unroll
            for (int q = 0; q < 4; ++q) {                         // registers 4q .. 4q+3: four consecutive keys (any alignment: T is odd)
                float keep[4];
                dropout_keep4u(a.drop_seed, rowbase + (unsigned)(k0 + 8 * q + 4 * hf), a.drop_p, inv_keep, keep);
#pragma unroll
                for (int e = 0; e < 4; ++e) p[4 * q + e] *= keep[e];
            }
        }
#pragma unroll
        for (int n = 0; n < ND; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[n][r] *= alpha;
        // O^T += V^T . P^T ; MFMA step s contracts key (s&3)+8*(s>>2)+4*hf = the key held in register s
#pragma unroll
        for (int n = 0; n < ND; ++n)
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const float vv = Vs[((s & 3) + 8 * (s >> 2) + 4 * hf) * 64 + ((32 * n + li) & 63)];
                o[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv, p[s], o[n], 0, 0, 0);
            }
        ATT_STAMP(7);
    };

    if (tracer) a.trace[10] = __builtin_amdgcn_s_memrealtime();          // prologue done (Q, first K/V tile, first band tile)
    if constexpr (NW == 4) {
        for (int kt = kt_begin; kt < ntiles; ++kt) {
            const int buf = (kt - kt_begin) & 1;
            ATT_STAMP(0);
            // stage the NEXT key tile into the other buffer (its last readers passed the barrier that ended tile kt-1), then
            // request the one after it
            if (kt + 1 < ntiles) {
                commit(buf ^ 1);
                if (kt + 2 < ntiles) prefetch(kt + 2);
            }
            ATT_STAMP(1);
            if (active) {
                phase1(kt);
                phase2(kt);
            }
            // one barrier per key tile: tile kt+1 (written above) is visible, tile kt's buffer may be overwritten next round
            if (kt + 1 < ntiles) __syncthreads();
            ATT_STAMP(8);
        }
    } else {
        // Interval iv: group A (waves 0-3) runs step iv, group B (waves 4-7) step iv-1; step 2j = phase 1 of tile j, step 2j+1 =
        // phase 2.  Tile j's K is read in intervals 2j (A) and 2j+1 (B), its V in 2j+1 (A) and 2j+2 (B); tile j+1 is committed
        // at the start of interval 2j+1 into the buffer tile j-1 left free after interval 2j.
        const int nt = ntiles - kt_begin, grp = wave >> 2;
        // (s_setprio 1 for waves 4-7, or around phase 2's softmax section, measured: 72.2 / 71.9 us against 71.4 without)
        const bool tr8 = a.trace && blockIdx.x == 0 && blockIdx.y == 0 && (tid & 255) == 0 && nt <= 15;   // waves 0 and 4
        for (int iv = 0; iv <= 2 * nt; ++iv) {
            if (tr8) a.trace[4 * iv + 2 * grp] = __builtin_amdgcn_s_memrealtime();                        // interval start
            if (iv & 1) {
                const int j = (iv + 1) >> 1;                       // tile (relative) to stage now
                if (j < nt) {
                    commit(j & 1);
                    ring_commit(j);
                    if (j + 1 < nt) {
                        prefetch(kt_begin + j + 1);
                        ring_load(j + 1);
                    }
                }
            }
            const int st = iv - grp;
            if (active && st >= 0 && st < 2 * nt) {
                const int kt = kt_begin + (st >> 1);
                if (st & 1) phase2(kt); else phase1(kt);
            }
            if (tr8) a.trace[4 * iv + 2 * grp + 1] = __builtin_amdgcn_s_memrealtime();                    // this wave's phase done
            if (iv < 2 * nt) __syncthreads();
        }
    }
#undef ATT_STAMP
    if (tracer) a.trace[11] = __builtin_amdgcn_s_memrealtime();          // key loop done

    // ---- normalise and store: lane = query row, registers = head dims (4 consecutive dims per r>>2 group)
    if (active && i0 + li < a.q_end) {
        const float inv = 1.0f / lrow;
        const int qc = a.q_end - a.q_begin, il = i0 + li - a.q_begin;
        float* orow = a.nsplit > 1 ? a.part_ctx + (((int64_t)split * a.B + b) * qc + il) * a.ldo + h * dh
                                   : a.ctx + ((int64_t)b * T + i0 + li) * a.ldo + h * dh;
#pragma unroll
        for (int n = 0; n < ND; ++n)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int dd = 32 * n + 8 * gq + 4 * hf;
                if (dd < dh) {
                    f32x4 out = {o[n][4 * gq] * inv, o[n][4 * gq + 1] * inv, o[n][4 * gq + 2] * inv, o[n][4 * gq + 3] * inv};
                    *reinterpret_cast<f32x4*>(orow + dd) = out;
                }
            }
        if (a.nsplit > 1) {
            if (hf == 0) a.part_lse[(((int64_t)split * a.B + b) * a.H + h) * qc + il] = mrow * 0.69314718055994530942f + logf(lrow);
        } else if (a.lse && hf == 0) {
            a.lse[((int64_t)b * a.H + h) * T + i0 + li] = mrow * 0.69314718055994530942f + logf(lrow);
        }
    }
    if (tracer) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        a.trace[12] = __builtin_amdgcn_s_memrealtime();                  // output stores drained
    }
}

// ---- round 3: the software-pipelined 8-wave form (inference: no weight dropout, no key split) -----------------------------
// The phase stamps of the forms above (profiles/r03_attn_fwd_trace.txt, r03_attn8_interval_trace.txt) say that the matrix pipe idles
// under every wave's OWN skew / softmax section (~250 VALU + 48 LDS instructions per key tile, ~3 us) and that pairing the two waves
// of a SIMD differently does not cover it.  Here each wave covers it itself: ONE instruction stream in which the 64 matrix
// instructions of key tile t+1 (content scores + positional band) are interleaved with the softmax of key tile t, and the 32 of
// P.V(t) with the spill / skew / shift of tile t+1 -- the scores of a tile are finished one iteration before they are consumed.
//   iteration t:   X   K(t+1), band block t+1-w  ->  sn, ga   (64 MFMAs)        ||  sc(t) -> p, alpha, l, m        (VALU)
//                  Y   O *= alpha;  O += V(t)^T . p  (32 MFMAs)                  ||  ga -> skew tile -> sn += shift  (LDS, VALU)
//                  staging: K(t+2), V(t+1), ring block t+2 from registers into the buffers last read in iteration t-1; one barrier
// 8 waves = 256 query rows (a whole (batch, head) at T' = 249), K / V double-buffered, the positional band as the LDS ring of 32-row
// table blocks (10 slots: block b is read in iterations b-1 .. b+6).  153 KB LDS, one workgroup per CU.
template <int NC, int ND>
__global__ __launch_bounds__(512, 2) void relpos_attn_fwd8p_kernel(const AttnArgs a) {
    constexpr int NW = 8, RING = 10;
    __shared__ __attribute__((aligned(16))) float smem[2 * 32 * KROW + 2 * 32 * 64 + NW * 32 * 32 + RING * 32 * KROW];
    float* Ksb = smem;                           // [2][32][KROW]
    float* Vsb = Ksb + 2 * 32 * KROW;            // [2][32][64]
    float* gs = Vsb + 2 * 32 * 64 + (threadIdx.x >> 6) * 1024;   // per-wave skew tile [32][32]
    float* ring = Vsb + 2 * 32 * 64 + NW * 32 * 32;              // [RING][32][KROW]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, hf = lane >> 5;
    const int bh = blockIdx.y, b = bh / a.H, h = bh % a.H;
    const int T = a.T, dh = a.dh;
    const int q0 = a.q_begin + (int)blockIdx.x * 256;
    const int i0 = q0 + wave * 32;
    const bool active = i0 < a.q_end;                              // wave-uniform; idle waves still stage + barrier

    int klen = T;
    bool uniform = false;
    if (a.lengths) {
        const int64_t L = a.lengths[b];
        if (L <= 0) uniform = true;
        else if (L < T) klen = (int)L;
    }
    const int nt = (klen + 31) / 32;
    const float* kbase = a.k + (int64_t)b * T * a.ld + h * dh;
    const float* vbase = a.v + (int64_t)b * T * a.ld + h * dh;
    const float* pbase = a.pos + h * dh;
    const int jmax = 2 * T - 2;
    const int R0 = T - 1 - q0;

    // ---- cooperative staging: thread (row srow, 16-byte chunk sch) of a 32-row x 256-byte tile; tiles beyond the keys read clamped rows
    const int srow = tid >> 4, sch = tid & 15;
    const bool sok = sch * 4 < dh;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 pk, pv, pr;
    auto loadK = [&](int kt) { pk = sok ? *reinterpret_cast<const f32x4*>(kbase + (int64_t)min(32 * kt + srow, T - 1) * a.ld + sch * 4) : z4; };
    auto loadV = [&](int kt) { pv = sok ? *reinterpret_cast<const f32x4*>(vbase + (int64_t)min(32 * kt + srow, T - 1) * a.ld + sch * 4) : z4; };
    auto loadR = [&](int blk) { pr = sok ? *reinterpret_cast<const f32x4*>(pbase + (int64_t)max(0, min(R0 + 32 * blk + srow, jmax)) * a.ldp + sch * 4) : z4; };
    auto ring_slot = [](int blk) { return ((blk % RING) + RING) % RING; };
    auto storeK = [&](int buf) { *reinterpret_cast<f32x4*>(Ksb + buf * 32 * KROW + srow * KROW + sch * 4) = pk; };
    auto storeV = [&](int buf) { *reinterpret_cast<f32x4*>(Vsb + buf * 32 * 64 + srow * 64 + sch * 4) = pv; };
    auto storeR = [&](int blk) { *reinterpret_cast<f32x4*>(ring + (ring_slot(blk) * 32 + srow) * KROW + sch * 4) = pr; };
    loadK(0); loadV(0); storeK(0); storeV(0);
    for (int blk = -8; blk <= 1; ++blk) { loadR(blk); storeR(blk); }
    loadK(1); storeK(1);
    loadK(2); loadV(1); loadR(2);                                  // in flight: committed in iteration 0
    __syncthreads();

    // ---- (Q+u)^T and (Q+v)^T as MFMA B operands: lane (query li, half hf) holds dims 8c+4hf+e at step 4c+e
    float qu[4 * NC], qv[4 * NC];
    {
        const int qi = min(i0 + li, T - 1);
        const float* qrow = a.q + ((int64_t)b * T + qi) * a.ld + h * dh;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int dd = 8 * c + 4 * hf;
            f32x4 x = z4, uu = z4, vv = z4;
            if (dd < dh) {
                x = *reinterpret_cast<const f32x4*>(qrow + dd);
                uu = *reinterpret_cast<const f32x4*>(a.u + h * dh + dd);
                vv = *reinterpret_cast<const f32x4*>(a.vb + h * dh + dd);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) { qu[4 * c + e] = x[e] + uu[e]; qv[4 * c + e] = x[e] + vv[e]; }
        }
    }
    f32x16 o[ND];
#pragma unroll
    for (int n = 0; n < ND; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[n][r] = 0.f;
    float mrow = -INFINITY, lrow = 0.f;                              // running maximum in the log2 domain (scores * scale2), row sum
    const float scale2 = a.inv_sqrt_dh * 1.44269504088896340736f;

    // products of key tile kt: content scores S^T[key][query] from K buffer kt & 1, band tile of ring block blk (A-operand row li <->
    // block row 31 - li)
    auto content = [&](int kt) {
        const float* Ks = Ksb + (kt & 1) * 32 * KROW + li * KROW + 4 * hf;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const f32x4 kf = *reinterpret_cast<const f32x4*>(Ks + 8 * c);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qu[4 * c + e], acc, 0, 0, 0);
        }
        return acc;
    };
    auto band = [&](int blk) {
        const float* rp = ring + (ring_slot(blk) * 32 + (31 - li)) * KROW + 4 * hf;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const f32x4 f = *reinterpret_cast<const f32x4*>(rp + 8 * c);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f[e], qv[4 * c + e], acc, 0, 0, 0);
        }
        return acc;
    };
    auto spill_band = [&](const f32x16& ga) {
#pragma unroll
        for (int r = 0; r < 16; ++r) gs[((r & 3) + 8 * (r >> 2) + 4 * hf) * 32 + li] = ga[r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto skew_reads = [&](float (&dst)[16]) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int jj = li - ((r & 3) + 8 * (r >> 2) + 4 * hf) + 31;     // 0..62
            dst[r] = gs[(jj & 31) * 32 + li];
        }
    };
    // sn += the shifted band: band tile 1 of the key tile (carried in skp: it was band tile 0 of the previous key tile) or band tile 0
    auto shift_add = [&](f32x16& sn, float (&skp)[16], const float (&skn)[16]) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int jj = li - ((r & 3) + 8 * (r >> 2) + 4 * hf) + 31;
            sn[r] += (jj >> 5) ? skp[r] : skn[r];
            skp[r] = skn[r];
        }
    };

    float skp[16];
    f32x16 sc;                                                       // finished scores of the tile the softmax works on
    if (active) {
        // key tile 0: band tile 1 = block -w-1 (explicitly), then content + band tile 0 (block -w)
        {
            const f32x16 g1 = band(-wave - 1);
            spill_band(g1);
            skew_reads(skp);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        sc = content(0);
        const f32x16 g0 = band(-wave);
        spill_band(g0);
        float skn[16];
        skew_reads(skn);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        shift_add(sc, skp, skn);
    }
    __syncthreads();                   // iteration 0 overwrites K buffer 0 (K(0)) and ring slot 2 (block -8), both read just above

    for (int t = 0; t < nt; ++t) {
        // ---- staging: the registers loaded one iteration ago go into the buffers whose last readers passed the previous barrier
        storeK(t & 1);                 // K(t+2): K(t) was read in iteration t-1
        storeV((t + 1) & 1);           // V(t+1): V(t-1) was read in iteration t-1
        storeR(t + 2);
        loadK(t + 3); loadV(t + 2); loadR(t + 3);
        if (active) {
            const int k0 = 32 * t;
            // ---- X: the products of key tile t+1 (the compiler interleaves them with the softmax below: one basic block, no fences)
            f32x16 sn = content(t + 1);
            const f32x16 ga = band(t + 1 - wave);
            // (VALU / LDS issue and the fp32 matrix instructions of a SIMD do not overlap on this part -- SQ_VALU_MFMA_COEXEC_CYCLES is 0
            //  for every fp32 kernel -- so every instruction of this section is time: the scores live in the log2 domain (one multiply
            //  by inv_sqrt_dh * log2(e) instead of a scale and a second multiply inside exp), and the key-padding selects run only in
            //  the tile that holds the end of the utterance)
            float p[16];
            float tmax = -INFINITY;
            if (uniform || k0 + 32 > klen) {                        // (wave-uniform) the last key tile, or an all-masked utterance
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kk = (r & 3) + 8 * (r >> 2) + 4 * hf;
                    float sv = sc[r] * scale2;
                    if (uniform) sv = 0.f;
                    if (k0 + kk >= klen) sv = -INFINITY;
                    p[r] = sv;
                    tmax = fmaxf(tmax, sv);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) { p[r] = sc[r] * scale2; tmax = fmaxf(tmax, p[r]); }
            }
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            const float mnew = fmaxf(mrow, tmax);                   // finite: key k0 (< klen) is always valid
            const float alpha = __builtin_amdgcn_exp2f(mrow - mnew);
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { p[r] = __builtin_amdgcn_exp2f(p[r] - mnew); psum += p[r]; }
            psum += __shfl_xor(psum, 32, 64);
            lrow = lrow * alpha + psum;
            mrow = mnew;
#pragma unroll
            for (int n = 0; n < ND; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[n][r] *= alpha;
            // ---- Y: P.V of key tile t, with the spill / skew / shift of tile t+1's band riding beside its MFMAs
            spill_band(ga);
            float skn[16];
            skew_reads(skn);
            const float* Vs = Vsb + (t & 1) * 32 * 64;
#pragma unroll
            for (int n = 0; n < ND; ++n)
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const float vv = Vs[((s & 3) + 8 * (s >> 2) + 4 * hf) * 64 + ((32 * n + li) & 63)];
                    o[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv, p[s], o[n], 0, 0, 0);
                }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the skew reads have landed before the tile is rewritten
            __builtin_amdgcn_wave_barrier();
            shift_add(sn, skp, skn);
            sc = sn;
        }
        __syncthreads();
    }

    // ---- normalise and store: lane = query row, registers = head dims (4 consecutive dims per r>>2 group)
    if (active && i0 + li < a.q_end) {
        const float inv = 1.0f / lrow;
        float* orow = a.ctx + ((int64_t)b * T + i0 + li) * a.ldo + h * dh;
#pragma unroll
        for (int n = 0; n < ND; ++n)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int dd = 32 * n + 8 * gq + 4 * hf;
                if (dd < dh) {
                    f32x4 out = {o[n][4 * gq] * inv, o[n][4 * gq + 1] * inv, o[n][4 * gq + 2] * inv, o[n][4 * gq + 3] * inv};
                    *reinterpret_cast<f32x4*>(orow + dd) = out;
                }
            }
        if (a.lse && hf == 0) a.lse[((int64_t)b * a.H + h) * T + i0 + li] = mrow * 0.69314718055994530942f + logf(lrow);
    }
}

// ctx[b, q_begin+il, h*dh + :] = sum_s exp(lse_s - lse) * part_s, lse = logsumexp over the splits that own keys of utterance b
__global__ __launch_bounds__(256) void attn_merge_splits_kernel(const AttnArgs a) {
    const int qc = a.q_end - a.q_begin, d4 = a.dh / 4;
    const int64_t total = (int64_t)a.B * qc * a.H * d4;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c4 = (int)(idx % d4);
    const int h = (int)((idx / d4) % a.H);
    const int il = (int)((idx / ((int64_t)d4 * a.H)) % qc);
    const int b = (int)(idx / ((int64_t)d4 * a.H * qc));
    int klen = a.T;
    if (a.lengths) { const int64_t L = a.lengths[b]; if (L > 0 && L < a.T) klen = (int)L; }
    const int ntiles_all = (klen + 31) / 32, tps = (ntiles_all + a.nsplit - 1) / a.nsplit;
    const int live = (ntiles_all + tps - 1) / tps;                 // splits with kt_begin < ntiles_all
    float lmax = -INFINITY;
    for (int s = 0; s < live; ++s) lmax = fmaxf(lmax, a.part_lse[(((int64_t)s * a.B + b) * a.H + h) * qc + il]);
    float wsum = 0.f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < live; ++s) {
        const float w = exp_fast(a.part_lse[(((int64_t)s * a.B + b) * a.H + h) * qc + il] - lmax);
        const f32x4 p = *reinterpret_cast<const f32x4*>(a.part_ctx + (((int64_t)s * a.B + b) * qc + il) * a.ldo + h * a.dh + 4 * c4);
        acc = acc + p * w;
        wsum += w;
    }
    const float inv = 1.0f / wsum;
    *reinterpret_cast<f32x4*>(a.ctx + ((int64_t)b * a.T + a.q_begin + il) * a.ldo + h * a.dh + 4 * c4) = acc * inv;
}

}  // namespace

static int attention_launch(const float* q, const float* k, const float* v, int64_t ld, const float* pos, int64_t ldp,
                            const float* u, const float* vbias, const int64_t* lengths_or_null, float* ctx, int64_t ldo,
                            float* lse_or_null, int B, int T, int H, int dh, float drop_p, uint64_t drop_seed,
                            cfm_stream_t stream, void* trace = nullptr, int q_begin = 0, int q_count = -1, int nsplit = 1,
                            float* workspace = nullptr);

// diagnostics (tools/attn_probe.py): force the workgroup shape of the fp32 attention forward (0 = built-in choice | 4 | 8 waves);
// returns the previous setting.  The two shapes agree to fp32 rounding (the 8-wave form sums its score products in two chains).
static int g_attn_force_nw = 0;
extern "C" int cfm_debug_set_attention_waves(int nw) {
    const int prev = g_attn_force_nw;
    if (nw == 0 || nw == 4 || nw == 8 || nw == 9) g_attn_force_nw = nw;
    return prev;
}

// diagnostics only (tools/attn_probe.py trace): as cfm_relpos_attention_fwd_f32, plus s_memrealtime (100 MHz) stamps of
// wave 0 of workgroup (0,0) at 9 phase boundaries of every key tile: trace[16*tile + phase], 16*ceil(T/32) uint64.
extern "C" int cfm_debug_attention_trace_f32(const float* q, const float* k, const float* v, int64_t ld, const float* pos,
                                             int64_t ldp, const float* u, const float* vbias,
                                             const int64_t* lengths_or_null, float* ctx, int64_t ldo, int B, int T, int H,
                                             int dh, void* trace, cfm_stream_t stream) {
    const int prev = g_attn_force_nw;
    if (prev != 8) g_attn_force_nw = 4;                // per-phase stamps: the 4-wave form; cfm_debug_set_attention_waves(8): per-interval stamps
    else g_attn_force_nw = 8;
    const int st = attention_launch(q, k, v, ld, pos, ldp, u, vbias, lengths_or_null, ctx, ldo, nullptr, B, T, H, dh, 0.f, 0, stream,
                                    trace);
    g_attn_force_nw = prev;
    return st;
}

extern "C" int cfm_relpos_attention_fwd_f32(const float* q, const float* k, const float* v, int64_t ld,
                                            const float* pos, int64_t ldp, const float* u, const float* vbias,
                                            const int64_t* lengths_or_null, float* ctx, int64_t ldo,
                                            float* lse_or_null, int B, int T, int H, int dh, cfm_stream_t stream) {
    return attention_launch(q, k, v, ld, pos, ldp, u, vbias, lengths_or_null, ctx, ldo, lse_or_null, B, T, H, dh, 0.f, 0,
                            stream);
}

// Streaming / incremental form: only the query rows [q_begin, q_begin + q_count) are computed (ctx rows outside are left
// untouched); K / V / the positional table are those of the whole (B,T,.) buffers and `lengths` bounds the visible keys --
// rows of a growing K/V cache attend to everything cached so far (conformer_amd/streaming.py, BASELINE cfg-5).
// nsplit > 1 (flash-decoding form, for long caches and few query rows): the key tiles of every (b, h, 128-row block) are
// divided over nsplit workgroups that write normalised partial contexts + log-sum-exps into `workspace`
// (nsplit*B*q_count*(H*dh + H) floats; ldo must equal H*dh, lengths required, every length >= 1), merged by a second kernel.
extern "C" int cfm_relpos_attention_rows_f32(const float* q, const float* k, const float* v, int64_t ld,
                                             const float* pos, int64_t ldp, const float* u, const float* vbias,
                                             const int64_t* lengths_or_null, float* ctx, int64_t ldo, int B, int T, int H,
                                             int dh, int q_begin, int q_count, int nsplit, float* workspace_or_null,
                                             cfm_stream_t stream) {
    return attention_launch(q, k, v, ld, pos, ldp, u, vbias, lengths_or_null, ctx, ldo, nullptr, B, T, H, dh, 0.f, 0, stream,
                            nullptr, q_begin, q_count, nsplit, workspace_or_null);
}

// training variant: dropout with probability drop_p on the softmax weights (mask index ((b*H+h)*T + i)*T + k)
extern "C" int cfm_relpos_attention_train_f32(const float* q, const float* k, const float* v, int64_t ld,
                                              const float* pos, int64_t ldp, const float* u, const float* vbias,
                                              const int64_t* lengths_or_null, float* ctx, int64_t ldo, float* lse,
                                              int B, int T, int H, int dh, float drop_p, uint64_t drop_seed,
                                              cfm_stream_t stream) {
    CFM_REQUIRE(drop_p >= 0.f && drop_p < 1.f, CFM_ERR_BAD_SHAPE);
    return attention_launch(q, k, v, ld, pos, ldp, u, vbias, lengths_or_null, ctx, ldo, lse, B, T, H, dh, drop_p, drop_seed,
                            stream);
}

static int attention_launch(const float* q, const float* k, const float* v, int64_t ld, const float* pos, int64_t ldp,
                            const float* u, const float* vbias, const int64_t* lengths_or_null, float* ctx, int64_t ldo,
                            float* lse_or_null, int B, int T, int H, int dh, float drop_p, uint64_t drop_seed,
                            cfm_stream_t stream, void* trace, int q_begin, int q_count, int nsplit, float* workspace) {
    CFM_REQUIRE(q && k && v && pos && u && vbias && ctx, CFM_ERR_NULL);
    CFM_REQUIRE(B > 0 && T > 0 && H > 0 && dh > 0 && (dh & 3) == 0, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE((ld & 3) == 0 && (ldp & 3) == 0 && (ldo & 3) == 0, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(dh <= 64, CFM_ERR_UNSUPPORTED);
    CFM_REQUIRE(CFM_ALIGNED16(q) && CFM_ALIGNED16(k) && CFM_ALIGNED16(v) && CFM_ALIGNED16(pos) && CFM_ALIGNED16(u) &&
                CFM_ALIGNED16(vbias) && CFM_ALIGNED16(ctx), CFM_ERR_ALIGN);
    CFM_REQUIRE((int64_t)B * H <= 65535 && T < (1 << 28), CFM_ERR_UNSUPPORTED);
    if (q_count < 0) q_count = T - q_begin;
    CFM_REQUIRE(q_begin >= 0 && q_count > 0 && q_begin + q_count <= T, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(nsplit >= 1 && nsplit <= 16 && (nsplit == 1 || (workspace && lengths_or_null && !lse_or_null && ldo == (int64_t)H * dh)),
                CFM_ERR_BAD_SHAPE);
    AttnArgs a{q, k, v, ld, pos, ldp, u, vbias, lengths_or_null, ctx, ldo, lse_or_null, B, T, H, dh, 1.0f / sqrtf((float)dh),
               q_begin, q_begin + q_count, nsplit, workspace,
               workspace ? workspace + (int64_t)nsplit * B * q_count * ldo : nullptr, drop_p, drop_seed,
               static_cast<unsigned long long*>(trace)};
    // 8-wave workgroups (256 query rows, the two halves half a key tile apart) once a launch has more than 128 query rows and
    // no key split; g_attn_force_nw (diagnostics) overrides
    // force 9 = the software-pipelined 8-wave form (inference only: no weight dropout / key split)
    const bool can_pipe = nsplit == 1 && drop_p == 0.f && !trace;
    // (after the instruction diet of the softmax section the half-tile-staggered 8-wave form is the fastest again: 69.8 us against
    //  71.4 us pipelined and 74.2 us with 4 waves, profiles/r03_attn_ab.txt; force 9 selects the pipelined form)
    int nw = g_attn_force_nw ? g_attn_force_nw : ((q_count > 128 && nsplit == 1) ? 8 : 4);
    if (nw == 9 && !can_pipe) nw = 8;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (nw == 9) {
        const dim3 grid9((unsigned)((q_count + 255) / 256), (unsigned)(B * H)), block9(512);
#define ATT_LAUNCH9(NC, ND) hipLaunchKernelGGL((relpos_attn_fwd8p_kernel<NC, ND>), grid9, block9, 0, s, a)
        if (dh <= 8) ATT_LAUNCH9(1, 1);
        else if (dh <= 16) ATT_LAUNCH9(2, 1);
        else if (dh <= 32) ATT_LAUNCH9(4, 1);
        else if (dh <= 40) ATT_LAUNCH9(5, 2);
        else ATT_LAUNCH9(8, 2);
#undef ATT_LAUNCH9
        return cfm_launch_status();
    }
    const dim3 grid((unsigned)((q_count + 32 * nw - 1) / (32 * nw)) * nsplit, (unsigned)(B * H)), block(64 * nw);
#define ATT_LAUNCH(NC, ND) do { if (nw == 8) hipLaunchKernelGGL((relpos_attn_fwd_kernel<NC, ND, 8>), grid, block, 0, s, a); \
                                else hipLaunchKernelGGL((relpos_attn_fwd_kernel<NC, ND, 4>), grid, block, 0, s, a); } while (0)
    if (dh <= 8) ATT_LAUNCH(1, 1);
    else if (dh <= 16) ATT_LAUNCH(2, 1);
    else if (dh <= 32) ATT_LAUNCH(4, 1);
    else if (dh <= 40) ATT_LAUNCH(5, 2);
    else ATT_LAUNCH(8, 2);
#undef ATT_LAUNCH
    if (nsplit > 1) {
        const int64_t total = (int64_t)B * q_count * H * (dh / 4);
        hipLaunchKernelGGL(attn_merge_splits_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
    }
    return cfm_launch_status();
}
