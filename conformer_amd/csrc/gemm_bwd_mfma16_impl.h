// Implementation of the 16-bit-MFMA backward GEMM, compiled once per element type (gemm_bwd_mfma16_bf16.hip /
// gemm_bwd_mfma16_f16.hip define CFM_T16 and CFM_T16_FN and include this file) so that the two halves of the instantiation
// set build in parallel.  See gemm_bwd_mfma16.hip for the description and the C entry points.
#include "gemm_bwd_args.h"

namespace {

// B16 (contraction-major B only): B is already stored in the 16-bit type (cached weight cast for dX = dY.W; a 16-bit
// activation for dW): a thread loads a 4(k) x 8(index) block with four 16-byte loads and writes eight 8-byte k-runs.
// ALIGNED: Kc % 4 == 0 for index-major operands, I / J % 4 == 0 for contraction-major ones (J % 8 with B16): every 16-byte
// chunk is entirely inside or outside the valid range, so only the unconditional-load + select path is compiled.
template <typename T16, int BM, int BN, bool AROW, bool BROW, int EPI, int GATHER = 0, bool SPLITK = false, bool B16 = false,
          bool ALIGNED = false>
__global__ __launch_bounds__(256, 2) void gemm_bwd_mfma16_kernel(const BwdArgs g) {
    static_assert(!B16 || (!BROW && GATHER == 0), "16-bit B operand: contraction-major, no gather");
    using x8 = typename Lowp<T16>::x8;
    using x4 = typename Lowp<T16>::x4;
    constexpr int TM = BM / 64, TN = BN / 64, BK = 64;
    constexpr int NA = BM / 16, NB = BN / 16;                 // float4 loads per thread per tile
    // LDS image per operand and stage.  Index-major (ROW) operands: [index][64 k], 128-byte rows, XOR-swizzled 16-byte
    // blocks, fragments by ds_read_b128.  Contraction-major (COL) operands: stored AS LOADED, [64 k][index] with rows of
    // (BT + 32) elements (the 64-byte pad puts four consecutive k rows on disjoint bank ranges), and the MFMA fragment --
    // 8 consecutive k of one index -- is fetched with two ds_read_b64_tr_b16 (gfx950's transposing LDS read): no register
    // transposes and full-width (8 / 16-byte) LDS writes.
    constexpr int RSA = BM + 32, RSB = BN + 32;               // COL row strides in elements
    constexpr int AF = AROW ? BM * BK : BK * RSA, BF = BROW ? BN * BK : BK * RSB;
    __shared__ __attribute__((aligned(16))) T16 lds[2 * (AF + BF)];
    T16* As = lds;                    // [2][AF]
    T16* Bs = lds + 2 * AF;           // [2][BF]

    const unsigned nwg = g.tiles_i * g.tiles_j;
    const unsigned tile = xcd_remap(blockIdx.x, nwg);
    const unsigned ti = tile / g.tiles_j, tj = tile % g.tiles_j;
    const int i0 = (int)ti * BM, j0 = (int)tj * BN;
    const int64_t kbeg = (int64_t)blockIdx.y * g.k_per_split;
    const int64_t kend = min(g.Kc, kbeg + g.k_per_split);
    if (kbeg >= kend) return;
    const int zb0 = blockIdx.z / g.nb1, zb1 = blockIdx.z % g.nb1;
    const float* Ab = g.A + zb0 * g.sa0 + zb1 * g.sa1;
    const float* Bb = g.B + zb0 * g.sb0 + zb1 * g.sb1;
    const T16* Bh = reinterpret_cast<const T16*>(g.B) + zb0 * g.sb0 + zb1 * g.sb1;
    float* Cb = g.C + zb0 * g.sc0 + zb1 * g.sc1;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int li = lane & 31, hf = lane >> 5;

    // element (idx, k) of a tile lives at idx*64 + ((k>>3) ^ ((idx>>2)&7))*8 + (k&7)
    auto lds_off = [](int idx, int k) { return idx * 64 + ((((k >> 3) ^ (idx >> 2)) & 7) << 3) + (k & 7); };

    // GATHER 2: the class rows this thread stages (ROW mapping: row = (tid>>4) + 16 p), decoded once
    int g2_b[NA], g2_a[NA], g2_c[NA];
    if (GATHER == 2) {
#pragma unroll
        for (int p = 0; p < NA; ++p) {
            int idx = i0 + (tid >> 4) + 16 * p;
            if (idx >= g.I) idx = g.I - 1;
            const int per = g.pA * g.pC;
            g2_b[p] = idx / per;
            const int r = idx - g2_b[p] * per;
            g2_a[p] = r / g.pC;
            g2_c[p] = r - g2_a[p] * g.pC;
        }
    }

    // ---- global -> registers.  ROW: pass p = row (tid>>4) + 16p, k chunk (tid&15)*4.
    //      COL: pass p = 4x4 block: k group kg = (slot / cpr), idx chunk ch = slot % cpr, slot = tid + 256*(p>>2), k row kg*4 + (p&3)
    // Every load below is UNCONDITIONAL: out-of-range rows / taps / contraction indices read a clamped (valid) address
    // and are zeroed by a select afterwards.  (Conditional loads compile to one branch + s_waitcnt vmcnt(0) per load: the
    // first version of this kernel had 160-280 branches and ~80 full memory waits per tile around its 4-16 MFMAs.)
    // The select-based path needs 16-byte chunks that are entirely inside or outside the valid range: Kc % 4 == 0 for
    // index-major operands, IDX % 4 == 0 for contraction-major ones; ragged shapes take the element-wise path.
    const bool kfast = ALIGNED || (g.Kc & 3) == 0;
    // (the zeroing select is applied at staging time, AFTER the MFMAs of the current tile: `keep` carries one bit per
    // register; selecting right after the load would make the wave wait for the load before it multiplies)
    auto load_operand = [&](auto& regs, unsigned& keep, const float* X, int64_t ld, int idx0, int IDX, bool row, int bt,
                            int64_t k0, int gather) {
        constexpr int NV = sizeof(regs) / sizeof(f32x4);
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        keep = ~0u;
#pragma unroll
        for (int p = 0; p < NV; ++p) {
            f32x4 v = zero;
            if (gather == 2) {                                         // A rows of dz2 per (class row, tap); C % 64 == 0
                const int tap = (int)(k0 / g.cC);
                const int co = (int)(k0 - (int64_t)tap * g.cC) + (tid & 15) * 4;
                const int t2 = g2_a[p] + g.tap_dt[tap], f2 = g2_c[p] + g.tap_df[tap];
                const bool ok = t2 >= 0 && t2 < g.cT2 && f2 >= 0 && f2 < g.cF2;
                const int t2c = min(max(t2, 0), g.cT2 - 1), f2c = min(max(f2, 0), g.cF2 - 1);
                v = *reinterpret_cast<const f32x4*>(X + (((int64_t)g2_b[p] * g.cT2 + t2c) * g.cF2 + f2c) * g.cC + co);
                if (!ok) keep &= ~(1u << p);
            } else if (row) {
                const int idx = min(idx0 + (tid >> 4) + 16 * p, IDX - 1);
                const int64_t k = k0 + (tid & 15) * 4;
                if (ALIGNED || kfast) {
                    v = *reinterpret_cast<const f32x4*>(X + (int64_t)idx * ld + min(k, g.Kc - 4));
                    if (k >= kend) keep &= ~(1u << p);
                } else if (k + 3 < kend) {
                    v = *reinterpret_cast<const f32x4*>(X + (int64_t)idx * ld + k);
                } else if (k < kend) {                                 // ragged end of the contraction
                    const float* s = X + (int64_t)idx * ld + k;
                    v.x = s[0];
                    if (k + 1 < kend) v.y = s[1];
                    if (k + 2 < kend) v.z = s[2];
                }
            } else {
                const int cpr = bt >> 2;                               // idx chunks per k row (32 or 16)
                const int slot = tid + 256 * (p >> 2);
                const int kg = slot / cpr, ch = slot - kg * cpr;
                const int64_t k = k0 + kg * 4 + (p & 3);
                const int idx = idx0 + 4 * ch;
                const int64_t kc = min(k, kend - 1);
                const bool ok = k < kend && idx < IDX;
                if (gather == 1) {                                     // B = im2col(h1): k = output position m, idx = (tap, ci)
                    const int idc = min(idx, IDX - 4);                 // IDX = 9C: chunks are never ragged
                    const int f2 = (int)(kc % g.cF2);
                    const int64_t bt2 = kc / g.cF2;
                    const int t2 = (int)(bt2 % g.cT2);
                    const int64_t b = bt2 / g.cT2;
                    const int tap = idc / g.cC, ci = idc - tap * g.cC;
                    const int kf = tap / 3, ktp = tap - 3 * kf;
                    v = *reinterpret_cast<const f32x4*>(
                        X + (((b * g.cT1 + 2 * t2 + ktp) * g.cF1 + 2 * f2 + kf) * (int64_t)g.cC) + ci);
                    if (!ok) keep &= ~(1u << p);
                } else if (ALIGNED || (IDX & 3) == 0) {
                    v = *reinterpret_cast<const f32x4*>(X + kc * ld + min(idx, IDX - 4));
                    if (!ok) keep &= ~(1u << p);
                } else if (ok) {
                    const float* s = X + k * ld + idx;
                    if (idx + 3 < IDX) v = *reinterpret_cast<const f32x4*>(s);
                    else {                                             // ragged index edge: zero-filled
                        v.x = s[0];
                        if (idx + 1 < IDX) v.y = s[1];
                        if (idx + 2 < IDX) v.z = s[2];
                    }
                }
            }
            regs[p] = v;
        }
    };
    // ---- registers -> LDS (16-bit)
    auto store_operand = [&](const auto& regs_in, unsigned keep, T16* S, bool row, int bt) {
        constexpr int NV = sizeof(regs_in) / sizeof(f32x4);
        f32x4 regs[NV];
#pragma unroll
        for (int p = 0; p < NV; ++p) regs[p] = ((keep >> p) & 1u) ? regs_in[p] : f32x4{0.f, 0.f, 0.f, 0.f};
        if (row) {
#pragma unroll
            for (int p = 0; p < NV; ++p)
                *reinterpret_cast<x4*>(S + lds_off((tid >> 4) + 16 * p, (tid & 15) * 4)) = Lowp<T16>::cvt4(regs[p]);
        } else {
            const int cpr = bt >> 2;
#pragma unroll
            for (int q = 0; q < NV / 4; ++q) {
                const int slot = tid + 256 * q;
                const int kg = slot / cpr, ch = slot - kg * cpr;
#pragma unroll
                for (int j = 0; j < 4; ++j)                            // k row kg*4+j, indices 4ch .. 4ch+3: as loaded
                    *reinterpret_cast<x4*>(S + (kg * 4 + j) * (bt + 32) + 4 * ch) = Lowp<T16>::cvt4(regs[4 * q + j]);
            }
        }
    };
    // 16-bit contraction-major B: slot = (k group kg of 4 rows, 8-index chunk ch); BN/8 chunks per k row, 16 k groups
    constexpr int CPR8 = BN / 8;
    const int hkg = tid / CPR8, hch = tid - hkg * CPR8;       // threads >= 16*CPR8 idle (BN = 64)
    x8 rbh[4];                                                 // (dead and eliminated when !B16)
    unsigned keep_a = ~0u, keep_b = ~0u;
    auto load_b16 = [&](int64_t k0) {
        keep_b = ~0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t k = k0 + hkg * 4 + j;
            const int idx = j0 + 8 * hch;
            x8 zero8;
#pragma unroll
            for (int e = 0; e < 8; ++e) zero8[e] = (T16)0.f;
            rbh[j] = zero8;
            const bool ok = hkg < 16 && k < kend && idx < g.J;
            if (ALIGNED || (g.J & 7) == 0) {                       // unconditional clamped load + select (see load_operand)
                rbh[j] = *reinterpret_cast<const x8*>(Bh + min(k, kend - 1) * g.ldb + min(idx, g.J - 8));
                if (!ok) keep_b &= ~(1u << j);
            } else if (ok) {
                const T16* s = Bh + k * g.ldb + idx;
                if (idx + 7 < g.J) rbh[j] = *reinterpret_cast<const x8*>(s);
                else
                    for (int e = 0; e < 8 && idx + e < g.J; ++e) rbh[j][e] = s[e];     // ragged index edge
            }
        }
    };
    auto store_b16 = [&](T16* S) {
        if (hkg >= 16) return;
        x8 z8;
#pragma unroll
        for (int e = 0; e < 8; ++e) z8[e] = (T16)0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *reinterpret_cast<x8*>(S + (hkg * 4 + j) * RSB + 8 * hch) = ((keep_b >> j) & 1u) ? rbh[j] : z8;
    };
    f32x4 ra[NA], rb[B16 ? 1 : NB];
    // ---- FAST path (ALIGNED, no gather): every load has a running 64-bit pointer that advances by a constant per K-tile;
    // the only per-tile work is one 32-bit "still inside the split?" compare per load and a select between the running and
    // a precomputed always-valid pointer.  (The general path re-derives each address with 64-bit min / multiply / add:
    // ~10 VALU instructions per load, which made the 64x64 variants instruction-bound at 4 MFMAs per tile.)
    // Per-operand fast paths (ALIGNED).  Plain operands: running pointers.  Stem gathers: the index arithmetic that does not
    // change from tile to tile is hoisted -- GATHER 1 (B = im2col(h1)): the (tap, ci) part is a per-thread constant and the
    // output position (b, t2, f2) of a thread's first k row is advanced by 64 rows per tile with carries instead of being
    // re-derived with two 64-bit divisions per load; GATHER 2 (A = dz2 rows per tap): the row pointer is recomputed only when
    // the contraction crosses into the next tap (every C/64 tiles) and advanced by 64 otherwise.
    constexpr bool FAST_A = ALIGNED && GATHER != 2, FAST_B = ALIGNED && GATHER != 1;
    constexpr bool FAST_G1 = ALIGNED && GATHER == 1, FAST_G2 = ALIGNED && GATHER == 2;
    const int klen = (int)(kend - kbeg);
    const float* pa_run[NA]; const float* pa_safe[NA]; int ka_row[NA]; unsigned ia_ok = 0;
    const float* pb_run[B16 ? 1 : NB]; const float* pb_safe[B16 ? 1 : NB]; int kb_row[B16 ? 1 : NB]; unsigned ib_ok = 0;
    const T16* ph_run[4]; const T16* ph_safe[4]; bool ih_ok = false;
    int64_t step_a = 0, step_b = 0;
    auto init = [&](const float* X, int64_t ld, int idx0, int IDX, bool row, int bt, auto& run, auto& safe, auto& krow,
                    unsigned& iok, int64_t& step) {
        constexpr int NV = sizeof(run) / sizeof(run[0]);
        step = row ? BK : (int64_t)BK * ld;
#pragma unroll
        for (int p = 0; p < NV; ++p) {
            int kr, idx; bool ok = true;
            int64_t off_idx;
            if (row) {
                idx = min(idx0 + (tid >> 4) + 16 * p, IDX - 1);
                kr = (tid & 15) * 4;
                off_idx = (int64_t)idx * ld;
                const int64_t kc = min(kbeg + kr, (g.Kc - 1) & ~(int64_t)3);      // (ragged Kc under pad4: the padded chunk)
                run[p] = X + off_idx + kbeg + kr;
                safe[p] = X + off_idx + kc;
            } else {
                const int cpr = bt >> 2;
                const int slot = tid + 256 * (p >> 2);
                const int kg = slot / cpr, ch = slot - kg * cpr;
                kr = kg * 4 + (p & 3);
                idx = idx0 + 4 * ch;
                ok = idx < IDX;
                off_idx = min(idx, (IDX - 1) & ~3);                               // (ragged IDX under pad4: the padded chunk)
                run[p] = X + (kbeg + kr) * ld + off_idx;
                safe[p] = X + min(kbeg + kr, kend - 1) * ld + off_idx;
            }
            krow[p] = kr;
            if (ok) iok |= 1u << p;
        }
    };
    if constexpr (FAST_A) init(Ab, g.lda, i0, g.I, AROW, BM, pa_run, pa_safe, ka_row, ia_ok, step_a);
    if constexpr (FAST_B) {
        if constexpr (B16) {
            step_b = (int64_t)BK * g.ldb;
            const int idx = j0 + 8 * hch;
            ih_ok = hkg < 16 && idx < g.J;
            const int64_t off_idx = min(idx, g.J - 8);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int kr = min(hkg, 15) * 4 + j;
                ph_run[j] = Bh + (kbeg + kr) * g.ldb + off_idx;
                ph_safe[j] = Bh + min(kbeg + kr, kend - 1) * g.ldb + off_idx;
            }
        } else {
            init(Bb, g.ldb, j0, g.J, BROW, BN, pb_run, pb_safe, kb_row, ib_ok, step_b);
        }
    }
    // GATHER 1 state: per register group q (4 consecutive k rows), the output position of its first row
    constexpr int NQ = NB / 4;
    int g1_f2[NQ], g1_t2[NQ]; int64_t g1_b[NQ]; int g1_kr[NQ]; int64_t g1_tap[NQ]; unsigned g1_ok = 0;
    const int q64 = g.cF2 > 0 ? BK / g.cF2 : 0, r64 = g.cF2 > 0 ? BK - q64 * g.cF2 : 0;       // 64 rows = q64 full f2 rows + r64
    if constexpr (FAST_G1) {
        const int cpr = BN >> 2;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int slot = tid + 256 * q;
            const int kg = slot / cpr, ch = slot - kg * cpr;
            const int idx = j0 + 4 * ch, idc = min(idx, g.J - 4);
            const int tap = idc / g.cC, ci = idc - tap * g.cC;
            const int kf = tap / 3, ktp = tap - 3 * kf;
            g1_tap[q] = ((int64_t)ktp * g.cF1 + kf) * g.cC + ci;
            g1_kr[q] = kg * 4;
            const int64_t m0 = kbeg + kg * 4;
            g1_f2[q] = (int)(m0 % g.cF2);
            const int64_t bt2 = m0 / g.cF2;
            g1_t2[q] = (int)(bt2 % g.cT2);
            g1_b[q] = bt2 / g.cT2;
            if (idx < g.J) g1_ok |= 1u << q;
        }
    }
    // GATHER 2 state: row pointer per staged class row for the current tap
    const float* g2_ptr[NA]; unsigned g2_ok = 0;
    auto g2_retarget = [&](int64_t k0) {
        const int tap = (int)(k0 / g.cC);
        const int co = (int)(k0 - (int64_t)tap * g.cC) + (tid & 15) * 4;
        g2_ok = 0;
#pragma unroll
        for (int p = 0; p < NA; ++p) {
            const int t2 = g2_a[p] + g.tap_dt[tap], f2 = g2_c[p] + g.tap_df[tap];
            if (t2 >= 0 && t2 < g.cT2 && f2 >= 0 && f2 < g.cF2) g2_ok |= 1u << p;
            const int t2c = min(max(t2, 0), g.cT2 - 1), f2c = min(max(f2, 0), g.cF2 - 1);
            g2_ptr[p] = Ab + (((int64_t)g2_b[p] * g.cT2 + t2c) * g.cF2 + f2c) * g.cC + co;
        }
    };
    auto load_tile = [&](int kt) {
        const int kofs = kt * BK;
        const int64_t k0 = kbeg + (int64_t)kofs;
        if constexpr (FAST_A) {
            keep_a = ia_ok;
#pragma unroll
            for (int p = 0; p < NA; ++p) {
                const bool in = ka_row[p] + kofs < klen;
                ra[p] = *reinterpret_cast<const f32x4*>(in ? pa_run[p] : pa_safe[p]);
                if (!in) keep_a &= ~(1u << p);
                pa_run[p] += step_a;
            }
        } else if constexpr (FAST_G2) {
            if (k0 % g.cC == 0) g2_retarget(k0);                 // uniform: a new tap every C/64 tiles
            keep_a = g2_ok;
#pragma unroll
            for (int p = 0; p < NA; ++p) {
                ra[p] = *reinterpret_cast<const f32x4*>(g2_ptr[p]);
                g2_ptr[p] += BK;
            }
        } else {
            load_operand(ra, keep_a, Ab, g.lda, i0, g.I, AROW, BM, k0, GATHER == 2 ? 2 : 0);
        }
        if constexpr (FAST_B && B16) {
            keep_b = ih_ok ? 15u : 0u;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool in = min(hkg, 15) * 4 + j + kofs < klen;
                rbh[j] = *reinterpret_cast<const x8*>(in ? ph_run[j] : ph_safe[j]);
                if (!in) keep_b &= ~(1u << j);
                ph_run[j] += step_b;
            }
        } else if constexpr (FAST_B) {
            keep_b = ib_ok;
#pragma unroll
            for (int p = 0; p < NB; ++p) {
                const bool in = kb_row[p] + kofs < klen;
                rb[p] = *reinterpret_cast<const f32x4*>(in ? pb_run[p] : pb_safe[p]);
                if (!in) keep_b &= ~(1u << p);
                pb_run[p] += step_b;
            }
        } else if constexpr (FAST_G1) {
            keep_b = 0;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                int f2 = g1_f2[q], t2 = g1_t2[q];
                int64_t bb = g1_b[q];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool in = g1_kr[q] + j + kofs < klen;
                    const float* src = Bb + (((bb * g.cT1 + 2 * t2) * g.cF1 + 2 * f2) * (int64_t)g.cC) + g1_tap[q];
                    rb[4 * q + j] = *reinterpret_cast<const f32x4*>(in ? src : Bb);
                    if (in && ((g1_ok >> q) & 1u)) keep_b |= 1u << (4 * q + j);
                    if (++f2 == g.cF2) { f2 = 0; if (++t2 == g.cT2) { t2 = 0; ++bb; } }      // next output position
                }
                // advance the group's first row by 64 output positions
                int nf = g1_f2[q] + r64, nt = g1_t2[q] + q64;
                if (nf >= g.cF2) { nf -= g.cF2; ++nt; }
                while (nt >= g.cT2) { nt -= g.cT2; ++g1_b[q]; }
                g1_f2[q] = nf; g1_t2[q] = nt;
            }
        } else if constexpr (B16) {
            load_b16(k0);
        } else {
            load_operand(rb, keep_b, Bb, g.ldb, j0, g.J, BROW, BN, k0, GATHER == 1 ? 1 : 0);
        }
    };
    auto store_tile = [&](int buf) {
        store_operand(ra, keep_a, As + buf * AF, AROW, BM);
        if constexpr (B16) store_b16(Bs + buf * BF);
        else store_operand(rb, keep_b, Bs + buf * BF, BROW, BN);
    };

    // Transposing fragment read of a [k][index] image: the 16-lane group (lane >> 4) covers indices idx0 + 16*(group & 1) ..
    // +15 and k rows 16 s + 8 hf + {0..3} (first read) / {4..7} (second); lane 4q+p of the group supplies the address of
    // row q, indices 4p .. 4p+3, and receives its own index (lane & 31) at the four rows.  EXEC is all ones here.
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    auto tr_frag = [&](const T16* S, int rs, int idx0, int s) -> x8 {
        const int q = (lane >> 2) & 3, p = lane & 3, grp = (lane >> 4) & 1;
        const T16* a0 = S + (16 * s + 8 * hf + q) * rs + idx0 + 16 * grp + 4 * p;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(const_cast<T16*>(a0)));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(const_cast<T16*>(a0 + 4 * rs)));
        union { struct { s16x4 l, h; } p2; x8 v; } u;
        u.p2.l = lo; u.p2.h = hi;
        return u.v;
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // lane (index li, half hf) feeds k = 16 s + 8 hf + {0..7} of MFMA step s: one swizzled 16-byte read
    const int a_idx = wr * (BM / 2) + li, b_idx = wc * (BN / 2) + li;
    const int nkt = (int)((kend - kbeg + BK - 1) / BK);
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nkt;
        if (more) load_tile(kt + 1);
        __builtin_amdgcn_sched_barrier(0);                    // loads stay in flight across the MFMAs of this tile
        x8 fa[2][TM], fb[2][TN];                              // fragments of step s+1 are read while step s multiplies
        auto read_frags = [&](int s) {
#pragma unroll
            for (int t = 0; t < TM; ++t)
                fa[s & 1][t] = AROW ? *reinterpret_cast<const x8*>(As + cur * AF + lds_off(a_idx + 32 * t, 16 * s + 8 * hf))
                                    : tr_frag(As + cur * AF, RSA, wr * (BM / 2) + 32 * t, s);
#pragma unroll
            for (int t = 0; t < TN; ++t)
                fb[s & 1][t] = BROW ? *reinterpret_cast<const x8*>(Bs + cur * BF + lds_off(b_idx + 32 * t, 16 * s + 8 * hf))
                                    : tr_frag(Bs + cur * BF, RSB, wc * (BN / 2) + 32 * t, s);
        };
        read_frags(0);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s < 3) read_frags(s + 1);
#pragma unroll
            for (int mt = 0; mt < TM; ++mt)
#pragma unroll
                for (int nt = 0; nt < TN; ++nt)
                    acc[mt][nt] = SPLITK ? Lowp<T16>::mfma(fa[s & 1][mt], fb[s & 1][nt], acc[mt][nt])
                                         : Lowp<T16>::mfma(fb[s & 1][nt], fa[s & 1][mt], acc[mt][nt]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (more) store_tile(cur ^ 1);
        __syncthreads();
    }
    if (SPLITK) {
        bwd_epilogue_atomic<BM, BN, TM, TN>(g, Cb, acc, i0, j0, wr, wc, li, hf);
    } else if (bwd_rows_lds_ok(g, Cb, EPI)) {                 // (kernel-uniform; the loop ended on a barrier: the staging LDS is free)
        static_assert(4 * 32 * (32 * TN + 4) * 4 <= 2 * (AF + BF) * 2, "row-major epilogue scratch must fit the staging LDS");
        bwd_epilogue_rows_lds<BM, BN, EPI, GATHER, TM, TN>(g, Cb, acc, i0, j0, wr, wc, lane,
                                                           reinterpret_cast<float*>(lds) + wave * 32 * (32 * TN + 4));
    } else {
        bwd_epilogue_rows<BM, BN, EPI, GATHER, TM, TN>(g, Cb, acc, i0, j0, wr, wc, li, hf);
    }
}

template <typename T16, int BM, int BN, bool AROW, bool BROW, int EPI, int GATHER = 0, bool B16 = false>
int launch_one(BwdArgs g, hipStream_t s) {
    constexpr bool kCanSplit = EPI == BEPI_SCALE && GATHER != 2;
    g.tiles_i = (unsigned)((g.I + BM - 1) / BM);
    g.tiles_j = (unsigned)((g.J + BN - 1) / BN);
    const unsigned tiles = g.tiles_i * g.tiles_j;
    int splits = 1;
    if (g.splits == 0 && kCanSplit) {                  // 0 = auto, 1 = forbid
        while ((int64_t)tiles * g.nbatch * splits < 1024 && g.Kc / (splits * 2) >= 512 && splits < 32) splits *= 2;
    }
    g.splits = splits;
    const int64_t per = (g.Kc + splits - 1) / splits;
    g.k_per_split = (per + 63) / 64 * 64;
    const dim3 grid(tiles, (unsigned)splits, (unsigned)g.nbatch);
    // pad4: the caller guarantees that ragged extents are physically padded to a multiple of 4 with ZEROS (the attention
    // backward's (T x T4) / (T x P4) tensors), so their partial 16-byte chunks may be loaded whole
    const bool aligned = ((AROW ? (g.Kc & 3) == 0 : (g.I & 3) == 0) || g.pad4) &&
                         ((BROW ? (g.Kc & 3) == 0 : (g.J & (B16 ? 7 : 3)) == 0) || (g.pad4 && !B16));
    if constexpr (kCanSplit) {
        if (splits > 1) {
            if (aligned) hipLaunchKernelGGL((gemm_bwd_mfma16_kernel<T16, BM, BN, AROW, BROW, EPI, GATHER, true, B16, true>), grid, dim3(256), 0, s, g);
            else hipLaunchKernelGGL((gemm_bwd_mfma16_kernel<T16, BM, BN, AROW, BROW, EPI, GATHER, true, B16, false>), grid, dim3(256), 0, s, g);
            return cfm_launch_status();
        }
    }
    if (aligned) hipLaunchKernelGGL((gemm_bwd_mfma16_kernel<T16, BM, BN, AROW, BROW, EPI, GATHER, false, B16, true>), grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((gemm_bwd_mfma16_kernel<T16, BM, BN, AROW, BROW, EPI, GATHER, false, B16, false>), grid, dim3(256), 0, s, g);
    return cfm_launch_status();
}

template <typename T16, bool AROW, bool BROW, int EPI, bool B16 = false>
int launch_layout(const BwdArgs& g, hipStream_t s) {
    const int64_t t128 = (int64_t)((g.I + 127) / 128) * ((g.J + 127) / 128) * g.nbatch;
    int tile = cfm_bwd_debug_tile();
    if (tile < 0) {
        // measured (tools/gemm_tune.py bwd16, M = 7968 and 15936): dX-type products switch to 128x128 from ~400 tiles;
        // weight-gradient products (both operands contraction-major, split-K) with a narrow dY (I <= 512) want 128x64
        // (the 512 x 2048 FFN-out gradient: 113 vs 163 us at M = 15936), the wide-dY ones 64x64
        if (!AROW && !BROW && g.nbatch == 1 && g.I <= 512 && g.I >= 96 && g.J >= 512) tile = 1;
        else if (g.I < 96 || g.J < 96) tile = 3;
        else if (t128 >= 400) tile = 0;
        else if (t128 >= 200 && g.J >= 64 && (AROW || BROW)) tile = 1;        // e.g. dX 7968 x 512: 504 tiles of 128x64
        else tile = 3;
    }
    if (tile == 0) return launch_one<T16, 128, 128, AROW, BROW, EPI, 0, B16>(g, s);
    if (tile == 1) return launch_one<T16, 128, 64, AROW, BROW, EPI, 0, B16>(g, s);
    return launch_one<T16, 64, 64, AROW, BROW, EPI, 0, B16>(g, s);
}

template <typename T16>
int gemm_bwd_dispatch(BwdArgs& g, int a_col, int b_col, hipStream_t s) {
    if (g.Z) {
        g.splits = 1;
        return g.b16 ? launch_layout<T16, true, false, BEPI_DSWISH, true>(g, s) : launch_layout<T16, true, false, BEPI_DSWISH>(g, s);
    }
    if (g.b16) {
        if (!b_col) return CFM_ERR_UNSUPPORTED;
        return a_col ? launch_layout<T16, false, false, BEPI_SCALE, true>(g, s) : launch_layout<T16, true, false, BEPI_SCALE, true>(g, s);
    }
    if (!a_col && !b_col) return launch_layout<T16, true, true, BEPI_SCALE>(g, s);
    if (!a_col && b_col) return launch_layout<T16, true, false, BEPI_SCALE>(g, s);
    if (a_col && b_col) return launch_layout<T16, false, false, BEPI_SCALE>(g, s);
    return launch_layout<T16, false, true, BEPI_SCALE>(g, s);
}

template <typename T16>
int conv2_bwd_input(const float* dz2, const float* w2c, float* dh1, int B, int F1, int T1, int C, hipStream_t s) {
    const int T2 = (T1 - 1) / 2, F2 = (F1 - 1) / 2;
    const int64_t woff[4] = {0, 4, 6, 8};
    for (int pt = 0; pt < 2; ++pt)
        for (int pf = 0; pf < 2; ++pf) {
            BwdArgs g{};
            g.cT1 = T1; g.cF1 = F1; g.cT2 = T2; g.cF2 = F2; g.cC = C; g.pt = pt; g.pf = pf;
            g.pA = (T1 - pt + 1) / 2; g.pC = (F1 - pf + 1) / 2;
            if (g.pA <= 0 || g.pC <= 0) continue;
            int nt = 0;                                   // taps (kt,kf) with kt = pt (mod 2), kf = pf (mod 2); t2 = a - (kt-pt)/2
            for (int kt = pt; kt < 3; kt += 2)
                for (int kf = pf; kf < 3; kf += 2) { g.tap_dt[nt] = -(kt - pt) / 2; g.tap_df[nt] = -(kf - pf) / 2; ++nt; }
            const int q = 2 * pt + pf;
            g.A = dz2; g.B = w2c + woff[q] * C * C; g.C = dh1;
            g.I = B * g.pA * g.pC; g.J = C; g.Kc = (int64_t)nt * C;
            g.lda = 0; g.ldb = (int64_t)nt * C; g.ldc = C; g.alpha = 1.f; g.splits = 1; g.nbatch = 1; g.nb1 = 1;
            int st = launch_one<T16, 128, 128, true, true, BEPI_SCALE, 2>(g, s);
            if (st) return st;
        }
    return CFM_OK;
}


}  // namespace

#define CFM_CAT2(a, b) a##b
#define CFM_CAT(a, b) CFM_CAT2(a, b)

int CFM_CAT(cfm_bwd16_gemm_, CFM_T16_FN)(BwdArgs& g, int a_col, int b_col, hipStream_t s) {
    return gemm_bwd_dispatch<CFM_T16>(g, a_col, b_col, s);
}
int CFM_CAT(cfm_bwd16_conv2_weight_, CFM_T16_FN)(BwdArgs g, hipStream_t s) {
    return launch_one<CFM_T16, 128, 128, false, false, BEPI_SCALE, 1>(g, s);
}
int CFM_CAT(cfm_bwd16_conv2_input_, CFM_T16_FN)(const float* dz2, const float* w2c, float* dh1, int B, int F1, int T1, int C,
                                                hipStream_t s) {
    return conv2_bwd_input<CFM_T16>(dz2, w2c, dh1, B, F1, T1, C, s);
}
