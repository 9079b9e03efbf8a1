// Weight-gradient GEMM of a Linear layer on the 16-bit matrix pipe, compiled once per element type (gemm_dw16_bf16.hip /
// gemm_dw16_f16.hip define CFM_T16 / CFM_T16_FN and include this file):
//
//     dW[i][j] += alpha * sum_m r16(dY[m][i]) * r16(X[m][j])          db[i] += alpha * sum_m dY[m][i]   (optional)
//
// (reference: the autograd of nn.Linear under autocast, ffn.py:15-23 / attention.py / convolution.py pointwise convs.)
// Both operands are contraction-major -- row m of dY and of X is contiguous -- and either may already be stored in the
// 16-bit type (A16 / B16: activations and gradients the producer wrote as bf16/fp16 because a GEMM is their only consumer).
//
// Why a second kernel next to gemm_bwd_mfma16_impl.h: that kernel prefetches ONE K-tile ahead in registers and, for these
// products (contraction over M = B*T' = 8-16 k rows, output a few hundred tiles at most), ran at ~0.12 of the matrix pipe:
// a K-tile's 16 MFMAs per wave (512 cycles) do not cover an L2 round trip, and the 16-way split-K it needed to fill the chip
// cost 16 M float atomics per 2048 x 512 gradient.  Here:
//   * 128 x 128 output tile, 4 waves (2 x 2, 64 x 64 each), contraction tile 64;
//   * operands are requested TWO tiles ahead into two register sets (a fp32 operand is rounded when it is written to LDS);
//     LDS holds two stages in the as-loaded [k][index] layout and fragments are fetched with ds_read_b64_tr_b16, one
//     LDS-only barrier per tile (__syncthreads() would drain the prefetch);
//   * addressing is a per-thread 32-bit offset from a uniform running base (one VGPR per load instead of two 64-bit
//     pointers), the ragged last tile is handled by clamping to row 0 of the tile + a zeroing select at staging time;
//   * split-K only as far as needed for ~2 workgroups per CU; the bias gradient is accumulated by the workgroups of the first
//     column tile from the dY registers they stage anyway (no separate column-sum pass over dY).
// Requirements (checked by the launcher; otherwise the caller falls back to the general kernel): I % 8 == 0, J % 8 == 0,
// 16-byte aligned bases and leading dimensions, (k_per_split + 64) * ld * sizeof(elem) < 2^31.
#include <stdlib.h>

#include <type_traits>

#include "cfm_common.h"

namespace {

struct DwArgs {
    const void* A; const void* B; float* C; float* colsum;
    int64_t lda, ldb, ldc;            // in elements of the operand's stored type
    int I, J; int64_t Kc, k_per_split;
    float alpha;
    unsigned tiles_i, tiles_j;
    unsigned long long* trace;        // diagnostics: s_memrealtime stamps of thread 0 of workgroups 0 and nwg/2 (64 each)
    // GATHER (B operand = im2col of the stem's h1 for the conv2 weight gradient): row m of B is position rowtab[m] of h1 (rows
    // of gC elements); column j = (tap, ci) adds ((tap % 3) * gF1 + tap / 3) * gC + ci.  rowtab has Kc rounded up to 64, + 64 entries.
    const int* rowtab; int gC, gF1;
};

template <typename T16, bool A16, bool B16, bool GATHER = false>
__global__ __launch_bounds__(256, 2) void gemm_dw16_kernel(const DwArgs g) {
    static_assert(!GATHER || B16, "the im2col gather is built for a 16-bit h1");
    using x8 = typename Lowp<T16>::x8;
    using x4 = typename Lowp<T16>::x4;
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    constexpr int BM = 128, BN = 128, BK = 64, RS = BM + 32;          // LDS row stride (elements): see gemm_bwd_mfma16_impl.h
    constexpr int STAGE = BK * RS;
    __shared__ __attribute__((aligned(16))) T16 lds[2 * 2 * STAGE];     // [stage][A | B][k][index]

    const unsigned nwg = g.tiles_i * g.tiles_j;
    const unsigned tile = xcd_remap(blockIdx.x, nwg);
    const unsigned ti = tile / g.tiles_j, tj = tile % g.tiles_j;
    const int i0 = (int)ti * BM, j0 = (int)tj * BN;
    const int64_t kbeg = (int64_t)blockIdx.y * g.k_per_split;
    const int64_t kend = min(g.Kc, kbeg + g.k_per_split);
    if (kbeg >= kend) return;
    const int klen = (int)(kend - kbeg);
    const int nkt = (klen + BK - 1) / BK;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int li = lane & 31, hf = lane >> 5;

    // ---- loader mapping.  16-bit operand: thread = (k group kg = tid >> 4, 8-index chunk ch = tid & 15), rows 4 kg + j.
    //      fp32 operand: slot = tid + 256 q (q = 0, 1): kg = slot >> 5, 4-index chunk ch = slot & 31, rows 4 kg + j.
    constexpr int NLA = A16 ? 4 : 8, NLB = B16 ? 4 : 8;
    using RegA = typename std::conditional<A16, x8, f32x4>::type;
    using RegB = typename std::conditional<B16, x8, f32x4>::type;
    unsigned offa[NLA], offb[NLB], offa0 = 0, offb0 = 0;                // byte offsets from the tile base; *0: row 0 (always valid)
    int rowa[NLA], rowb[NLB];
    bool cola_ok = false, colb_ok = false;                              // this thread's column chunk is inside the matrix
    // (a fp32 operand's second slot group q = 1 covers the same columns: (tid + 256) & 31 == tid & 31)
#pragma unroll
    for (int p = 0; p < NLA; ++p) {
        const int kg = A16 ? (tid >> 4) : ((tid + 256 * (p >> 2)) >> 5);
        const int col = A16 ? (tid & 15) * 8 : (tid & 31) * 4;
        const int c = min(i0 + col, g.I - (A16 ? 8 : 4));               // (I % 8 == 0: a chunk is all in or all out)
        rowa[p] = kg * 4 + (p & 3);
        offa[p] = (unsigned)(rowa[p] * g.lda + c) * (A16 ? 2u : 4u);         // BYTE offsets (saddr + 32-bit voffset loads)
        if (p == 0) { offa0 = (unsigned)c * (A16 ? 2u : 4u); cola_ok = i0 + col < g.I; }
    }
#pragma unroll
    for (int p = 0; p < NLB; ++p) {
        const int kg = B16 ? (tid >> 4) : ((tid + 256 * (p >> 2)) >> 5);
        const int col = B16 ? (tid & 15) * 8 : (tid & 31) * 4;
        const int c = min(j0 + col, g.J - (B16 ? 8 : 4));
        rowb[p] = kg * 4 + (p & 3);
        offb[p] = (unsigned)(rowb[p] * g.ldb + c) * (B16 ? 2u : 4u);
        if (p == 0) { offb0 = (unsigned)c * (B16 ? 2u : 4u); colb_ok = j0 + col < g.J; }
    }
    unsigned gcol = 0;                                                  // GATHER: byte offset of this thread's column chunk inside an h1 row group
    typedef int i32x4g __attribute__((ext_vector_type(4)));
    i32x4g rt = {0, 0, 0, 0};                                           // GATHER: h1 positions of this thread's 4 rows of the NEXT tile to request
    if constexpr (GATHER) {
        const int j = min(j0 + (tid & 15) * 8, g.J - 8);
        const int tap = j / g.gC, ci = j - tap * g.gC;
        gcol = (unsigned)(((tap % 3) * g.gF1 + tap / 3) * g.gC + ci) * 2u;
        rt = *reinterpret_cast<const i32x4g*>(g.rowtab + kbeg + (tid >> 4) * 4);
    }
    const char* baseA = static_cast<const char*>(g.A) + kbeg * g.lda * (A16 ? 2 : 4);
    const char* baseB = static_cast<const char*>(g.B) + kbeg * g.ldb * (B16 ? 2 : 4);
    const int64_t stepA = (int64_t)BK * g.lda * (A16 ? 2 : 4), stepB = (int64_t)BK * g.ldb * (B16 ? 2 : 4);

    struct Set { RegA a[NLA]; RegB b[NLB]; };
    Set s0, s1;
    using Fast = std::integral_constant<bool, true>;
    using Slow = std::integral_constant<bool, false>;
    auto load_set = [&](auto fast, Set& s, int kt) {
        if constexpr (decltype(fast)::value) {                          // tile kt is known to be a full tile of this split
            const char* pa = baseA + (int64_t)kt * stepA;
            const char* pb = baseB + (int64_t)kt * stepB;
#pragma unroll
            for (int p = 0; p < NLA; ++p) s.a[p] = *reinterpret_cast<const RegA*>(pa + offa[p]);
#pragma unroll
            for (int p = 0; p < NLB; ++p) s.b[p] = *reinterpret_cast<const RegB*>(pb + offb[p]);
        } else {                                   // UNCONDITIONAL: tiles past the split re-read its last tile (never staged), rows
            kt = min(kt, nkt - 1);                 // past the split read row 0 of the tile (zeroed when staged)
            const int rmax = klen - 1 - kt * BK;
            const char* pa = baseA + (int64_t)kt * stepA;
            const char* pb = baseB + (int64_t)kt * stepB;
#pragma unroll
            for (int p = 0; p < NLA; ++p) s.a[p] = *reinterpret_cast<const RegA*>(pa + (rowa[p] <= rmax ? offa[p] : offa0));
            if constexpr (GATHER) {
                // (load_set is called for kt = 0, 1, 2, ... in order: `rt` holds tile kt's positions, the next tile's are requested now,
                //  one call ahead of the h1 loads that depend on them)
                const i32x4g cur = rt;
                rt = *reinterpret_cast<const i32x4g*>(g.rowtab + kbeg + (int64_t)min(kt + 1, nkt - 1) * BK + (tid >> 4) * 4);
                const char* hb = static_cast<const char*>(g.B);
                const unsigned rowbytes = (unsigned)g.gC * 2u;
#pragma unroll
                for (int p = 0; p < NLB; ++p)
                    s.b[p] = *reinterpret_cast<const RegB*>(hb + (size_t)((unsigned)(rowb[p] <= rmax ? cur[p] : cur[0]) * rowbytes + gcol));
            } else {
#pragma unroll
                for (int p = 0; p < NLB; ++p) s.b[p] = *reinterpret_cast<const RegB*>(pb + (rowb[p] <= rmax ? offb[p] : offb0));
            }
        }
    };
    // bias gradient: the workgroups of column tile 0 sum the dY values they stage (per thread: its 8 / 4 fixed columns)
    const bool do_colsum = g.colsum != nullptr && tj == 0;
    float csum[A16 ? 8 : 4];
#pragma unroll
    for (int e = 0; e < (A16 ? 8 : 4); ++e) csum[e] = 0.f;

    auto store_set = [&](auto fast_c, const Set& s, int kt, int stage) {
        constexpr bool fast = decltype(fast_c)::value;                  // full tile inside the matrix: nothing to zero
        const int rmax = klen - 1 - kt * BK;
        T16* As = lds + stage * 2 * STAGE;
        T16* Bs = As + STAGE;
#pragma unroll
        for (int p = 0; p < NLA; ++p) {
            const bool keep = fast || (rowa[p] <= rmax && cola_ok);
            if constexpr (A16) {
                x8 v = s.a[p];
                if constexpr (!fast) v = __builtin_bit_cast(x8, keep ? __builtin_bit_cast(i32x4, v) : i32x4{0, 0, 0, 0});
                *reinterpret_cast<x8*>(As + rowa[p] * RS + (tid & 15) * 8) = v;
                if (do_colsum)
#pragma unroll
                    for (int e = 0; e < 8; ++e) csum[e] += (float)v[e];
            } else {
                const f32x4 v = keep ? s.a[p] : f32x4{0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<x4*>(As + rowa[p] * RS + (tid & 31) * 4) = Lowp<T16>::cvt4(v);
                if (do_colsum) { csum[0] += v.x; csum[1] += v.y; csum[2] += v.z; csum[3] += v.w; }
            }
        }
#pragma unroll
        for (int p = 0; p < NLB; ++p) {
            const bool keep = fast || (rowb[p] <= rmax && colb_ok);
            if constexpr (B16) {
                x8 v = s.b[p];
                if constexpr (!fast) v = __builtin_bit_cast(x8, keep ? __builtin_bit_cast(i32x4, v) : i32x4{0, 0, 0, 0});
                *reinterpret_cast<x8*>(Bs + rowb[p] * RS + (tid & 15) * 8) = v;
            } else {
                const f32x4 v = keep ? s.b[p] : f32x4{0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<x4*>(Bs + rowb[p] * RS + (tid & 31) * 4) = Lowp<T16>::cvt4(v);
            }
        }
    };

    // transposing fragment read of a [k][index] image (see gemm_bwd_mfma16_impl.h)
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    auto tr_frag = [&](const T16* S, int idx0, int s) -> x8 {
        const int q = (lane >> 2) & 3, p = lane & 3, grp = (lane >> 4) & 1;
        const T16* a0 = S + (16 * s + 8 * hf + q) * RS + idx0 + 16 * grp + 4 * p;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(const_cast<T16*>(a0)));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(const_cast<T16*>(a0 + 4 * RS)));
        union { struct { s16x4 l, h; } p2; x8 v; } u;
        u.p2.l = lo; u.p2.h = hi;
        return u.v;
    };
    auto lds_barrier = [&]() {                                          // LDS only: the register prefetch stays in flight
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto multiply = [&](int stage, auto&& after_step0, auto&& after_step1) {
        const T16* As = lds + stage * 2 * STAGE;
        const T16* Bs = As + STAGE;
        x8 fa[2][2], fb[2][2];
        auto read_frags = [&](int s) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                fa[s & 1][t] = tr_frag(As, wr * 64 + 32 * t, s);
                fb[s & 1][t] = tr_frag(Bs, wc * 64 + 32 * t, s);
            }
        };
        read_frags(0);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s < 3) read_frags(s + 1);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = Lowp<T16>::mfma(fa[s & 1][mt], fb[s & 1][nt], acc[mt][nt]);
            if (s == 0) after_step0();
            if (s == 1) after_step1();
        }
    };

    const bool tracer = g.trace && tid == 0 && blockIdx.y == 0 && (blockIdx.x == 0 || blockIdx.x == nwg / 2);
    unsigned long long* tr = g.trace + (blockIdx.x == 0 ? 0 : 64);
    if (tracer) tr[0] = __builtin_amdgcn_s_memrealtime();
    // ---- pipeline: on entry to step kt, LDS stage (kt & 1) holds tile kt, register set `nxt` holds tile kt + 1 and the other
    //      set tile kt + 2 (both requested earlier); the step multiplies tile kt, stages `nxt` and refills it with tile kt + 3.
    load_set(Slow{}, s0, 0);
    store_set(Slow{}, s0, 0, 0);
    load_set(Slow{}, s1, 1);
    load_set(Slow{}, s0, 2);
    lds_barrier();
    if (tracer) tr[1] = __builtin_amdgcn_s_memrealtime();
    auto step = [&](auto fast, int kt, Set& nxt) {
        // the staging of tile kt+1 (its LDS stage has been free since the last barrier) and the refill request ride behind the
        // first two MFMA steps instead of following the last one, when every wave of the CU did them with the matrix pipe idle
        multiply(kt & 1,
                 [&]() { if (decltype(fast)::value || kt + 1 < nkt) store_set(fast, nxt, kt + 1, (kt + 1) & 1); },
                 [&]() { load_set(fast, nxt, kt + 3); });
        lds_barrier();
        if (tracer && kt < 58) tr[2 + kt] = __builtin_amdgcn_s_memrealtime();
    };
    // FAST steps: the tile staged (kt + 1) and the tile requested (kt + 3) are full tiles inside the matrix: no selects, no
    // zeroing, no clamps.  (The first version spent ~2000 of its ~2500 cycles per K-tile on that bookkeeping around 16 MFMAs.)
    // (only the all-16-bit instantiation: with a fp32 operand the second copy of the loop costs registers the sets need)
    const int nfast = (A16 && B16 && !GATHER && i0 + BM <= g.I && j0 + BN <= g.J) ? (klen / BK - 3) & ~1 : 0;       // even: the sets keep their roles
    int kt = 0;
    if constexpr (A16 && B16 && !GATHER) {
        for (; kt < nfast; kt += 2) {
            step(Fast{}, kt, s1);
            step(Fast{}, kt + 1, s0);
        }
    }
    for (; kt + 1 < nkt; kt += 2) {
        step(Slow{}, kt, s1);
        step(Slow{}, kt + 1, s0);
    }
    if (kt < nkt) step(Slow{}, kt, s1);

    // ---- epilogue: natural MFMA orientation (lane li = column, register r = row (r&3) + 8 (r>>2) + 4 hf), atomics
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int col = j0 + wc * 64 + nt * 32 + li;
            if (col >= g.J) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i0 + wr * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf;
                if (row < g.I) atomicAdd(g.C + (int64_t)row * g.ldc + col, g.alpha * acc[mt][nt][r]);
            }
        }
    if (tracer) { tr[60] = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_s_waitcnt(0); tr[61] = __builtin_amdgcn_s_memrealtime(); }
    if (do_colsum) {
        // threads with equal column chunk: 16-bit: tid & 15 (lanes l, l+16, l+32, l+48); fp32: tid & 31 (lanes l, l+32)
#pragma unroll
        for (int e = 0; e < (A16 ? 8 : 4); ++e) {
            float v = csum[e];
            v += __shfl_xor(v, 32);
            if (A16) v += __shfl_xor(v, 16);
            csum[e] = v;
        }
        const int w = A16 ? 8 : 4, nl = A16 ? 16 : 32;
        if (lane < nl && cola_ok) {
            const int c = i0 + (A16 ? (tid & 15) * 8 : (tid & 31) * 4);
#pragma unroll
            for (int e = 0; e < w; ++e) atomicAdd(g.colsum + c + e, g.alpha * csum[e]);
        }
    }
}

unsigned long long* g_dw16_trace = nullptr;

template <typename T16>
int launch_dw16(DwArgs g, int a16, int b16, hipStream_t s) {
    g.tiles_i = (unsigned)((g.I + 127) / 128);
    g.tiles_j = (unsigned)((g.J + 127) / 128);
    const unsigned tiles = g.tiles_i * g.tiles_j;
    int splits = 1;
    while ((int64_t)tiles * splits < 448 && g.Kc / (splits * 2) >= 256 && splits < 64) splits *= 2;   // ~2 workgroups per CU
    static const int forced = [] { const char* e = getenv("CFM_DW16_SPLITS"); return e ? atoi(e) : 0; }();   // tuning only
    if (forced > 0) splits = forced;
    const int64_t per = (g.Kc + splits - 1) / splits;
    g.k_per_split = (per + 63) / 64 * 64;
    const int64_t span_a = (g.k_per_split + 64) * g.lda * (a16 ? 2 : 4), span_b = g.rowtab ? 0 : (g.k_per_split + 64) * g.ldb * (b16 ? 2 : 4);
    if (span_a >= (int64_t)1 << 31 || span_b >= (int64_t)1 << 31) return CFM_ERR_UNSUPPORTED;
    const dim3 grid(tiles, (unsigned)((g.Kc + g.k_per_split - 1) / g.k_per_split));
#define DW(A_, B_) hipLaunchKernelGGL((gemm_dw16_kernel<T16, A_, B_>), grid, dim3(256), 0, s, g)
    if (g.rowtab) {
        if (!b16) return CFM_ERR_UNSUPPORTED;
        if (a16) hipLaunchKernelGGL((gemm_dw16_kernel<T16, true, true, true>), grid, dim3(256), 0, s, g);
        else hipLaunchKernelGGL((gemm_dw16_kernel<T16, false, true, true>), grid, dim3(256), 0, s, g);
    } else if (a16 && b16) DW(true, true);
    else if (a16) DW(true, false);
    else if (b16) DW(false, true);
    else DW(false, false);
#undef DW
    return cfm_launch_status();
}

}  // namespace

#define CFM_CAT2(a, b) a##b
#define CFM_CAT(a, b) CFM_CAT2(a, b)

void CFM_CAT(cfm_dw16_trace_, CFM_T16_FN)(void* p) { g_dw16_trace = static_cast<unsigned long long*>(p); }

int CFM_CAT(cfm_dw16_, CFM_T16_FN)(const void* dy, int dy16, int64_t ldy, const void* x, int x16, int64_t ldx, float* dw, int64_t ldw,
                                   float* db, int N, int K, int64_t M, float alpha, hipStream_t s) {
    DwArgs g{};
    g.trace = g_dw16_trace;
    g.A = dy; g.B = x; g.C = dw; g.colsum = db; g.lda = ldy; g.ldb = ldx; g.ldc = ldw; g.I = N; g.J = K; g.Kc = M; g.alpha = alpha;
    return launch_dw16<CFM_T16>(g, dy16, x16, s);
}

// conv2 weight gradient of the stem: dw2p (C, 9C) += dz2^T (M x C, fp32) . im2col(h1) (M x 9C gathered from the 16-bit h1 through
// rowtab: see DwArgs)
int CFM_CAT(cfm_dw16_conv2_, CFM_T16_FN)(const void* dz2, int dz16, const void* h1_16, const int* rowtab, float* dw2p, float* db2, int C,
                                         int F1, int64_t M, hipStream_t s) {
    DwArgs g{};
    g.trace = g_dw16_trace;
    g.A = dz2; g.B = h1_16; g.C = dw2p; g.colsum = db2; g.lda = C; g.ldb = 0; g.ldc = 9 * C; g.I = C; g.J = 9 * C; g.Kc = M;
    g.alpha = 1.f; g.rowtab = rowtab; g.gC = C; g.gF1 = F1;
    return launch_dw16<CFM_T16>(g, dz16, 1, s);
}
