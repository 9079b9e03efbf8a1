// Relative-position attention forward on the 16-bit matrix pipe (bf16 / fp16): the torch.autocast arithmetic of
// attention.py:47-72 -- Q, K, V and the projected positions are rounded to the 16-bit type for the three products
// (fp32 accumulation), the softmax runs in fp32, the probabilities are rounded for P.V.  All tensors in HBM stay fp32.
//
// Same algorithm and orientation as attention_f32.hip (flash-style, queries on lanes, relative shift = per-lane column
// skew through a per-wave LDS tile, K / V / positional ring staged in LDS and shared by 4 waves); what changes:
//   * v_mfma_f32_32x32x16_{bf16,f16}: 4 steps per 64-dim product instead of 32 -> 16 MFMAs per key tile instead of 128,
//     so the kernel is bound by the staging / skew / softmax instruction stream, not by the matrix pipe;
//   * K and the ring are staged as 16-bit rows of 144 B (conflict-free 16-byte fragment reads: 8 dims per lane per step);
//   * the P.V product contracts over keys, and a lane's accumulator registers hold keys {16s + 8(e>>2) + 4hf + (e&3)},
//     e = 0..7, for step s.  V is staged as loaded ([key][dim], 16-bit) and its V^T fragment is fetched with two
//     ds_read_b64_tr_b16 (gfx950's transposing LDS read) over key rows 16s+4hf+{0..3} and +8 -- exactly that key order,
//     so P goes from the softmax registers to the MFMA without any shuffle and V is never transposed by hand;
//   * (Q+u), (Q+v) live in 16 VGPRs each (64 in the fp32 kernel);
//   * round 2, as in the fp32 kernel: one band product per key tile (the skewed values of the other band tile are carried in
//     registers) and double-buffered K / V tiles: ONE workgroup barrier per key tile instead of two (60 KB LDS, 2 workgroups per CU).
#include "cfm_common.h"
#include <math.h>

namespace {

constexpr int KROWH = 72;                // LDS row (16-bit elements) of the K tile and the P ring: 64 + 8 pad = 144 B
constexpr int VROWH = 96;                // LDS row of the V tile [key][dim]: 64 dims + 32 pad = 192 B (tr-read conflict-free)
constexpr int RINGH = 160;

struct Attn16Args {
    const float* q; const float* k; const float* v; int64_t ld;
    const float* pos; int64_t ldp; const float* u; const float* vb;
    const int64_t* lengths; float* ctx; int64_t ldo; float* lse;
    int ctx16;                   // ctx is stored in the kernel's 16-bit type (inference: its only consumer is the out-projection GEMM)
    int qkv16;                   // q / k / v are stored in that type (ld in elements): what torch.autocast hands the attention core
    int B, T, H, dh; float inv_sqrt_dh;
    float drop_p; unsigned long long drop_seed;
    int q_begin, q_end;          // query rows computed: [q_begin, q_end) (the whole utterance, or a streaming chunk against its K/V cache)
};

// QKV16: q / k / v are stored in T16 (compile-time: a run-time test around the loads costs their overlap -- the compiler waits
// for every conditional load at once)
template <typename T16, bool QKV16>
__global__ __launch_bounds__(256, 2) void relpos_attn_fwd_mfma16_kernel(const Attn16Args a) {
    using x8 = typename Lowp<T16>::x8;
    using x4 = typename Lowp<T16>::x4;
    __shared__ __attribute__((aligned(16))) T16 smem16[2 * 32 * KROWH + 2 * 32 * VROWH + RINGH * KROWH];
    __shared__ __attribute__((aligned(16))) float gsm[4 * 32 * 32];
    T16* Ksb = smem16;                           // [2][32 keys][KROWH]     double-buffered: one workgroup barrier per key tile
    T16* Vtb = Ksb + 2 * 32 * KROWH;             // [2][32 keys][VROWH]: stored as loaded, read transposed (ds_read_b64_tr_b16)
    T16* Pr = Vtb + 2 * 32 * VROWH;              // [RINGH][KROWH]
    float* gs = gsm + (threadIdx.x >> 6) * 1024; // per-wave skew tile [32][32]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, hf = lane >> 5;
    const int bh = blockIdx.y, b = bh / a.H, h = bh % a.H;
    const int T = a.T, dh = a.dh;
    const int q0 = a.q_begin + blockIdx.x * 128;
    const int i0 = q0 + wave * 32;
    const bool active = i0 < a.q_end;

    int klen = T;
    bool uniform = false;
    if (a.lengths) {
        const int64_t L = a.lengths[b];
        if (L <= 0) uniform = true;
        else if (L < T) klen = (int)L;
    }
    const int ntiles = (klen + 31) / 32;

    const int64_t qkv_off = (int64_t)b * T * a.ld + h * dh;          // element offset of this (utterance, head) in q / k / v
    auto load4 = [&](const float* base, int64_t off) -> f32x4 {      // four consecutive q / k / v values at element offset `off`
        if constexpr (QKV16) {
            const x4 t = *reinterpret_cast<const x4*>(reinterpret_cast<const T16*>(base) + off);
            return f32x4{(float)t[0], (float)t[1], (float)t[2], (float)t[3]};
        }
        else return *reinterpret_cast<const f32x4*>(base + off);
    };
    const float* pbase = a.pos + h * dh;
    const int jmax = 2 * T - 2;
    const int ring_bias = RINGH * ((T + 128 + q0) / RINGH + 2);

    // ---- cooperative staging: thread -> (row srow + 16*pass, 4-dim chunk sch) of a 32-row x 64-dim fp32 tile
    const int srow = tid >> 4, sch = tid & 15;
    const bool sok = sch * 4 < dh;
    f32x4 pk[2], pv[2], pp[2];
    x4 pk16[2], pv16[2];                                           // (QKV16: the staged K / V chunks, raw)
    auto prefetch = [&](int kt) {
        const int k0 = kt * 32;
        const int jnew = T - 1 - q0 + k0;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int r = srow + 16 * p;
            const int key = min(k0 + r, T - 1);
            const int j = max(0, min(jnew + r, jmax));
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            if constexpr (QKV16) {                                 // raw 16-bit values: nothing may touch them before commit(),
                x4 z16;                                              // or the prefetch stops overlapping the tile's products
#pragma unroll
                for (int e = 0; e < 4; ++e) z16[e] = (T16)0.f;
                const int64_t off = qkv_off + (int64_t)key * a.ld + sch * 4;
                pk16[p] = sok ? *reinterpret_cast<const x4*>(reinterpret_cast<const T16*>(a.k) + off) : z16;
                pv16[p] = sok ? *reinterpret_cast<const x4*>(reinterpret_cast<const T16*>(a.v) + off) : z16;
            } else {
                pk[p] = sok ? load4(a.k, qkv_off + (int64_t)key * a.ld + sch * 4) : z;
                pv[p] = sok ? load4(a.v, qkv_off + (int64_t)key * a.ld + sch * 4) : z;
            }
            pp[p] = sok ? *reinterpret_cast<const f32x4*>(pbase + (int64_t)j * a.ldp + sch * 4) : z;
        }
    };
    auto commit = [&](int kt) {                                    // tile kt -> buffer (kt & 1); its 32 new ring rows replace rows
        const int k0 = kt * 32;                                    // that only band tile 1 of the PREVIOUS key tile could still need
        const int jnew = T - 1 - q0 + k0;
        T16* Ks = Ksb + (kt & 1) * 32 * KROWH;
        T16* Vt = Vtb + (kt & 1) * 32 * VROWH;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int r = srow + 16 * p;
            *reinterpret_cast<x4*>(Ks + r * KROWH + sch * 4) = QKV16 ? pk16[p] : Lowp<T16>::cvt4(pk[p]);
            *reinterpret_cast<x4*>(Vt + r * VROWH + sch * 4) = QKV16 ? pv16[p] : Lowp<T16>::cvt4(pv[p]);
            const int slot = (jnew + r + ring_bias) % RINGH;
            *reinterpret_cast<x4*>(Pr + slot * KROWH + sch * 4) = Lowp<T16>::cvt4(pp[p]);
        }
    };

    {   // prologue: ring rows [jlo, jlo+127] of tile 0 (the top 32 rows arrive with prefetch(0))
        const int jlo = T - 1 - q0 - 128;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int r = srow + 16 * p;
            const int j = max(0, min(jlo + r, jmax));
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            const f32x4 val = sok ? *reinterpret_cast<const f32x4*>(pbase + (int64_t)j * a.ldp + sch * 4) : z;
            const int slot = (jlo + r + ring_bias) % RINGH;
            *reinterpret_cast<x4*>(Pr + slot * KROWH + sch * 4) = Lowp<T16>::cvt4(val);
        }
    }
    prefetch(0);
    commit(0);
    if (ntiles > 1) prefetch(1);
    __syncthreads();

    // ---- (Q+u)^T, (Q+v)^T as MFMA B operands: lane (query li, half hf) holds dims 16s + 8hf + {0..7} at step s
    x8 qu[4], qv[4];
    {
        const int qi = min(i0 + li, T - 1);
        const int64_t qoff = ((int64_t)b * T + qi) * a.ld + h * dh;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int dd = 16 * s + 8 * hf + 4 * half;
                f32x4 x = {0.f, 0.f, 0.f, 0.f}, uu = x, vv = x;
                if (dd < dh) {
                    x = load4(a.q, qoff + dd);
                    uu = *reinterpret_cast<const f32x4*>(a.u + h * dh + dd);
                    vv = *reinterpret_cast<const f32x4*>(a.vb + h * dh + dd);
                }
                const x4 cu = Lowp<T16>::cvt4(x + uu), cv = Lowp<T16>::cvt4(x + vv);
#pragma unroll
                for (int e = 0; e < 4; ++e) { qu[s][4 * half + e] = cu[e]; qv[s][4 * half + e] = cv[e]; }
            }
    }

    f32x16 o[2];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[n][r] = 0.f;
    float mrow = -INFINITY, lrow = 0.f;                              // running maximum in the log2 domain, row sum
    const float scale2 = a.inv_sqrt_dh * 1.44269504088896340736f;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // Positional band: one 32-row band tile per key tile (band tile 1 of key tile kt+1 = band tile 0 of key tile kt, and the
    // 16 unconditional skew reads of band tile 0 already fetch what the next key tile needs from its band tile 1: carried in
    // `skp`, see attention_f32.hip).
    auto band = [&](int jbase, int mt) {
        const int slot = (jbase - (32 * mt + li) + ring_bias) % RINGH;
        const T16* prow = Pr + slot * KROWH + 8 * hf;
        f32x16 ga = zero16;
#pragma unroll
        for (int s = 0; s < 4; ++s) ga = Lowp<T16>::mfma(*reinterpret_cast<const x8*>(prow + 16 * s), qv[s], ga);
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
            const int jj = li - ((r & 3) + 8 * (r >> 2) + 4 * hf) + 31;
            dst[r] = gs[(jj & 31) * 32 + li];
        }
    };
    float skp[16];
    if (active) {                                                   // first key tile: its band tile 1 explicitly
        spill_band(band(T - 1 - i0 + 31, 1));
        skew_reads(skp);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();                                                // every wave has read its band-1 rows: commit(1) may replace them

    for (int kt = 0; kt < ntiles; ++kt) {
        const int k0 = kt * 32;
        const T16* Ks = Ksb + (kt & 1) * 32 * KROWH;
        const T16* Vt = Vtb + (kt & 1) * 32 * VROWH;
        // stage the next key tile into the other buffer (its readers passed the barrier that ended tile kt-1), request the one after
        if (kt + 1 < ntiles) {
            commit(kt + 1);
            if (kt + 2 < ntiles) prefetch(kt + 2);
        }
        if (active) {
            const int jbase = T - 1 - i0 + k0 + 31;
            spill_band(band(jbase, 0));
            f32x16 sc = zero16;
            {
                const T16* krow = Ks + li * KROWH + 8 * hf;
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    sc = Lowp<T16>::mfma(*reinterpret_cast<const x8*>(krow + 16 * s), qu[s], sc);
            }
            float skn[16], sk[16];
            skew_reads(skn);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int jj = li - ((r & 3) + 8 * (r >> 2) + 4 * hf) + 31;
                sk[r] = (jj >> 5) ? skp[r] : skn[r];
                skp[r] = skn[r];
            }
            // ---- scale + mask, online softmax
            // (log2 domain: one multiply by inv_sqrt_dh * log2 e, v_exp_f32 directly; the key-padding selects only in the tile that holds
            //  the end of the utterance: this section, not the matrix pipe, bounds the kernel)
            float p[16];
            float tmax = -INFINITY;
            if (uniform || k0 + 32 > klen) {                    // (wave-uniform)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kk = (r & 3) + 8 * (r >> 2) + 4 * hf;
                    float s = (sc[r] + sk[r]) * scale2;
                    if (uniform) s = 0.f;
                    if (k0 + kk >= klen) s = -INFINITY;
                    p[r] = s;
                    tmax = fmaxf(tmax, s);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) { p[r] = (sc[r] + sk[r]) * scale2; tmax = fmaxf(tmax, p[r]); }
            }
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            const float mnew = fmaxf(mrow, tmax);
            const float alpha = __builtin_amdgcn_exp2f(mrow - mnew);
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { p[r] = __builtin_amdgcn_exp2f(p[r] - mnew); psum += p[r]; }
            psum += __shfl_xor(psum, 32, 64);
            lrow = lrow * alpha + psum;
            mrow = mnew;
            if (a.drop_p > 0.f) {
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
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[n][r] *= alpha;
            // ---- O^T += V^T . P^T : step s contracts the 16 keys held in registers 8s..8s+7 of both lane halves
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                x8 pb;
#pragma unroll
                for (int e = 0; e < 8; ++e) pb[e] = (T16)p[8 * s + e];
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    // V^T fragment by the transposing LDS read: 16-lane group -> dims 32n + 16*(group&1) .. +15, key rows
                    // 16s + 4hf + {0..3} and + 8: lane 4q+p supplies row q, dims 4p..4p+3; each lane gets its own dim
                    typedef short s16x4 __attribute__((ext_vector_type(4)));
                    const int tq = (lane >> 2) & 3, tp = lane & 3, grp = (lane >> 4) & 1;
                    const T16* a0 = Vt + (16 * s + 4 * hf + tq) * VROWH + 32 * n + 16 * grp + 4 * tp;
                    union { struct { s16x4 l, h; } p2; x8 v; } u;
                    u.p2.l = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(const_cast<T16*>(a0)));
                    u.p2.h = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(const_cast<T16*>(a0 + 8 * VROWH)));
                    o[n] = Lowp<T16>::mfma(u.v, pb, o[n]);
                }
            }
        }
        if (kt + 1 < ntiles) __syncthreads();                       // one barrier per key tile
    }

    if (active && i0 + li < a.q_end) {
        const float inv = 1.0f / lrow;
        float* orow = a.ctx + ((int64_t)b * T + i0 + li) * a.ldo + h * dh;
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int dd = 32 * n + 8 * gq + 4 * hf;
                if (dd < dh) {
                    f32x4 out = {o[n][4 * gq] * inv, o[n][4 * gq + 1] * inv, o[n][4 * gq + 2] * inv, o[n][4 * gq + 3] * inv};
                    if (a.ctx16)
                        *reinterpret_cast<typename Lowp<T16>::x4*>(reinterpret_cast<T16*>(a.ctx) + ((int64_t)b * T + i0 + li) * a.ldo + h * dh + dd) =
                            Lowp<T16>::cvt4(out);
                    else *reinterpret_cast<f32x4*>(orow + dd) = out;
                }
            }
        if (a.lse && hf == 0) a.lse[((int64_t)b * a.H + h) * T + i0 + li] = mrow * 0.69314718055994530942f + logf(lrow);
    }
}

}  // namespace

// 16-bit-MFMA form of cfm_relpos_attention_train_f32 (prec = CFM_PREC_BF16 | CFM_PREC_FP16; lse_or_null; drop_p may be 0).
static int attn16_launch(int prec, const float* q, const float* k, const float* v, int64_t ld, const float* pos, int64_t ldp,
                         const float* u, const float* vbias, const int64_t* lengths_or_null, void* ctxv, int ctx16, int64_t ldo,
                         float* lse_or_null, int B, int T, int H, int dh, float drop_p, uint64_t drop_seed, cfm_stream_t stream,
                         int qkv16 = 0, int q_begin = 0, int q_count = -1) {
    if (q_count < 0) q_count = T - q_begin;
    CFM_REQUIRE(q_begin >= 0 && q_count >= 0 && q_begin + q_count <= T, CFM_ERR_BAD_SHAPE);
    if (q_count == 0) return CFM_OK;
    float* ctx = static_cast<float*>(ctxv);
    CFM_REQUIRE(q && k && v && pos && u && vbias && ctx, CFM_ERR_NULL);
    CFM_REQUIRE(B > 0 && T > 0 && H > 0 && dh > 0 && (dh & 3) == 0, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE((ld & 3) == 0 && (ldp & 3) == 0 && (ldo & 3) == 0 && drop_p >= 0.f && drop_p < 1.f, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(dh <= 64, CFM_ERR_UNSUPPORTED);
    CFM_REQUIRE(CFM_ALIGNED16(q) && CFM_ALIGNED16(k) && CFM_ALIGNED16(v) && CFM_ALIGNED16(pos) && CFM_ALIGNED16(u) &&
                CFM_ALIGNED16(vbias) && CFM_ALIGNED16(ctx), CFM_ERR_ALIGN);
    CFM_REQUIRE((int64_t)B * H <= 65535 && T < (1 << 28), CFM_ERR_UNSUPPORTED);
    const Attn16Args a{q, k, v, ld, pos, ldp, u, vbias, lengths_or_null, ctx, ldo, lse_or_null, ctx16, qkv16, B, T, H, dh,
                       1.0f / sqrtf((float)dh), drop_p, drop_seed, q_begin, q_begin + q_count};
    const dim3 grid((unsigned)((q_count + 127) / 128), (unsigned)(B * H)), block(256);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (prec == CFM_PREC_BF16) {
        if (qkv16) hipLaunchKernelGGL((relpos_attn_fwd_mfma16_kernel<__bf16, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((relpos_attn_fwd_mfma16_kernel<__bf16, false>), grid, block, 0, s, a);
    } else if (prec == CFM_PREC_FP16) {
        if (qkv16) hipLaunchKernelGGL((relpos_attn_fwd_mfma16_kernel<_Float16, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((relpos_attn_fwd_mfma16_kernel<_Float16, false>), grid, block, 0, s, a);
    } else return CFM_ERR_UNSUPPORTED;
    return cfm_launch_status();
}

extern "C" int cfm_relpos_attention_mfma16_f32(int prec, const float* q, const float* k, const float* v, int64_t ld,
                                               const float* pos, int64_t ldp, const float* u, const float* vbias,
                                               const int64_t* lengths_or_null, float* ctx, int64_t ldo, float* lse_or_null,
                                               int B, int T, int H, int dh, float drop_p, uint64_t drop_seed,
                                               cfm_stream_t stream) {
    return attn16_launch(prec, q, k, v, ld, pos, ldp, u, vbias, lengths_or_null, ctx, 0, ldo, lse_or_null, B, T, H, dh, drop_p,
                         drop_seed, stream);
}
// Incremental attention over a K/V cache under autocast (streaming: BASELINE cfg-5): the 16-bit form of
// cfm_relpos_attention_rows_f32 -- only query rows [q_begin, q_begin + q_count) are computed, against keys < lengths[b]; q / k / v
// are the fp32 cache (rounded to `prec` where they enter a product, as torch.autocast's attention core does), ctx fp32.  No key
// split: a workgroup walks the whole cache.
extern "C" int cfm_relpos_attention_rows_mfma16_f32(int prec, const float* q, const float* k, const float* v, int64_t ld,
                                                    const float* pos, int64_t ldp, const float* u, const float* vbias,
                                                    const int64_t* lengths, float* ctx, int64_t ldo, int B, int T, int H, int dh,
                                                    int q_begin, int q_count, cfm_stream_t stream) {
    CFM_REQUIRE(lengths != nullptr, CFM_ERR_NULL);
    return attn16_launch(prec, q, k, v, ld, pos, ldp, u, vbias, lengths, ctx, 0, ldo, nullptr, B, T, H, dh, 0.f, 0, stream, 0, q_begin,
                         q_count);
}
// Inference under autocast with 16-bit tensors either side of the core: qkv_is_16bit -- q / k / v stored in `prec` (ld in elements;
// what torch.autocast's projections hand the attention: the reference rounds them there too); ctx_is_16bit -- the context stored in
// `prec` (ldo in elements): its only consumer is the out-projection GEMM, which rounds an fp32 context to that type anyway.
extern "C" int cfm_relpos_attention_io16_mfma16_f32(int prec, const void* q, const void* k, const void* v, int qkv_is_16bit, int64_t ld,
                                                    const float* pos, int64_t ldp, const float* u, const float* vbias,
                                                    const int64_t* lengths_or_null, void* ctx, int ctx_is_16bit, int64_t ldo, int B,
                                                    int T, int H, int dh, cfm_stream_t stream) {
    CFM_REQUIRE(!qkv_is_16bit || (ld & 7) == 0, CFM_ERR_BAD_SHAPE);
    return attn16_launch(prec, static_cast<const float*>(q), static_cast<const float*>(k), static_cast<const float*>(v), ld, pos, ldp, u,
                         vbias, lengths_or_null, ctx, ctx_is_16bit ? 1 : 0, ldo, nullptr, B, T, H, dh, 0.f, 0, stream,
                         qkv_is_16bit ? 1 : 0);
}
