// 16-bit-MFMA form of the backward GEMM (gemm_bwd_f32.hip):  C[i][j] (+)= alpha * sum_k r16(A(i,k)) * r16(B(j,k)) [* swish'(Z)]
//
// Under torch.autocast the reference's backward Linear/Conv products run on 16-bit inputs with fp32 accumulation; this is
// that arithmetic with every tensor still fp32 in HBM: operands are rounded (RNE, bf16 or fp16) on their way into LDS.
// v_mfma_f32_32x32x16_{bf16,f16} is 16x the fp32 MFMA rate, so the kernel is bound by the fp32 operand stream; the
// staging is built for that: contraction tile 64, LDS image [index][64 k] in the 16-bit type (128-byte rows, no padding,
// 16-byte blocks XOR-swizzled by (index>>2)&7 -- conflict-free for both the 8-byte staging writes and the 16-byte
// fragment reads), double buffered, one barrier per tile.
//   ROW operand (X[idx*ld + k]): 16 lanes load the 256 B of one tile row, convert, one 8-byte LDS write each.
//   COL operand (X[k*ld + idx]): a thread loads a 4(k) x 4(idx) fp32 block (four coalesced 16-byte loads along idx),
//     transposes it in registers and writes four 8-byte k-runs -- the transposition never touches HBM.
// GATHER 1/2 (stem convolution backward) and split-K with natural-orientation atomics as in the fp32 kernel.
#include "gemm_bwd_args.h"

// per-type halves (gemm_bwd_mfma16_{bf16,f16}.hip)
int cfm_bwd16_gemm_bf16(BwdArgs& g, int a_col, int b_col, hipStream_t s);
int cfm_bwd16_gemm_f16(BwdArgs& g, int a_col, int b_col, hipStream_t s);
int cfm_bwd16_conv2_weight_bf16(BwdArgs g, hipStream_t s);
int cfm_bwd16_conv2_weight_f16(BwdArgs g, hipStream_t s);
int cfm_bwd16_conv2_input_bf16(const float* dz2, const float* w2c, float* dh1, int B, int F1, int T1, int C, hipStream_t s);
int cfm_bwd16_conv2_input_f16(const float* dz2, const float* w2c, float* dh1, int B, int F1, int T1, int C, hipStream_t s);
// gemm_dw16_{bf16,f16}.hip
int cfm_dw16_conv2_bf16(const void* dz2, int dz16, const void* h1_16, const int* rowtab, float* dw2p, float* db2, int C, int F1, int64_t M,
                        hipStream_t s);
int cfm_dw16_conv2_f16(const void* dz2, int dz16, const void* h1_16, const int* rowtab, float* dw2p, float* db2, int C, int F1, int64_t M,
                       hipStream_t s);
void cfm_dw16_trace_bf16(void* p);
void cfm_dw16_trace_f16(void* p);
int cfm_dw16_bf16(const void* dy, int dy16, int64_t ldy, const void* x, int x16, int64_t ldx, float* dw, int64_t ldw, float* db, int N,
                  int K, int64_t M, float alpha, hipStream_t s);
int cfm_dw16_f16(const void* dy, int dy16, int64_t ldy, const void* x, int x16, int64_t ldx, float* dw, int64_t ldw, float* db, int N,
                 int K, int64_t M, float alpha, hipStream_t s);


// Argument rules of cfm_gemm_bwd_batched_f32; prec = CFM_PREC_BF16 | CFM_PREC_FP16.
extern "C" int cfm_gemm_bwd_batched_mfma16_f32(int prec, const float* A, int a_col, int64_t lda, const void* B, int b_col,
                                               int b_is_16bit, int64_t ldb, const void* Z_or_null, int z_is_16bit, int64_t ldz, float alpha, void* C,
                                               int64_t ldc, int c_is_16bit, int I, int J, int64_t Kc, int allow_split, int accumulate,
                                               int nbatch, int nb1, int64_t sa0, int64_t sa1, int64_t sb0, int64_t sb1,
                                               int64_t sc0, int64_t sc1, float drop_p, uint64_t drop_seed,
                                               int operands_zero_padded4, cfm_stream_t stream) {
    CFM_REQUIRE(A && B && C, CFM_ERR_NULL);
    CFM_REQUIRE(I > 0 && J > 0 && Kc > 0 && nbatch > 0 && nb1 > 0 && nbatch % nb1 == 0 && nbatch <= 65535, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE((ldc & 3) == 0 && (lda & 3) == 0 && (ldb & 3) == 0, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(((sa0 | sa1 | sb0 | sb1 | sc0 | sc1) & 3) == 0, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(!b_is_16bit || (((ldb | sb0 | sb1) & 7) == 0 && b_col), CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(CFM_ALIGNED16(A) && CFM_ALIGNED16(B) && CFM_ALIGNED16(C), CFM_ERR_ALIGN);
    CFM_REQUIRE(!Z_or_null || (CFM_ALIGNED16(Z_or_null) && (ldz & 3) == 0), CFM_ERR_ALIGN);
    CFM_REQUIRE(!Z_or_null || (!a_col && b_col && nbatch == 1), CFM_ERR_UNSUPPORTED);
    // a 16-bit Z is read by the row-major (vectorised) epilogue only
    CFM_REQUIRE(!z_is_16bit || (Z_or_null && (J & 7) == 0 && (ldz & 7) == 0 && (ldc & 3) == 0), CFM_ERR_UNSUPPORTED);
    // a 16-bit C: only the swish'(Z) product (its result feeds GEMM operands), whole 16-byte chunks, plain stores
    CFM_REQUIRE(!c_is_16bit || (Z_or_null && !accumulate && (J & 7) == 0 && (ldc & 7) == 0), CFM_ERR_UNSUPPORTED);
    BwdArgs g{};
    g.b16 = b_is_16bit != 0;
    g.pad4 = operands_zero_padded4 != 0;
    g.c16 = c_is_16bit ? prec : 0;
    g.z16 = z_is_16bit ? prec : 0;
    g.A = A; g.B = static_cast<const float*>(B); g.Z = static_cast<const float*>(Z_or_null); g.C = static_cast<float*>(C); g.I = I; g.J = J; g.Kc = Kc;
    g.lda = lda; g.ldb = ldb; g.ldz = ldz; g.ldc = ldc; g.alpha = alpha;
    g.splits = (allow_split && !accumulate) ? 0 : 1;
    g.accumulate = accumulate; g.nbatch = nbatch; g.nb1 = nb1; g.drop_p = drop_p; g.drop_seed = drop_seed;
    g.sa0 = sa0; g.sa1 = sa1; g.sb0 = sb0; g.sb1 = sb1; g.sc0 = sc0; g.sc1 = sc1;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (prec == CFM_PREC_BF16) return cfm_bwd16_gemm_bf16(g, a_col, b_col, s);
    if (prec == CFM_PREC_FP16) return cfm_bwd16_gemm_f16(g, a_col, b_col, s);
    return CFM_ERR_UNSUPPORTED;
}

extern "C" int cfm_subsample_conv2_bwd_weight_mfma16_f32(int prec, const float* dz2, const float* h1, float* dw2p, int B,
                                                         int F1, int T1, int C, cfm_stream_t stream) {
    CFM_REQUIRE(dz2 && h1 && dw2p, CFM_ERR_NULL);
    CFM_REQUIRE(B > 0 && F1 >= 3 && T1 >= 3 && C > 0 && (C & 3) == 0, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(CFM_ALIGNED16(dz2) && CFM_ALIGNED16(h1) && CFM_ALIGNED16(dw2p), CFM_ERR_ALIGN);
    BwdArgs g{};
    g.cT1 = T1; g.cF1 = F1; g.cT2 = (T1 - 1) / 2; g.cF2 = (F1 - 1) / 2; g.cC = C;
    g.A = dz2; g.B = h1; g.C = dw2p; g.I = C; g.J = 9 * C; g.Kc = (int64_t)B * g.cT2 * g.cF2;
    g.lda = C; g.ldb = 0; g.ldc = 9 * C; g.alpha = 1.f; g.splits = 0; g.nbatch = 1; g.nb1 = 1;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (prec == CFM_PREC_BF16) return cfm_bwd16_conv2_weight_bf16(g, s);
    if (prec == CFM_PREC_FP16) return cfm_bwd16_conv2_weight_f16(g, s);
    return CFM_ERR_UNSUPPORTED;
}

extern "C" int cfm_subsample_conv2_bwd_input_mfma16_f32(int prec, const float* dz2, const float* w2c, float* dh1, int B,
                                                        int F1, int T1, int C, cfm_stream_t stream) {
    CFM_REQUIRE(dz2 && w2c && dh1, CFM_ERR_NULL);
    CFM_REQUIRE(B > 0 && F1 >= 3 && T1 >= 3 && C > 0, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE(C % 64 == 0, CFM_ERR_UNSUPPORTED);
    CFM_REQUIRE(CFM_ALIGNED16(dz2) && CFM_ALIGNED16(w2c) && CFM_ALIGNED16(dh1), CFM_ERR_ALIGN);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (prec == CFM_PREC_BF16) return cfm_bwd16_conv2_input_bf16(dz2, w2c, dh1, B, F1, T1, C, s);
    if (prec == CFM_PREC_FP16) return cfm_bwd16_conv2_input_f16(dz2, w2c, dh1, B, F1, T1, C, s);
    return CFM_ERR_UNSUPPORTED;
}

// Weight (and bias) gradient of y = x.W^T + b on the 16-bit matrix pipe (gemm_dw16_impl.h):
//   dw (N,K) += alpha * dy^T.x ;  db (N) += alpha * column sums of dy   (db_or_null = NULL: skipped).  Both ACCUMULATE with
// atomics: the caller zero-fills.  dy (M,N) and x (M,K) are row-major; *_is_16bit: stored in the 16-bit type of `prec`
// (leading dimensions in elements of the stored type).  N % 8 == 0, K % 8 == 0, 16-byte aligned rows; CFM_ERR_UNSUPPORTED
// otherwise (use cfm_gemm_bwd_batched_mfma16_f32 + cfm_colsum_f32).
extern "C" int cfm_linear_bwd_weight_mfma16_f32(int prec, const void* dy, int dy_is_16bit, int64_t ldy, const void* x,
                                                int x_is_16bit, int64_t ldx, float* dw, int64_t ldw, float* db_or_null, int N,
                                                int K, int64_t M, float alpha, cfm_stream_t stream) {
    CFM_REQUIRE(dy && x && dw, CFM_ERR_NULL);
    CFM_REQUIRE(N > 0 && K > 0 && M > 0 && ldy >= N && ldx >= K && ldw >= K, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE((N & 7) == 0 && (K & 7) == 0, CFM_ERR_UNSUPPORTED);
    CFM_REQUIRE((ldy & (dy_is_16bit ? 7 : 3)) == 0 && (ldx & (x_is_16bit ? 7 : 3)) == 0, CFM_ERR_UNSUPPORTED);
    CFM_REQUIRE(CFM_ALIGNED16(dy) && CFM_ALIGNED16(x) && CFM_ALIGNED16(dw), CFM_ERR_ALIGN);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (prec == CFM_PREC_BF16) return cfm_dw16_bf16(dy, dy_is_16bit, ldy, x, x_is_16bit, ldx, dw, ldw, db_or_null, N, K, M, alpha, s);
    if (prec == CFM_PREC_FP16) return cfm_dw16_f16(dy, dy_is_16bit, ldy, x, x_is_16bit, ldx, dw, ldw, db_or_null, N, K, M, alpha, s);
    return CFM_ERR_UNSUPPORTED;
}

// diagnostics: 2 x 64 uint64 stamps (100 MHz) of two workgroups of the next cfm_linear_bwd_weight_mfma16_f32 launches
extern "C" int cfm_debug_dw16_trace(void* trace_or_null) {
    cfm_dw16_trace_bf16(trace_or_null);
    cfm_dw16_trace_f16(trace_or_null);
    return CFM_OK;
}

namespace {
// rowtab[m] = h1 position ((b*T1 + 2*t2)*F1 + 2*f2) of output position m = (b, t2, f2); entries past M (the table is padded
// to a multiple of 64 plus 64) repeat position 0
__global__ __launch_bounds__(256) void conv2_rowtab_kernel(int* __restrict__ tab, int64_t M, int64_t n, int T2, int F2, int T1, int F1) {
    const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (m >= n) return;
    int v = 0;
    if (m < M) {
        const int f2 = (int)(m % F2);
        const int64_t bt = m / F2;
        const int t2 = (int)(bt % T2);
        const int b = (int)(bt / T2);
        v = (b * T1 + 2 * t2) * F1 + 2 * f2;
    }
    tab[m] = v;
}
}  // namespace

// Weight gradient of the stem's conv2 with a 16-bit h1 (cfm_subsample_conv1_relu_out16_f32) on the weight-gradient kernel of
// gemm_dw16_impl.h: dw2p (C, 3(kf), 3(kt), C) += dz2^T . im2col(h1), db2 (C) += column sums of dz2 (NULL: skipped; dz2 fp32 or
// 16-bit); `rowtab_scratch`: cfm_subsample_conv2_rowtab_elems(B, F1, T1)
// ints (any contents; 16-byte aligned).  C % 64 == 0; B*T1*F1*C < 2^31.
extern "C" int64_t cfm_subsample_conv2_rowtab_elems(int B, int F1, int T1) {
    if (B <= 0 || F1 < 3 || T1 < 3) return 0;
    const int64_t M = (int64_t)B * ((T1 - 1) / 2) * ((F1 - 1) / 2);
    return (M + 63) / 64 * 64 + 64;
}
extern "C" int cfm_subsample_conv2_bwd_weight_h16_mfma16_f32(int prec, const void* dz2, int dz2_is_16bit, const void* h1_16,
                                                             int* rowtab_scratch, float* dw2p, float* db2_or_null, int B, int F1,
                                                             int T1, int C, cfm_stream_t stream) {
    CFM_REQUIRE(dz2 && h1_16 && rowtab_scratch && dw2p, CFM_ERR_NULL);
    CFM_REQUIRE(B > 0 && F1 >= 3 && T1 >= 3 && C > 0 && (C % 64) == 0, CFM_ERR_BAD_SHAPE);
    CFM_REQUIRE((int64_t)B * T1 * F1 * C < ((int64_t)1 << 31), CFM_ERR_UNSUPPORTED);
    CFM_REQUIRE(CFM_ALIGNED16(dz2) && CFM_ALIGNED16(h1_16) && CFM_ALIGNED16(rowtab_scratch) && CFM_ALIGNED16(dw2p), CFM_ERR_ALIGN);
    const int T2 = (T1 - 1) / 2, F2 = (F1 - 1) / 2;
    const int64_t M = (int64_t)B * T2 * F2, n = cfm_subsample_conv2_rowtab_elems(B, F1, T1);
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(conv2_rowtab_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, rowtab_scratch, M, n, T2, F2, T1, F1);
    if (prec == CFM_PREC_BF16) return cfm_dw16_conv2_bf16(dz2, dz2_is_16bit, h1_16, rowtab_scratch, dw2p, db2_or_null, C, F1, M, s);
    if (prec == CFM_PREC_FP16) return cfm_dw16_conv2_f16(dz2, dz2_is_16bit, h1_16, rowtab_scratch, dw2p, db2_or_null, C, F1, M, s);
    return CFM_ERR_UNSUPPORTED;
}
