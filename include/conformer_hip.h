/*
 * conformer_hip.h -- C ABI of libconformer_hip.so: hand-written gfx950 (MI355X, CDNA4) kernels for the
 * Conformer encoder hot path.
 *
 * The reference (Alan-404/Conformer) has no FFI layer: its "operator API" is torch.nn.  Each entry
 * point below therefore cites the reference nn.Module call site(s) it replaces (file:line under the
 * reference tree).  Conventions:
 *   - plain pointers + sizes, no torch types; all pointers are DEVICE pointers unless stated;
 *   - the caller owns every buffer (outputs, workspaces); nothing here allocates, frees or
 *     synchronises; every launch is enqueued on `stream` (a hipStream_t passed as void*);
 *   - return 0 on success or a negative cfm_status; never throws; re-entrant (no mutable globals);
 *   - activations are row-major (B, T, d) fp32, d contiguous; weights keep the PyTorch layout of the
 *     reference state_dict unless a cfm_pack_* routine is named;
 *   - `lengths` are int64 device arrays of B valid-frame counts, or NULL for "no mask".
 */
#ifndef CONFORMER_HIP_H
#define CONFORMER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* cfm_stream_t; /* hipStream_t */

enum cfm_status {
    CFM_OK = 0,
    CFM_ERR_BAD_SHAPE = -1,    /* a dimension is <= 0 or violates a documented divisibility rule */
    CFM_ERR_UNSUPPORTED = -2,  /* e.g. head dim > 64, depthwise kernel > 63 */
    CFM_ERR_NULL = -3,         /* a required pointer is NULL */
    CFM_ERR_LAUNCH = -4,       /* hipGetLastError() reported a launch failure */
    CFM_ERR_DEVICE = -5,       /* current device is not gfx950 */
    CFM_ERR_ALIGN = -6         /* a pointer or leading dimension is not 16-byte aligned */
};

/* ---- library ------------------------------------------------------------------------------- */
int cfm_version(void);                  /* library version, currently 1 */
int cfm_abi_version(void);              /* CFM_ABI_VERSION this library was built against: bumped whenever an entry point's
                                           argument list changes or an entry point is removed; a binding refuses a mismatch */
#define CFM_ABI_VERSION 4
const char* cfm_strerror(int status);   /* static string */
int cfm_device_check(void);             /* CFM_OK iff the current HIP device is gfx950 */

/* ---- LayerNorm (nn.LayerNorm eps=1e-5 affine: ffn.py:8,16; attention.py:10,15;
 *      convolution.py:12,22; block.py:15,27).  One wave64 per row, shuffle reductions.
 *      y[r,:] = (x[r,:]-mean)/sqrt(var+eps)*gamma+beta.  d % 4 == 0, d <= 8192.
 *      mean/rstd (rows) are optional outputs for a later backward. */
int cfm_layernorm_fwd_f32(const float* x, const float* gamma, const float* beta, float* y,
                          float* mean_or_null, float* rstd_or_null,
                          int64_t rows, int d, float eps, cfm_stream_t stream);

/* ---- dense GEMMs on the fp32 MFMA pipe (v_mfma_f32_32x32x2_f32), C = A . W^T with epilogues.
 *      A: (M,K) row-major with leading dim lda; W: (N,K) row-major (= nn.Linear.weight, or
 *      Conv1d(k=1).weight viewed (N,K)); bias: (N).  K % 4 == 0, lda % 4 == 0, A and W 16-byte aligned.
 *
 *      _bias          : nn.Linear                      encoder.py:23, attention.py:78-81,90
 *      _bias_swish    : Linear + Swish                 ffn.py:17-18
 *      _bias_relu     : (used by the conv stem)        convolution.py:46-47
 *      _bias_glu      : Conv1d(k=1, d->2d) + GLU(dim=1) convolution.py:24-25; N = 2*n_out is the weight
 *                       row count, C is (M, n_out): C = (A.Wv^T+bv) * sigmoid(A.Wg^T+bg) with
 *                       Wv = W[0:n_out], Wg = W[n_out:2n_out]
 *      _bias_residual : Linear then `alpha*y + R`      block.py:19,21,23,25 (R: (M,N), leading dim ldr) */
int cfm_gemm_bias_f32(const float* A, const float* W, const float* bias, float* C,
                      int64_t M, int N, int K, int64_t lda, int64_t ldc, cfm_stream_t stream);
int cfm_gemm_bias_swish_f32(const float* A, const float* W, const float* bias, float* C,
                            int64_t M, int N, int K, int64_t lda, int64_t ldc, cfm_stream_t stream);
int cfm_gemm_bias_relu_f32(const float* A, const float* W, const float* bias, float* C,
                           int64_t M, int N, int K, int64_t lda, int64_t ldc, cfm_stream_t stream);
int cfm_gemm_bias_glu_f32(const float* A, const float* W, const float* bias, float* C,
                          int64_t M, int n_out, int K, int64_t lda, int64_t ldc, cfm_stream_t stream);
int cfm_gemm_bias_residual_f32(const float* A, const float* W, const float* bias, const float* R,
                               float alpha, float* C, int64_t M, int N, int K,
                               int64_t lda, int64_t ldr, int64_t ldc, cfm_stream_t stream);

/* ---- split-K form of the three plain epilogues for SMALL M (streaming chunks: M = 1280): a wave that owns a whole-K output
 *      tile runs K/2 dependent MFMAs however few tiles the product has, so the contraction is cut into `splits` slices (one
 *      workgroup per tile and slice, raw partial tiles into workspace (splits, M, N) fp32) and summed in a fixed order by a
 *      reduce + epilogue pass (bit-reproducible).  epi: 0 bias | 1 +swish | 4 alpha*y + R.  N, ldc, ldr % 4 == 0. */
int cfm_gemm_splitk_f32(int epi, const float* A, const float* W, const float* bias, const float* R_or_null,
                        float alpha, float* C, float* workspace, int splits, int64_t M, int N, int K,
                        int64_t lda, int64_t ldr, int64_t ldc, cfm_stream_t stream);

/* ---- LayerNorm folded into the GEMMs either side of it (inference, fp32 MFMA): the LayerNorms in front of the FFN, the
 *      attention projections and pointwise_conv_1 (ffn.py:16, attention.py:15, convolution.py:22) leave the launch list.
 *      Producer side: the GEMM (or LayerNorm) that WRITES a residual-stream row also writes its statistics partials,
 *        stats[row][part] = (sum, M2 about the partial's own mean) of `d / parts` consecutive stored values:
 *        cfm_gemm_bias_stats_f32 / cfm_gemm_bias_residual_stats_f32: one partial per 32 columns (N % 32 == 0, stats (M, N/32, 2));
 *        cfm_layernorm_fwd_stats_f32 (block.py:27 followed by the next block's ffn.py:16): one partial per row of its OUTPUT.
 *      Consumer side: cfm_gemm_lnfold_f32 computes epi(LN(A).W^T + b) from the UN-normalised A as
 *        rstd * (A.Wf^T - mean * colsum) + bias_f with Wf = W.diag(gamma), bias_f = b + W.beta, colsum[n] = sum_k Wf[n,k]
 *        (folded once per weight version by the caller); mean / rstd are merged from the partials (Chan's formula, fixed
 *        order) by each workgroup for its rows.  epi: 0 bias | 1 +swish | 3 +GLU (N = n_out; Wf, bias_f, colsum have 2N
 *        rows: values then gates).  ln_parts in {1, 2, 4, 8, 16}, K % ln_parts == 0; N, ldc % 4 == 0; 16-byte aligned C / bias_f /
 *        colsum / ln_stats. */
int cfm_gemm_bias_stats_f32(const float* A, const float* W, const float* bias, float* C, float* stats_out,
                            int64_t M, int N, int K, int64_t lda, int64_t ldc, cfm_stream_t stream);
int cfm_gemm_bias_residual_stats_f32(const float* A, const float* W, const float* bias, const float* R,
                                     float alpha, float* C, float* stats_out, int64_t M, int N, int K,
                                     int64_t lda, int64_t ldr, int64_t ldc, cfm_stream_t stream);
int cfm_gemm_lnfold_f32(int epi, const float* A, const float* ln_stats, int ln_parts, float ln_eps,
                        const float* Wf, const float* bias_f, const float* colsum, float* C,
                        int64_t M, int N, int K, int64_t lda, int64_t ldc, cfm_stream_t stream);
int cfm_layernorm_fwd_stats_f32(const float* x, const float* gamma, const float* beta, float* y,
                                float* stats_out, int64_t rows, int d, float eps, cfm_stream_t stream);

/* ---- fused Macaron feed-forward sub-layer (inference, fp32 MFMA; round 3): ffn.py:15-23 + the residual of block.py:19,25 and,
 *      in mode 2, the block-closing LayerNorm of block.py:27 -- ONE kernel per 32 rows, the (rows, hidden) activation stays on chip:
 *        Y = alpha * (swish(LN(X).W1^T + b1).W2^T + b2) + X            (LN folded as in cfm_gemm_lnfold_f32: ln_stats of X's rows,
 *                                                                        b1f / colsum1 = the folded bias / column sums of W1f)
 *      mode 0: Y only | 1: + stats_out (M, d/32, 2), the statistics partials of the stored rows (as cfm_gemm_bias_residual_stats_f32)
 *           | 2: Y = LayerNorm(...; gamma2, beta2, eps2), stats_out (M, 1, 2) of the LayerNorm OUTPUT or NULL.
 *      Wp: both weights in MFMA fragment order (cfm_ffn_pack_f32 from W1f (hidden, d) = W1.diag(gamma) and W2 (d, hidden);
 *      cfm_ffn_pack_elems floats; once per weight version).  d in {128, 256, 512}, hidden % 128 == 0; ln_parts in {1,...,16}. */
int64_t cfm_ffn_pack_elems(int d, int hidden);
int cfm_ffn_pack_f32(const float* W1f, const float* W2, float* Wp, int d, int hidden, cfm_stream_t stream);
int cfm_ffn_fused_f32(const float* X, int64_t ldx, const float* ln_stats, int ln_parts, float ln_eps, const float* Wp,
                      const float* b1f, const float* colsum1, const float* b2, float alpha, float* Y, int64_t ldy,
                      int mode, float* stats_out, const float* gamma2, const float* beta2, float eps2, int64_t M,
                      int d, int hidden, cfm_stream_t stream);

/* ---- row-local chains of a Conformer block (inference, fp32 MFMA; round 3): everything between two operators that mix rows
 *      (attention, the depthwise convolution) runs in ONE kernel per 32 rows, the rows resident in LDS, weights streamed in MFMA
 *      fragment order (cfm_rowgemm_pack_f32 for the d-input Linear layers, cfm_ffn_pack_f32 for the feed-forward):
 *        K1 = (pre 0, core 1, post 1, mode 0)  block.py:19-21: Y = x + FFN1(x)/2 (as cfm_ffn_fused_f32, ln_stats of x's rows);
 *                                              Z (M, 3d) = LN(Y).Wqkv^T + b, the LayerNorm of attention.py:15 folded (Wpost = the
 *                                              packed folded weight, bpost / cspost = folded bias / column sums, post_eps)
 *        K2 = (pre 1, core 0, post 2)          block.py:21-23: Y1 = X.Wout^T + b + R (attention.py:90 + residual; stored);
 *                                              Z (M, d) = GLU(LN(Y1).Wpw1^T + b), convolution.py:22-25 (ln_eps = that LayerNorm's)
 *        K3 = (pre 1, core 1, post 0, mode 2)  block.py:23-27: Y1 = X.Wpw2^T + b + R (convolution.py:29 + residual; Y1 may be NULL);
 *                                              Y = LayerNorm(Y1 + FFN2(Y1)/2; gamma2, beta2, eps2), stats_out (M, 1, 2) or NULL
 *      Unused stage arguments are NULL / 0.  d in {128, 256, 512}; leading dimensions % 4 == 0, 16-byte aligned pointers.
 *      cfm_rowgemm_pack_f32: W (N, d) -> packed (same element count); glu = 1: W is (2 d, d), value rows then gate rows. */
int cfm_rowgemm_pack_f32(const float* W, float* Wp, int N, int d, int glu, cfm_stream_t stream);
int cfm_rowchain_f32(int pre, int core, int post, int mode, const float* X, int64_t ldx, const float* Wpre_packed,
                     const float* bpre, const float* R, int64_t ldr, float* Y1, int64_t ldy1, const float* ln_stats,
                     int ln_parts, float ln_eps, const float* Wffn_packed, const float* b1f, const float* colsum1,
                     const float* b2, float alpha, int hidden, float* Y, int64_t ldy, float* stats_out,
                     const float* gamma2, const float* beta2, float eps2, const float* Wpost_packed, const float* bpost,
                     const float* cspost, float post_eps, float* Z, int64_t ldz, int64_t M, int d, cfm_stream_t stream);
int64_t cfm_ffn_tile_stride_f4(int d);   /* layout parameters of cfm_ffn_pack_f32 shared by the two kernels (internal) */
int cfm_ffn_rotate(void);

/* ---- relative positional encoding table (RelativePositionalEncoding.forward, position.py:11-27,
 *      WITHOUT the batch repeat of position.py:26).  pe: (2T-1, d); row j encodes r = T-1-j:
 *      pe[j,2c] = sin(r*div_term[c]), pe[j,2c+1] = cos(r*div_term[c]).  div_term: (d/2). */
int cfm_relpos_table_f32(const float* div_term, float* pe, int T, int d, cfm_stream_t stream);

/* ---- fused relative-position attention (scaled_dot_product_relative_attention + _relative_shift
 *      + head split/concat, attention.py:47-72,78-88,94-102).
 *      q,k,v: row (b*T+t) at ptr + (b*T+t)*ld, head h in columns [h*dh,(h+1)*dh) (so a fused
 *      (B*T, 3d) QKV buffer is passed as q=buf, k=buf+d, v=buf+2d, ld=3d);
 *      pos: (2T-1, H*dh) PROJECTED table (pos_proj applied once, not per batch), leading dim ldp;
 *      u, vbias: content_bias / position_bias (H, dh);  lengths: keys k >= lengths[b] are masked.
 *      score[i,k] = ((q_i+u).k_k + (q_i+vbias).pos[T-1-(i-k)]) / sqrt(dh); softmax over k; ctx = P.V.
 *      ctx: (B*T, H*dh) leading dim ldo.  lse_or_null: (B,H,T) log-sum-exp for a later backward.
 *      dh <= 64 and dh % 4 == 0; ld, ldp, ldo % 4 == 0.  Scores are never materialised. */
int cfm_relpos_attention_fwd_f32(const float* q, const float* k, const float* v, int64_t ld,
                                 const float* pos, int64_t ldp, const float* u, const float* vbias,
                                 const int64_t* lengths_or_null, float* ctx, int64_t ldo,
                                 float* lse_or_null, int B, int T, int H, int dh, cfm_stream_t stream);

/* ---- convolution module core: depthwise Conv1d(k=K, pad (K-1)/2, groups=C) + BatchNorm1d (eval,
 *      running stats) + Swish, channel-last (convolution.py:26-28).
 *      g: (B,T,C) GLU output; w: (C,1,K) = (C,K); y: (B,T,C).  Zero padding is per utterance over
 *      [0,T) and padded frames are NOT masked.  K odd, K <= 63. */
int cfm_dwconv_bn_swish_fwd_f32(const float* g, const float* w, const float* bias,
                                const float* bn_weight, const float* bn_bias,
                                const float* bn_mean, const float* bn_var, float bn_eps,
                                float* y, int B, int T, int C, int K, cfm_stream_t stream);

/* ---- convolution subsampling stem (ConvolutionSubsampling.forward, convolution.py:42-57)
 *      x: (B, F, T) log-mel (mel axis is the conv "height").  T1=(T-1)/2, F1=(F-1)/2, T2=(T1-1)/2,
 *      F2=(F1-1)/2 (integer division).
 *      conv1: h1 (B,T1,F1,C) channel-last = relu(conv2d(x, w1 (C,1,3,3), stride 2) + b1)
 *      pack : w2p (C, 3,3, C) = w2 (C_out, C_in, 3,3) permuted to (C_out, kf, kt, C_in)
 *      conv2: h2 (B,T2,F2,C) channel-last = relu(implicit-GEMM conv2d(h1, w2p, stride 2) + b2) on MFMA
 *      pack : wlp (d, F2*C) = linear.weight (d, C*F2) with columns reordered (c*F2+f) -> (f*C+c),
 *             so that encoder.py:23 becomes cfm_gemm_bias_f32 on h2 viewed (B*T2, F2*C).
 *      C % 16 == 0. */
int cfm_subsample_conv1_relu_f32(const float* x, const float* w1, const float* b1, float* h1,
                                 int B, int F, int T, int C, cfm_stream_t stream);
int cfm_pack_conv2_weight_f32(const float* w2, float* w2p, int C, cfm_stream_t stream);
int cfm_subsample_conv2_relu_f32(const float* h1, const float* w2p, const float* b2, float* h2,
                                 int B, int F1, int T1, int C, cfm_stream_t stream);
int cfm_pack_linear_weight_f32(const float* wl, float* wlp, int d_out, int C, int F2, cfm_stream_t stream);

/* ---- 16-bit-MFMA GEMMs with fp32 storage: the arithmetic torch.autocast gives nn.Linear / Conv (train.py:6,232 runs the
 *      model under torch.cuda.amp.autocast = fp16; bf16 is the MI355X-preferred variant).  Operands
 *      are rounded to `prec` on their way into LDS, accumulation and epilogues are fp32, tensors in HBM stay fp32.
 *      epi: 0 bias | 1 +swish | 2 +relu | 3 +GLU | 4 alpha*y+R | 5 backward of swish: C = alpha * (A.W^T) * swish'(Z)
 *      (Z_or_null is READ, ldr is its leading dimension, bias may be NULL, N % 8 == 0); argument rules as the fp32 entry points
 *      (cfm_gemm_train_f32 for Z_or_null / drop_p / drop_seed; for GLU, N = n_out and W has 2*n_out rows).
 *      The *_bwd_* entries mirror cfm_gemm_bwd_batched_f32 / cfm_subsample_conv2_bwd_{weight,input}_f32 (the stem
 *      forms need C % 64 == 0). */
#define CFM_PREC_F32 0
#define CFM_PREC_BF16 1
#define CFM_PREC_FP16 2
int cfm_gemm_mfma16_f32(int prec, int epi, const void* A, int a_is_16bit, const void* W, int w_is_16bit,
                        const float* bias, const float* R_or_null, float alpha, void* C, int c_is_16bit,
                        void* Z_or_null, int z_is_16bit, int64_t M, int N, int K, int64_t lda, int64_t ldr,
                        int64_t ldc, float drop_p, uint64_t drop_seed, cfm_stream_t stream);
/*      w_is_16bit / b_is_16bit: that operand is already stored in `prec` (cfm_cast16_f32 of the fp32 master weights, once
 *      per optimizer step): half the bytes of the operand every row tile re-reads; results are bit-identical. K % 8 == 0.
 *      a_is_16bit (needs w_is_16bit; lda % 8 == 0, in elements): A is a 16-bit tensor written by its producer
 *      (cfm_layernorm_fwd_out16_f32, or a GEMM with c_is_16bit).  c_is_16bit: C is stored in `prec` (ldc in elements).
 *      z_is_16bit (epi 1, N % 8 == 0): the saved pre-activation Z is stored in `prec` too (what autocast keeps for the
 *      backward of silu: reference model/utils/activation.py). */
/* Multi-tensor cast of fp32 master weights to `prec` (optionally transposed): one launch per CFM_CAST_BATCH items. */
#define CFM_CAST_BATCH 48
typedef struct cfm_cast_item {
    const float* src;   /* (rows, cols) fp32, contiguous */
    void* dst;          /* 16-bit: (rows, cols), or (cols, rows) when transpose != 0 */
    int64_t rows, cols;
    int transpose;
} cfm_cast_item;
int cfm_cast16_multi_f32(int prec, const cfm_cast_item* items, int count, cfm_stream_t stream);
int cfm_layernorm_fwd_out16_f32(int prec, const float* x, const float* gamma, const float* beta, void* y16,
                                float* mean_or_null, float* rstd_or_null, int64_t rows, int d, float eps,
                                cfm_stream_t stream);
int cfm_cast16_f32(int prec, const float* src, void* dst, int64_t n, cfm_stream_t stream);
int cfm_relpos_attention_mfma16_f32(int prec, const float* q, const float* k, const float* v, int64_t ld,
                                    const float* pos, int64_t ldp, const float* u, const float* vbias,
                                    const int64_t* lengths_or_null, float* ctx, int64_t ldo, float* lse_or_null, int B,
                                    int T, int H, int dh, float drop_p, uint64_t drop_seed, cfm_stream_t stream);
/* Streaming under autocast (round 3): the 16-bit form of cfm_relpos_attention_rows_f32 -- query rows [q_begin, q_begin + q_count)
 * against keys < lengths[b] of the fp32 K/V cache (rounded to `prec` where they enter a product), fp32 ctx; no key split. */
int cfm_relpos_attention_rows_mfma16_f32(int prec, const float* q, const float* k, const float* v, int64_t ld,
                                         const float* pos, int64_t ldp, const float* u, const float* vbias,
                                         const int64_t* lengths, float* ctx, int64_t ldo, int B, int T, int H, int dh,
                                         int q_begin, int q_count, cfm_stream_t stream);
/* Inference under autocast with 16-bit tensors either side of the core: qkv_is_16bit -- q / k / v stored in `prec` (ld in elements,
 * ld % 8 == 0: what autocast's projections hand the attention); ctx_is_16bit -- the context stored in `prec` (ldo in elements): its
 * only consumer is the out-projection GEMM, which rounds an fp32 context to that type anyway (bit-identical layer output). */
int cfm_relpos_attention_io16_mfma16_f32(int prec, const void* q, const void* k, const void* v, int qkv_is_16bit, int64_t ld,
                                         const float* pos, int64_t ldp, const float* u, const float* vbias,
                                         const int64_t* lengths_or_null, void* ctx, int ctx_is_16bit, int64_t ldo, int B, int T,
                                         int H, int dh, cfm_stream_t stream);
int cfm_subsample_conv2_relu_mfma16_f32(int prec, const void* h1, int h1_is_16bit, const void* w2p, int w_is_16bit,
                                        const float* b2, void* h2, int h2_is_16bit, int B, int F1, int T1, int C,
                                        cfm_stream_t stream);
int cfm_subsample_conv1_relu_out16_f32(int prec, const float* x, const float* w1, const float* b1, void* h1, int B, int F,
                                       int T, int C, cfm_stream_t stream);
int cfm_gemm_bwd_batched_mfma16_f32(int prec, const float* A, int a_col, int64_t lda, const void* B, int b_col,
                                    int b_is_16bit, int64_t ldb, const void* Z_or_null, int z_is_16bit, int64_t ldz, float alpha, void* C,
                                    int64_t ldc, int c_is_16bit, int I, int J, int64_t Kc, int allow_split, int accumulate,
                                    int nbatch, int nb1, int64_t sa0, int64_t sa1, int64_t sb0, int64_t sb1,
                                    int64_t sc0, int64_t sc1, float drop_p, uint64_t drop_seed,
                                    int operands_zero_padded4, cfm_stream_t stream);
/*      c_is_16bit (swish'(Z) product only, J % 8 == 0): C is stored in `prec` (ldc in elements) -- a gradient whose only
 *      consumers are GEMM operands.
 *      operands_zero_padded4: ragged Kc (index-major operands) / I, J (contraction-major operands) are physically padded
 *      to a multiple of 4 elements with zeros, so partial 16-byte chunks may be loaded whole (the fast load path). */
/* Weight (and bias) gradient of y = x.W^T + b on the 16-bit matrix pipe: dw (N,K) += alpha * dy^T.x,
 * db (N) += alpha * colsum(dy) (NULL: skipped); both accumulate with atomics (caller zero-fills).  dy (M,N), x (M,K)
 * row-major, either may be stored in the 16-bit type of `prec` (ld in elements).  N % 8 == 0, K % 8 == 0.
 * Replaces the autograd of nn.Linear under autocast (reference model/utils/ffn.py:15-23). */
int cfm_linear_bwd_weight_mfma16_f32(int prec, const void* dy, int dy_is_16bit, int64_t ldy, const void* x,
                                     int x_is_16bit, int64_t ldx, float* dw, int64_t ldw, float* db_or_null, int N,
                                     int K, int64_t M, float alpha, cfm_stream_t stream);
/* Weight gradient of the stem's conv2 from a 16-bit h1 (training under autocast keeps h1 in `prec`): im2col gather inside the
 * weight-gradient kernel; rowtab_scratch: cfm_subsample_conv2_rowtab_elems(B, F1, T1) ints.  dw2p and db2 (bias gradient,
 * may be NULL) accumulate (caller zero-fills); dz2 fp32 or stored in `prec`. */
int64_t cfm_subsample_conv2_rowtab_elems(int B, int F1, int T1);
int cfm_subsample_conv2_bwd_weight_h16_mfma16_f32(int prec, const void* dz2, int dz2_is_16bit, const void* h1_16,
                                                  int* rowtab_scratch, float* dw2p, float* db2_or_null, int B, int F1, int T1,
                                                  int C, cfm_stream_t stream);
/* 16-bit-output forms of producers whose results only feed 16-bit GEMM operands under autocast (identical results, half the
 * bytes): the conv module's depthwise-conv + BatchNorm + Swish output (K in {3,7,15,31}) and the GLU backward. */
int cfm_dwconv_bn_swish_fwd_out16_f32(int prec, const float* g, const float* w, const float* bias, const float* bn_weight,
                                      const float* bn_bias, const float* bn_mean, const float* bn_var, float bn_eps,
                                      void* y16, int B, int T, int C, int K, cfm_stream_t stream);
int cfm_glu_bwd_out16_f32(int prec, const float* z, const float* dy, void* dz16, int64_t rows, int n, cfm_stream_t stream);
/* ReLU backward with the result stored in `prec` (cfm_relu_bwd_f32 otherwise). */
int cfm_relu_bwd_out16_f32(int prec, const float* y, const float* dy, void* dz16, int64_t n, cfm_stream_t stream);
/* Input gradient of the stem's conv2 on the forward 16-bit GEMM kernel (four parity-class implicit GEMMs with a per-row
 * tap-validity gather): w2c16 = cfm_pack_conv2_weight_t_f32's pack cast to `prec`; zero_bias: C zeros.  C % 64 == 0.
 * Same result as cfm_subsample_conv2_bwd_input_mfma16_f32 (reference: autograd of convolution.py:46-47). */
int cfm_subsample_conv2_bwd_input_fwdkernel_mfma16_f32(int prec, const void* dz2, int dz2_is_16bit, const void* w2c16,
                                                       const float* zero_bias, float* dh1, int B, int F1, int T1,
                                                       int C, cfm_stream_t stream);
/* ... the same with dh1 stored in the 16-bit type `prec` (its only consumer is the conv1 parameter-gradient reduction; under
 * torch.autocast conv1's incoming gradient is a 16-bit tensor), and that reduction reading it: */
int cfm_subsample_conv2_bwd_input_fwdkernel_out16_mfma16_f32(int prec, const void* dz2, int dz2_is_16bit, const void* w2c16,
                                                             const float* zero_bias, void* dh1_16, int B, int F1, int T1, int C,
                                                             cfm_stream_t stream);
int cfm_subsample_conv1_bwd_d16_f32(int prec, const float* x, const float* w1, const float* b1, const void* dh1_16, float* dw1,
                                    float* db1, int B, int F, int T, int C, cfm_stream_t stream);
int cfm_subsample_conv2_bwd_weight_mfma16_f32(int prec, const float* dz2, const float* h1, float* dw2p, int B, int F1,
                                              int T1, int C, cfm_stream_t stream);
int cfm_subsample_conv2_bwd_input_mfma16_f32(int prec, const float* dz2, const float* w2c, float* dh1, int B, int F1,
                                             int T1, int C, cfm_stream_t stream);

/* ---- audio front end (processing/processor.py:53-63,155-158,373-394; processing/augment.py:7-19).
 *      log-mel = reflect_pad -> [frames * window * DFT] as cfm_gemm_bwd_batched_f32 over overlapping rows
 *      (A = padded wave, lda = hop; B = windowed (2*n_bins, n_fft) cos/-sin basis) -> power_mel_log:
 *      out (B, n_mels, T) = log(max(fb^T . (Re^2+Im^2), floor)); spec rows = frames (B*T, ld_spec), fb (n_bins, n_mels).
 *      specaugment_apply: bands (n,3) int32 device rows {axis 1=freq | 2=time, start, end}, applied to every utterance. */
int cfm_reflect_pad_f32(const float* x, float* xp, int B, int64_t L, int pad, int64_t ld_out, cfm_stream_t stream);
int cfm_power_mel_log_f32(const float* spec, int64_t ld_spec, const float* fb, float* out, int B, int T,
                          int n_bins, int n_mels, float floor_value, cfm_stream_t stream);
/*      the same on the fp32 matrix pipe (round 3): spec rows are [Re | 0-pad to nk | Im | 0-pad to 2 nk], nk = n_bins rounded up to
 *      8, as written by a DFT GEMM whose basis carries those zero rows; fbT (96, nk) = the filterbank transposed and zero-padded.
 *      One workgroup per 32 frames, mel products on v_mfma_f32_32x32x2_f32, the spectrum read once. */
int cfm_power_mel_log_mfma_f32(const float* spec, int64_t ld_spec, const float* fbT, float* out, int B, int T,
                               int rows_per_utt, int n_bins, int n_mels, float floor_value, cfm_stream_t stream);
/*      the DFT in front of it on the tuned forward GEMM: spec (rows, n_cols) = frames . basis^T, frame r = wave[r*hop : r*hop+n_fft]
 *      (overlapping rows).  Utterances are laid out with a pitch of rows_per_utt * hop floats (rows_per_utt >= T + 3 so that the
 *      pitch covers L + n_fft), row b * rows_per_utt + t = frame t of utterance b; n_fft floats of slack behind the last one. */
int cfm_dft_frames_f32(const float* wave, const float* basis, float* spec, int64_t rows, int n_cols, int n_fft,
                       int hop, cfm_stream_t stream);
int cfm_specaugment_apply_f32(float* spec, int B, int F, int T, const int* bands, int nbands, float value,
                              cfm_stream_t stream);

/* =============================== backward pass (fp32) =========================================
 * The reference relies on autograd (train.py:239); these are the explicit kernels behind the
 * torch.autograd.Function wrappers in conformer_amd/autograd.py.  Parameter-gradient outputs that
 * are documented as "accumulated" must be zero-filled by the caller (they are summed with fp32
 * atomics).  Layout conventions as in the forward section. */

/* training forward of ffn.py:17-18: C = swish(Z) and Z = A.W^T + bias are both stored. */
int cfm_gemm_bias_swish_save_f32(const float* A, const float* W, const float* bias, float* C, float* Z,
                                 int64_t M, int N, int K, int64_t lda, int64_t ldc, cfm_stream_t stream);

/* General MFMA GEMM of the backward pass: C (I x J) (+)= alpha * sum_k A(i,k) B(j,k) [* swish'(Z)].
 * a_col / b_col = 0: operand stored index-major X[idx*ld + k]; = 1: contraction-major X[k*ld + idx].
 *   dX = dY.W    : A = dY (a_col 0), B = W (b_col 1), I = M, J = K_in, Kc = N_out
 *   dW = dY^T.X  : A = dY (a_col 1), B = X (b_col 1), I = N_out, J = K_in, Kc = M  (allow_split: fp32 atomics
 *                  over contraction slices into a ZERO-FILLED C)
 * lda, ldb, ldc (and batch strides) multiples of 4; pointers 16-byte aligned.  Z only with (a_col 0, b_col 1). */
int cfm_gemm_bwd_f32(const float* A, int a_col, int64_t lda, const float* B, int b_col, int64_t ldb,
                     const float* Z_or_null, int64_t ldz, float alpha, float* C, int64_t ldc,
                     int I, int J, int64_t Kc, int allow_split, cfm_stream_t stream);
int cfm_gemm_bwd_batched_f32(const float* A, int a_col, int64_t lda, const float* B, int b_col, int64_t ldb,
                             const float* Z_or_null, int64_t ldz, float alpha, float* C, int64_t ldc,
                             int I, int J, int64_t Kc, int allow_split, int accumulate, int nbatch, int nb1,
                             int64_t sa0, int64_t sa1, int64_t sb0, int64_t sb1, int64_t sc0, int64_t sc1,
                             float drop_p, uint64_t drop_seed, cfm_stream_t stream);

/* Dropout (nn.Dropout of ffn.py:19,21, attention.py:17,67, convolution.py:30, encoder.py:25) is a counter-based mask:
 * element idx of a tensor is kept (scaled by 1/(1-p)) or zeroed as a pure function of (seed, idx), so the backward
 * regenerates it.  The stream is NOT torch's Philox stream (parity runs use p = 0).
 *   cfm_gemm_train_f32: the forward GEMMs with the mask fused in the epilogue (epi 0 bias, 1 swish [+Z saved], 4 residual)
 *   cfm_relpos_attention_train_f32: mask on the softmax weights, index ((b*H+h)*T + i)*T + k
 *   cfm_dropout_f32: y = x * mask (flat index), used on incoming gradients in the backward                      */
int cfm_gemm_train_f32(int epi, const float* A, const float* W, const float* bias, const float* R_or_null,
                       float alpha, float* C, float* Z_or_null, int64_t M, int N, int K, int64_t lda,
                       int64_t ldr, int64_t ldc, float drop_p, uint64_t drop_seed, cfm_stream_t stream);
int cfm_relpos_attention_train_f32(const float* q, const float* k, const float* v, int64_t ld,
                                   const float* pos, int64_t ldp, const float* u, const float* vbias,
                                   const int64_t* lengths_or_null, float* ctx, int64_t ldo, float* lse,
                                   int B, int T, int H, int dh, float drop_p, uint64_t drop_seed,
                                   cfm_stream_t stream);
int cfm_dropout_f32(const float* x, float* y, int64_t n, float p, uint64_t seed, cfm_stream_t stream);
/* ... with y stored in the 16-bit type `prec` (a masked gradient that only feeds GEMM operands); p = 0 is a plain cast.  n % 8 == 0. */
int cfm_dropout_out16_f32(int prec, const float* x, void* y16, int64_t n, float p, uint64_t seed, cfm_stream_t stream);

/* LayerNorm backward (mean/rstd = the forward's saved row statistics).  dx = LN'(dy) [+ dres];
 * dgamma/dbeta accumulated. */
int cfm_layernorm_bwd_dx_f32(const float* x, const float* gamma, const float* dy, const float* mean,
                             const float* rstd, const float* dres_or_null, float* dx, int64_t rows, int d,
                             cfm_stream_t stream);
int cfm_layernorm_bwd_params_f32(const float* x, const float* dy, const float* mean, const float* rstd,
                                 float* dgamma, float* dbeta, int64_t rows, int d, cfm_stream_t stream);
/* Both in one pass over x and dy (d <= 2048); per-workgroup partial sums go through `workspace` and are combined
 * in a fixed order (bit-reproducible parameter gradients). */
size_t cfm_layernorm_bwd_workspace_bytes(int64_t rows, int d);
int cfm_layernorm_bwd_f32(const float* x, const float* gamma, const float* dy, const float* mean,
                          const float* rstd, const float* dres_or_null, float* dx, float* dgamma, float* dbeta,
                          int64_t rows, int d, void* workspace, size_t workspace_bytes, cfm_stream_t stream);

/* out[c] += alpha * sum_r X[r][c]  (bias gradients; out accumulated). */
int cfm_colsum_f32(const float* X, int64_t ld, int64_t rows, int cols, float alpha, float* out,
                   cfm_stream_t stream);

/* GLU on a stored pre-activation z = [a | g] (rows, 2n): y = a*sigmoid(g); and its backward dz. */
int cfm_glu_fwd_f32(const float* z, float* y, int64_t rows, int n, cfm_stream_t stream);
int cfm_glu_bwd_f32(const float* z, const float* dy, float* dz, int64_t rows, int n, cfm_stream_t stream);

/* Backward of cfm_dwconv_bn_swish_fwd_f32.  train_stats = 0: bn_mean/bn_var are constants (eval);
 * 1: they are the batch statistics and BatchNorm's mean/variance coupling is applied (n = B*T).
 * dc_ws: (B,T,C) workspace; dg: (B,T,C); dw (C,K), dbias, dgamma, dbeta (C): accumulated.  K in {3,7,15,31}. */
int cfm_dwconv_bn_swish_bwd_f32(const float* g, const float* dy, const float* w, const float* bias,
                                const float* bn_weight, const float* bn_bias, const float* bn_mean,
                                const float* bn_var, float bn_eps, int train_stats, float* dc_ws, float* dg,
                                float* dw, float* dbias, float* dgamma, float* dbeta, int B, int T, int C,
                                int K, cfm_stream_t stream);

/* Train-mode BatchNorm1d statistics of the depthwise-conv output (convolution.py:26-27 under .train()):
 * batch_mean / batch_var (biased) over all B*T positions, padded frames included; running_mean/var (or NULL)
 * updated in place: running = (1-momentum)*running + momentum*{mean, unbiased var}.  Feed batch_mean/var to
 * cfm_dwconv_bn_swish_fwd_f32 as bn_mean/bn_var for the train-mode forward.  One pass over g, bit-reproducible
 * (per-workgroup (count, mean, M2) partials in `workspace`, merged in a fixed order; no atomics). */
size_t cfm_dwconv_bn_stats_workspace_bytes(int B, int T, int C);
int cfm_dwconv_bn_stats_f32(const float* g, const float* w, const float* bias, float* batch_mean,
                            float* batch_var, float* running_mean_or_null, float* running_var_or_null,
                            float momentum, int B, int T, int C, int K, void* workspace, size_t workspace_bytes,
                            cfm_stream_t stream);

/* ---- fused (flash-style) backward of the attention core: ONE launch, no (B,H,T,T) / (H,B,T,2T-1) tensor
 *      (replaces autograd through model/utils/attention.py:47-72,94-102).  Arguments as cfm_relpos_attention_train_f32,
 *      plus the forward's context `ctx` and log-sum-exp `lse` (B,H,T) and the context gradient `dctx` (layout of ctx,
 *      row stride ldo).  dq / dk / dv: row stride ldg.  dq, dpos ((2T-1) rows, stride lddp), du and dvbias (H*dh) are
 *      ACCUMULATED INTO with fp32 atomics (the caller zero-fills them); dk / dv are written.
 *      prec: CFM_PREC_F32, or the 16-bit type the forward kernel ran in under autocast (its operand rounding is replayed
 *      so the recomputed probabilities match the saved log-sum-exp; the products themselves stay fp32). */
int cfm_relpos_attention_bwd_f32(const float* q, const float* k, const float* v, int64_t ld, const float* pos,
                                 int64_t ldp, const float* u, const float* vbias, const int64_t* lengths_or_null,
                                 const float* ctx, const float* dctx, int64_t ldo, const float* lse, float* dq,
                                 float* dk, float* dv, int64_t ldg, float* dpos, int64_t lddp, float* du,
                                 float* dvbias, int B, int T, int H, int dh, float drop_p, uint64_t drop_seed,
                                 int prec, cfm_stream_t stream);

/* the same under torch.autocast (prec = CFM_PREC_BF16 / CFM_PREC_FP16, the type cfm_relpos_attention_mfma16_f32 ran in):
 * every product on the 16-bit matrix pipe with autocast's operand rounding, fp32 softmax / accumulation / outputs */
int cfm_relpos_attention_bwd_mfma16_f32(int prec, const float* q, const float* k, const float* v, int64_t ld,
                                        const float* pos, int64_t ldp, const float* u, const float* vbias,
                                        const int64_t* lengths_or_null, const float* ctx, const float* dctx, int64_t ldo,
                                        const float* lse, float* dq, float* dk, float* dv, int64_t ldg, float* dpos,
                                        int64_t lddp, float* du, float* dvbias, int B, int T, int H, int dh,
                                        float drop_p, uint64_t drop_seed, cfm_stream_t stream);

/* diagnostics: per-phase s_memrealtime stamps of one wave of the fused attention backward (see the .hip file) */
int cfm_debug_attention_bwd_trace_f32(void* trace_or_null);
int cfm_debug_attention_bwd_trace_mfma16(void* trace_or_null);
int cfm_debug_dw16_trace(void* trace_or_null);            /* 2 x 64 stamps of cfm_linear_bwd_weight_mfma16_f32 */
int cfm_debug_lstm_trace(void* trace_or_null);            /* T x 2 x 8 stamps of cfm_lstm_fwd_f32's steps */
int cfm_debug_gemm_mfma16_force_tile(int tile);           /* tuning: 0 auto, 1 128x128 family, 2 256x128, 3 256x256 */
int cfm_debug_gemm_mfma16_trace(void* trace_or_null);     /* per-K-tile stamps of two workgroups of cfm_gemm_mfma16_f32 */

/* ---- tuning / diagnostics: the residual-epilogue GEMM with a forced block-tile shape
 *      (cfg 0..3 = 128x128, 128x64, 64x128, 64x64; -1 = built-in heuristic).  Same results for every cfg.
 *      trace_or_null: 8 x uint64 per block {start, main-loop end, HW_ID, XCC_ID, epilogue issued,
 *      epilogue drained, -, -}; times in 100 MHz ticks. */
int cfm_debug_gemm_cfg_f32(int cfg, const float* A, const float* W, const float* bias, const float* R,
                           float alpha, float* C, int64_t M, int N, int K, void* trace_or_null,
                           cfm_stream_t stream);
int cfm_debug_ffn_trace(void* trace_or_null, int per_slice);   /* 128 x uint64 s_memrealtime stamps of cfm_ffn_fused_f32 (wave 0 of blocks 0 / 128), or NULL */
int cfm_debug_ffn_variant(int v);               /* 0 | 1 = main loop without weight loads (wrong results: the pure MFMA rate) */
int cfm_debug_ffn_layout(int pad_f4, int rotate); /* pad between packed tiles (16-byte units; < 0 keeps it) and the per-workgroup slice rotation: re-pack after changing */
int cfm_debug_set_conv2_bk(int bk);   /* K-tile of cfm_subsample_conv2_relu_f32: 16 (default) | 32; 0 / 1: K walked in storage order / channel-chunk-major (default); returns the previous K-tile */
/*      cfg + 16: bias epilogue; cfg + 32: swish epilogue; cfg + 64: K-tile 32 (a staged row = one whole 128-byte line;
 *      measured slower than the K-tile 16 loop at every hot-path shape but one: kept for tools/gemm_tune.py bk). */

/* Backward of the conv-subsampling stem (convolution.py:42-52).  With dz2 = relu'(h2) * dh2 (cfm_relu_bwd_f32):
 *   conv2_bwd_weight: dw2p (C, 9C) in the PACKED (co,kf,kt,ci) layout += dz2^T . im2col(h1)   (zero-filled by caller;
 *                     un-pack with the inverse of cfm_pack_conv2_weight_f32)
 *   pack_conv2_weight_t + conv2_bwd_input: dh1 (B,T1,F1,C) = conv-transpose(dz2, w2) as four parity-class implicit GEMMs
 *   conv1_bwd: dw1 (C,1,3,3), db1 (C) accumulated from dh1 (ReLU mask recomputed from x)                       */
int cfm_relu_bwd_f32(const float* y, const float* dy, float* dz, int64_t n, cfm_stream_t stream);
int cfm_subsample_conv2_bwd_weight_f32(const float* dz2, const float* h1, float* dw2p, int B, int F1, int T1,
                                       int C, cfm_stream_t stream);
int cfm_pack_conv2_weight_t_f32(const float* w2, float* w2c, int C, cfm_stream_t stream);
int cfm_subsample_conv2_bwd_input_f32(const float* dz2, const float* w2c, float* dh1, int B, int F1, int T1,
                                      int C, cfm_stream_t stream);
int cfm_subsample_conv1_bwd_f32(const float* x, const float* w1, const float* b1, const float* dh1, float* dw1,
                                float* db1, int B, int F, int T, int C, cfm_stream_t stream);

/* ---- "next" rows.  N2: one fused Adam step (torch.optim.Adam defaults, train.py:188) over many tensors:
 *      `tensors` is a HOST array of n_tensors descriptors holding DEVICE pointers; ceil(n/48) launches, the table rides
 *      in the kernel arguments.  bias_c1 = 1-beta1^t, sqrt_bias_c2 = sqrt(1-beta2^t) (t = step count after increment).
 *      N4: greedy CTC decode (processor.py:301-328): per-frame argmax (B,T) int64, decoded ids (B,T) padded with -1,
 *      counts (B); pad/unk frames are skipped WITHOUT resetting the repeat filter (reference behaviour). */
typedef struct cfm_adam_tensor {
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    int64_t numel;
} cfm_adam_tensor;
int cfm_adam_step_f32(const cfm_adam_tensor* tensors, int n_tensors, float lr, float beta1, float beta2, float eps,
                      float bias_c1, float sqrt_bias_c2, cfm_stream_t stream);
int cfm_greedy_ctc_decode_f32(const float* logits, const int64_t* lengths_or_null, int64_t* frame_ids,
                              int64_t* tokens, int64_t* counts, int B, int T, int V, int pad_id, int unk_id,
                              cfm_stream_t stream);

/* N1 decoder (decoder.py:10-27): LSTM recurrence over a packed batch.  gates_x (B,T,4H) = X.W_ih^T + b_ih + b_hh from
 *      one of the GEMM entries; w_hh (4H,H), gate order i|f|g|o; lengths_or_null: frames per utterance (outputs beyond are
 *      0, as pad_packed_sequence returns); y (B,T,H) <- h_t; c_state (B,H) scratch; save_* (B,T,4H)/(B,T,H) or NULL.
 *      H % 4 == 0.  cfm_lstm_bwd_f32: dy (B,T,H) -> dgates (B,T,4H) = gradient w.r.t. gates_x, from the saved gates /
 *      cells and whh_t (H,4H) = W_hh^T; dc_state (B,H) scratch.  cfm_swish_bn_eval_f32: out = BatchNorm1d(eval)(swish(h)) per channel, rows x C. */
int cfm_lstm_fwd_f32(const float* gates_x, const float* w_hh, const int64_t* lengths_or_null, float* y, float* c_state,
                     float* save_gates_or_null, float* save_c_or_null, int B, int T, int H, cfm_stream_t stream);
int cfm_lstm_bwd_f32(const float* dy, const float* gates, const float* cells, const float* whh_t,
                     const int64_t* lengths_or_null, float* dgates, float* dc_state, int B, int T, int H,
                     cfm_stream_t stream);
/* The same two entries with both operands of the recurrent product in MFMA-fragment order (a wave's 16-byte-per-lane load
 * is one contiguous 1 KB run; the step is bound by memory round trips).  H % 16 == 0.  Results are bit-identical to the
 * row-major entries.  w_hh_frag[ub][ch][kq][q][u][e] = W_hh[q*H + 4*ub + u][16*ch + 4*kq + e]  (ub < H/4, ch < H/16, kq,q,u,e < 4);
 * whh_t_frag[ub][ch][kq][u][e] = W_hh[16*ch + 4*kq + e][16*ub + u]  (ub < H/16, ch < 4H/16, u < 16);
 * h_frag_scratch: 2*ceil(B/16)*16*H floats; dg_frag_scratch: 2*ceil(B/16)*16*4H floats (neither needs initialising). */
int cfm_lstm_fwd_frag_f32(const float* gates_x, const float* w_hh_frag, const int64_t* lengths_or_null, float* y,
                          float* c_state, float* h_frag_scratch, float* save_gates_or_null, float* save_c_or_null, int B,
                          int T, int H, cfm_stream_t stream);
int cfm_lstm_bwd_frag_f32(const float* dy, const float* gates, const float* cells, const float* whh_t_frag,
                          const int64_t* lengths_or_null, float* dgates, float* dc_state, float* dg_frag_scratch, int B,
                          int T, int H, cfm_stream_t stream);
/* cfm_lstm_fwd_f32 with the recurrent product h.W_hh^T on the 16-bit matrix pipe -- what torch.autocast does to nn.LSTM
 * on a GPU (reference decoder.py:10,17-22 under train.py:232).  H % 16 == 0.  Operands in MFMA-fragment order:
 * w_hh16[ub][ch][hf][q][u][e] = W_hh[q*H + 8*ub + u][16*ch + 8*hf + e]  (ub < H/8, ch < H/16, hf < 2, q < 4, u,e < 8), a
 * 16-bit copy (cfm_cast16_f32 + re-order); h16_scratch: 2*ceil(B/32)*32*H 16-bit elements.  The backward form takes
 * whh_t16[ub][ch][hf][u][e] = W_hh[16*ch + 8*hf + e][16*ub + u]  (ub < H/16, ch < 4H/16, u < 16) and dg16_scratch:
 * 2*ceil(B/16)*16*4H 16-bit elements. */
int cfm_lstm_fwd_mfma16_f32(int prec, const float* gates_x, const void* w_hh16, const int64_t* lengths_or_null, float* y,
                            float* c_state, void* h16_scratch, float* save_gates_or_null, float* save_c_or_null, int B,
                            int T, int H, cfm_stream_t stream);
int cfm_lstm_bwd_mfma16_f32(int prec, const float* dy, const float* gates, const float* cells, const void* whh_t16,
                            const int64_t* lengths_or_null, float* dgates, float* dc_state, void* dg16_scratch, int B,
                            int T, int H, cfm_stream_t stream);
int cfm_swish_bn_eval_f32(const float* h, const float* bn_mean, const float* bn_var, const float* bn_weight,
                          const float* bn_bias, float eps, float* out, int64_t rows, int C, cfm_stream_t stream);
/*      train mode: cfm_swish_bn_stats_f32 = batch mean / biased variance of swish(h) (+ running update), then
 *      cfm_swish_bn_eval_f32 with those; cfm_swish_bn_bwd_f32: dh + accumulated dgamma, dbeta (caller zero-fills). */
int cfm_swish_bn_stats_f32(const float* h, float* batch_mean, float* batch_var, float* running_mean_or_null,
                           float* running_var_or_null, float momentum, int64_t rows, int C, cfm_stream_t stream);
int cfm_swish_bn_bwd_f32(const float* h, const float* dz, const float* bn_mean, const float* bn_var,
                         const float* bn_weight, float eps, int train_stats, float* dh, float* dgamma, float* dbeta,
                         int64_t rows, int C, cfm_stream_t stream);

/* ---- opt-in: fp32 GEMMs on the bf16 matrix pipe by exact operand splitting (csrc/gemm_split.hip).  Same contract as the
 *      fp32 entries above (fp32 A, bias, R, C; fp32 accumulation): every fp32 operand is expanded into `planes` bf16 terms
 *      (x = x0 + x1 + x2 exactly for planes = 3) and the products of total order < planes are accumulated -- planes = 3:
 *      six bf16 MFMAs per K-step, per-product relative error <= 2^-23 (fp32 class); planes = 2: three, <= 2^-15.
 *      W_split: the expansion of the (N,K) fp32 weight in MFMA fragment order ([row block of 32][K-step of 16][plane]
 *      [lane][8 bf16]; cfm_split_pack_elems bf16 elements), made once per weight version by cfm_split_pack_bf16_f32
 *      (epi 3: all 2*n_out rows, n_out % 32 == 0).  K % 16 == 0.  epi as cfm_gemm_mfma16_f32. */
int64_t cfm_split_pack_elems(int planes, int N, int K);
int cfm_split_pack_bf16_f32(int planes, const float* w, void* dst, int N, int K, cfm_stream_t stream);
int cfm_gemm_split_bf16_f32(int planes, int epi, const float* A, const void* W_split, const float* bias,
                            const float* R_or_null, float alpha, float* C, int64_t M, int N, int K, int64_t lda,
                            int64_t ldr, int64_t ldc, cfm_stream_t stream);
int cfm_subsample_conv2_relu_split_bf16_f32(int planes, const float* h1, const void* w2p_split, const float* b2, float* h2,
                                            int B, int F1, int T1, int C, cfm_stream_t stream);

/* N1, loss half: ConformerCriterion.ctc_loss (evaluation.py:12-16) = nn.CTCLoss(blank, reduction='mean',
 *      zero_infinity=True) over log_softmax(logits), the log-softmax folded in.  logits (B,T,V) fp32 (batch-first, as
 *      the model returns them: the transpose of evaluation.py:16 is index arithmetic); targets int64, label i of
 *      utterance b at targets[off_b + i], off_b = tgt_off_or_null ? tgt_off[b] : b*tgt_stride (2-D padded or 1-D
 *      concatenated targets), tgt_numel = elements in `targets`; in_len / tgt_len int64 (B) on the device (clamped to
 *      T / Lmax); Lmax <= CFM_CTC_MAX_TARGET bounds every tgt_len.  workspace: cfm_ctc_workspace_floats(B,T,Lmax)
 *      floats, written by fwd and consumed by bwd (same arguments).  loss: 1 float.  bwd: dlogits (B,T,V) =
 *      grad_out[0] * d loss / d logits (frames >= in_len and utterances with infinite loss get zeros). */
#define CFM_CTC_MAX_TARGET 1023
int64_t cfm_ctc_workspace_floats(int B, int T, int Lmax);
int cfm_ctc_loss_fwd_f32(const float* logits, const int64_t* targets, const int64_t* tgt_off_or_null,
                         int64_t tgt_stride, int64_t tgt_numel, const int64_t* in_len, const int64_t* tgt_len,
                         int B, int T, int V, int Lmax, int blank, float* workspace, float* loss, cfm_stream_t stream);
int cfm_ctc_loss_bwd_f32(const float* logits, const int64_t* targets, const int64_t* tgt_off_or_null,
                         int64_t tgt_stride, int64_t tgt_numel, const int64_t* in_len, const int64_t* tgt_len,
                         int B, int T, int V, int Lmax, int blank, float* workspace, const float* grad_out,
                         float* dlogits, cfm_stream_t stream);

/* incremental / streaming attention (no reference counterpart: the reference has no streaming code; BASELINE cfg-5): as
 *      cfm_relpos_attention_fwd_f32 but only the query rows [q_begin, q_begin+q_count) are computed; q/k/v/ctx are the
 *      whole (B,T,.) buffers (a K/V cache that grows in place), lengths = keys visible so far.  nsplit in [1,16]: > 1
 *      divides the key tiles over nsplit workgroups per (b, h, 128-row block) and merges their partial results
 *      (workspace: nsplit*B*q_count*(H*dh + H) floats; needs ldo == H*dh, lengths != NULL with every length >= 1). */
int cfm_relpos_attention_rows_f32(const float* q, const float* k, const float* v, int64_t ld, const float* pos,
                                  int64_t ldp, const float* u, const float* vbias, const int64_t* lengths_or_null,
                                  float* ctx, int64_t ldo, int B, int T, int H, int dh, int q_begin, int q_count,
                                  int nsplit, float* workspace_or_null, cfm_stream_t stream);

/* diagnostics only: cfm_relpos_attention_fwd_f32 + s_memrealtime stamps of one wave (trace: 16*ceil(T/32) uint64) */
int cfm_debug_attention_trace_f32(const float* q, const float* k, const float* v, int64_t ld, const float* pos,
                                  int64_t ldp, const float* u, const float* vbias, const int64_t* lengths_or_null,
                                  float* ctx, int64_t ldo, int B, int T, int H, int dh, void* trace, cfm_stream_t stream);
/*      force the workgroup shape of the fp32 attention forward: 0 = built-in choice (8 waves = 256 query rows with the two
 *      halves half a key tile apart once a launch has > 128 query rows; else 4 waves) | 4 | 8; returns the previous setting.
 *      The two shapes agree to fp32 rounding. */
int cfm_debug_set_attention_waves(int nw);
/* diagnostics only (process-global, not thread-safe): force the block tile of cfm_gemm_bwd* (-1 = heuristic) */
int cfm_debug_set_bwd_tile(int tile);

/* ---- integer helpers of the path (host-side, no device work) ---------------------------------
 *      frames after the stem: ((n-1)/2-1)/2, convolution.py:55 */
int64_t cfm_subsampled_length(int64_t n);
/*      the same applied to a device array of B lengths (floor division, negative values included), one launch */
int cfm_subsampled_lengths_i64(const int64_t* lengths, int64_t* out, int n, cfm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CONFORMER_HIP_H */
