"""ConformerBlock (surface of model/utils/block.py:8-29): the Macaron sandwich on gfx950 kernels.

Each of the four residual sub-layers ends in an MFMA GEMM whose epilogue applies `alpha*y + x`
(alpha = 1/2 for the two FFNs, block.py:19,25), so no stand-alone add/scale kernels run; the closing
LayerNorm (block.py:27) is the wave-per-row kernel.  16 launches per block in total; 12 on the folded-LayerNorm
inference path (fused_chain), 9 with the one-kernel feed-forward sub-layers (ops.ffn_fused: FFN2 also applies the closing
LayerNorm).
"""
from typing import Optional

import torch
import torch.nn as nn

from ... import autograd as ag
from ... import ops
from .attention import MultiHeadSelfAttentionModule
from .convolution import ConvolutionModule
from .ffn import FeedForwardModule
from ._guard import refuse_dropout
from .masking import lengths_from_key_padding_mask


class ConformerBlock(nn.Module):
    def __init__(self, d_model: int, n_heads: int, kernel_size: int, dropout_rate: float = 0.0) -> None:
        super().__init__()
        self.ffn_1 = FeedForwardModule(dim=d_model, dropout_rate=dropout_rate)
        self.attention = MultiHeadSelfAttentionModule(n_heads=n_heads, d_model=d_model, dropout_rate=dropout_rate)
        self.conv = ConvolutionModule(channels=d_model, kernel_size=kernel_size, dropout_rate=dropout_rate)
        self.ffn_2 = FeedForwardModule(dim=d_model, dropout_rate=dropout_rate)
        self.layer_norm = nn.LayerNorm(normalized_shape=d_model)

    def fused(self, x: torch.Tensor, pos_table: torch.Tensor, lengths: Optional[torch.Tensor],
              pos_projected: Optional[torch.Tensor] = None) -> torch.Tensor:
        return self.fused_chain(x, pos_table, lengths, pos_projected)[0]

    def _ln_fold(self, x: torch.Tensor) -> bool:
        """The folded-LayerNorm path (ops.ln_fold_ok): fp32 inference with eval-mode BatchNorm and no gradients."""
        bn = self.conv.batch_norm
        return (x.dtype == torch.float32 and ops.ln_fold_ok(self.layer_norm.normalized_shape[0])
                and not (bn.training or bn.running_mean is None) and not ag.needs_grad(self, x))

    def fused_chain(self, x: torch.Tensor, pos_table: torch.Tensor, lengths: Optional[torch.Tensor],
                    pos_projected: Optional[torch.Tensor] = None, x_stats: Optional[torch.Tensor] = None,
                    want_stats: bool = False):
        """Returns (block output, LayerNorm statistics partials of its rows or None).  On the folded path (fp32 inference) every
        residual GEMM's epilogue emits the statistics of the rows it stores and the next sub-layer's first GEMM applies the
        LayerNorm from them (ops.linear_lnfold): the LayerNorms of ffn.py:16, attention.py:15 and convolution.py:22 are not
        launched; the closing LayerNorm (block.py:27) runs and hands the statistics of ITS output to the next block
        (x_stats / want_stats).  12 launches per block instead of 16."""
        if not self._ln_fold(x):
            y = self.ffn_1.fused(x, residual=x, alpha=0.5)
            y = self.attention.fused(y, pos_table, lengths, residual=y, pos_projected=pos_projected)
            y = self.conv.fused(y, residual=y)
            y = self.ffn_2.fused(y, residual=y, alpha=0.5)
            if ag.needs_grad(self.layer_norm, y):
                return ag.LayerNormFn.apply(y, self.layer_norm.weight, self.layer_norm.bias, self.layer_norm.eps), None
            return ops.layernorm(y, self.layer_norm.weight, self.layer_norm.bias, self.layer_norm.eps), None
        if x_stats is not None and x.is_contiguous() and ops.rowchain_ok(x.shape[-1], self.ffn_1.hidden_linear.out_features,
                                                                          x.numel() // x.shape[-1]):
            return self._row_chains(x, pos_table, lengths, pos_projected, x_stats, want_stats)
        y, st = self.ffn_1.fused(x, residual=x, alpha=0.5, stats=x_stats, emit_stats=True)
        y, st = self.attention.fused(y, pos_table, lengths, residual=y, pos_projected=pos_projected, stats=st, emit_stats=True)
        y, st = self.conv.fused(y, residual=y, stats=st, emit_stats=True)
        ln = self.layer_norm
        if self.ffn_2.fuses(y, y, st):                  # FFN2 + the closing LayerNorm in the one-kernel feed-forward
            if want_stats:
                return self.ffn_2.fused(y, residual=y, alpha=0.5, stats=st, emit_stats=True, closing_ln=ln)
            return self.ffn_2.fused(y, residual=y, alpha=0.5, stats=st, closing_ln=ln), None
        y = self.ffn_2.fused(y, residual=y, alpha=0.5, stats=st)
        if want_stats:
            return ops.layernorm(y, ln.weight, ln.bias, ln.eps, emit_stats=True)
        return ops.layernorm(y, ln.weight, ln.bias, ln.eps), None

    def _row_chains(self, x, pos_table, lengths, pos_projected, x_stats, want_stats):
        """The block as 5 launches (ops.rowchain_*): K1 = FFN1 + q|k|v projection, attention, K2 = out_proj + pointwise_conv_1 + GLU,
        depthwise conv + BatchNorm + Swish, K3 = pointwise_conv_2 + FFN2 + closing LayerNorm -- every row-local stretch of
        block.py:17-29 in one kernel per 32 rows (csrc/rowchain_f32.hip)."""
        f1, f2, att, cv = self.ffn_1, self.ffn_2, self.attention, self.conv
        a, ln_a, ln_c, bn = att.attention, att.layer_norm, cv.layer_norm, cv.batch_norm
        for m, name in ((f1, "FeedForwardModule"), (att, "MultiHeadSelfAttentionModule"), (a, "RelativeMultiHeadAttention"),
                        (cv, "ConvolutionModule"), (f2, "FeedForwardModule")):
            refuse_dropout(m, name)
        ffn1 = f1._packs.get("ffn_pack", (f1.hidden_linear.weight, f1.hidden_linear.bias, f1.layer_norm.weight, f1.layer_norm.bias,
                                          f1.out_linear.weight), f1._pack)
        ffn2 = f2._packs.get("ffn_pack", (f2.hidden_linear.weight, f2.hidden_linear.bias, f2.layer_norm.weight, f2.layer_norm.bias,
                                          f2.out_linear.weight), f2._pack)
        wqkv, bqkv = a._qkv_params()

        def fold_pack(w, b, ln, glu=False):
            wf, bf, cs = ops.fold_layernorm(w, b, ln.weight, ln.bias)
            return ops.rowgemm_pack(wf, glu=glu), bf, cs

        wq_p, bq_f, csq = a._packs.get("qkv_chain", (wqkv, bqkv, ln_a.weight, ln_a.bias), lambda: fold_pack(wqkv, bqkv, ln_a))
        wo_p = a._packs.get("out_chain", (a.out_proj.weight,), lambda: ops.rowgemm_pack(a.out_proj.weight.detach()))
        pw1, pw2 = cv.pointwise_conv_1, cv.pointwise_conv_2
        wg_p, bg_f, csg = cv._packs.get("pw1_chain", (pw1.weight, pw1.bias, ln_c.weight, ln_c.bias),
                                        lambda: fold_pack(pw1.weight, pw1.bias, ln_c, glu=True))
        w2_p = cv._packs.get("pw2_chain", (pw2.weight,), lambda: ops.rowgemm_pack(pw2.weight.detach()))
        y, qkv = ops.rowchain_ffn_qkv(x, x_stats, ffn1, f1.out_linear.bias, 0.5, f1.layer_norm.eps, wq_p, bq_f, csq, ln_a.eps)
        pos = pos_projected if pos_projected is not None else ops.linear(pos_table, a.pos_proj.weight, a.pos_proj.bias)
        ctx = ops.relpos_attention(qkv, pos, a.content_bias, a.position_bias, lengths, a.n_heads)
        y2, g = ops.rowchain_out_glu(ctx, wo_p, a.out_proj.bias, y, wg_p, bg_f, csg, ln_c.eps)
        c = ops.dwconv_bn_swish(g, cv.deepwise_conv.weight, cv.deepwise_conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                bn.eps)
        ln = self.layer_norm
        return ops.rowchain_pw2_ffn_ln(c, w2_p, pw2.bias, y2, ffn2, f2.out_linear.bias, 0.5, f2.layer_norm.eps,
                                       (ln.weight, ln.bias, ln.eps), want_stats)

    def forward(self, x: torch.Tensor, pos_embedding: torch.Tensor, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        table = pos_embedding[0] if pos_embedding.dim() == 3 else pos_embedding
        lengths = None if mask is None else lengths_from_key_padding_mask(mask)
        return self.fused(x, table, lengths)
