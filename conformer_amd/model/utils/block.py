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

    def forward(self, x: torch.Tensor, pos_embedding: torch.Tensor, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        table = pos_embedding[0] if pos_embedding.dim() == 3 else pos_embedding
        lengths = None if mask is None else lengths_from_key_padding_mask(mask)
        return self.fused(x, table, lengths)
