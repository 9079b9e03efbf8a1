"""FeedForwardModule (surface of model/utils/ffn.py:5-23) on gfx950 kernels.

LN -> Linear(d,4d)+Swish (one MFMA GEMM, Swish in the epilogue) -> Linear(4d,d) (+ optional fused
`alpha*y + residual` epilogue used by ConformerBlock, block.py:19,25).  Sub-module names/shapes match the
reference state_dict.
"""
from typing import Optional

import torch
import torch.nn as nn

from ... import autograd as ag
from ... import ops
from ._guard import PackCache, active_dropout, refuse_dropout
from .activation import Swish


class FeedForwardModule(nn.Module):
    def __init__(self, dim: int, dropout_rate: float = 0.0) -> None:
        super().__init__()
        self.layer_norm = nn.LayerNorm(normalized_shape=dim)
        self.hidden_linear = nn.Linear(in_features=dim, out_features=4 * dim)
        self.swish = Swish()
        self.dropout_1 = nn.Dropout(p=dropout_rate)
        self.out_linear = nn.Linear(in_features=4 * dim, out_features=dim)
        self.dropout_2 = nn.Dropout(p=dropout_rate)
        self._packs = PackCache()

    def fused(self, x: torch.Tensor, residual: Optional[torch.Tensor] = None, alpha: float = 1.0,
              stats: Optional[torch.Tensor] = None, emit_stats: bool = False, closing_ln: Optional[nn.LayerNorm] = None):
        """stats / emit_stats (fp32 inference, ops.ln_fold_ok): `stats` are the LayerNorm statistics partials of the rows of x
        written by x's producer -- the LayerNorm of ffn.py:16 is then folded into the hidden GEMM (no LayerNorm launch);
        emit_stats returns (y, stats of y) from the residual GEMM's epilogue for the next sub-layer's LayerNorm.
        With residual = x and enough rows to fill the chip the whole sub-layer is ONE kernel (ops.ffn_fused: the hidden
        activation stays on chip); closing_ln (block.py:27) is then applied by the same kernel (callers check
        `fuses(x, residual, stats)` before passing it)."""
        refuse_dropout(self, "FeedForwardModule")
        if self.fuses(x, residual, stats):
            ln, lin, out = self.layer_norm, self.hidden_linear, self.out_linear
            wp, bf, cs = self._packs.get("ffn_pack", (lin.weight, lin.bias, ln.weight, ln.bias, out.weight), lambda: self._pack())
            cl = None if closing_ln is None else (closing_ln.weight, closing_ln.bias, closing_ln.eps)
            return ops.ffn_fused(x, stats, wp, bf, cs, out.bias, alpha, ln.eps, emit_stats=emit_stats, closing_ln=cl)
        if closing_ln is not None:
            raise NotImplementedError("FeedForwardModule.fused: closing_ln needs the one-kernel path (check fuses())")
        if ag.needs_grad(self, x, residual):
            if residual is not None and residual is not x:
                raise NotImplementedError("FeedForwardModule.fused: the differentiable path folds the residual of its "
                                          "own input (block.py:19,25); pass residual=x")
            out = ag.FeedForwardFn.apply(x, self.layer_norm.weight, self.layer_norm.bias, self.hidden_linear.weight,
                                         self.hidden_linear.bias, self.out_linear.weight, self.out_linear.bias,
                                         float(alpha), self.layer_norm.eps, active_dropout(self.dropout_1))
            # the Function always folds `+ x`; a stand-alone call (no residual) removes it again
            return out if residual is not None else out - x
        if stats is not None:
            ln, lin = self.layer_norm, self.hidden_linear
            wf, bf, cs = self._packs.get("ln_fold", (lin.weight, lin.bias, ln.weight, ln.bias),
                                         lambda: ops.fold_layernorm(lin.weight, lin.bias, ln.weight, ln.bias))
            h = ops.linear_lnfold(x, stats, wf, bf, cs, ln.eps, act="swish")
        else:
            h = ops.layernorm(x, self.layer_norm.weight, self.layer_norm.bias, self.layer_norm.eps, for_gemm=True)
            h = ops.linear(h, self.hidden_linear.weight, self.hidden_linear.bias, act="swish", for_gemm=True)
        if emit_stats and ops._splitk(h.numel() // h.shape[-1], self.out_linear.out_features, self.out_linear.in_features):
            # small M (a streaming chunk): the split-K form of this long-K product is worth more than the statistics its epilogue
            # could emit -- the next sub-layer then runs its LayerNorm kernel (stats = None)
            o = ops.linear(h, self.out_linear.weight, self.out_linear.bias) if residual is None else \
                ops.linear_residual(h, self.out_linear.weight, self.out_linear.bias, residual, alpha)
            return o, None
        if residual is None:
            if emit_stats:
                return ops.linear(h, self.out_linear.weight, self.out_linear.bias, emit_stats=True)
            return ops.linear(h, self.out_linear.weight, self.out_linear.bias)
        return ops.linear_residual(h, self.out_linear.weight, self.out_linear.bias, residual, alpha, emit_stats=emit_stats)

    def fuses(self, x: torch.Tensor, residual: Optional[torch.Tensor], stats: Optional[torch.Tensor]) -> bool:
        """The one-kernel form applies: folded-LayerNorm inference (stats given), the residual is the input itself, and the
        rows fill the chip (ops.ffn_fused_ok)."""
        d = x.shape[-1]
        return (stats is not None and residual is x and x.dtype == torch.float32 and x.is_contiguous()
                and not ag.needs_grad(self, x)
                and ops.ffn_fused_ok(d, self.hidden_linear.out_features, x.numel() // d))

    def _pack(self):
        ln, lin = self.layer_norm, self.hidden_linear
        wf, bf, cs = ops.fold_layernorm(lin.weight, lin.bias, ln.weight, ln.bias)
        return ops.ffn_pack(wf, self.out_linear.weight.detach()), bf, cs

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.fused(x)
