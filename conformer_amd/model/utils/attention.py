"""Relative-position MHSA (surface of model/utils/attention.py:7-102) on gfx950 kernels.

Differences in HOW (not in results):
  * Q, K, V are one fused (d -> 3d) MFMA GEMM; the three nn.Linear parameters stay separate in the
    state_dict and are concatenated into a cached device buffer;
  * pos_proj is applied ONCE to the (2T-1, d) table instead of to a batch-repeated (B, 2T-1, d) tensor
    (attention.py:81, position.py:26);
  * content scores, positional scores, the relative shift (attention.py:94-102), masking, softmax and
    attention.V run inside one flash-style kernel (attention_f32.hip) -- no (B,H,T,T) tensor exists;
  * the key-padding mask is consumed as `lengths`.
"""
from typing import Optional

import torch
import torch.nn as nn

from ... import autograd as ag
from ... import ops
from ._guard import PackCache, active_dropout, refuse_dropout
from .masking import lengths_from_key_padding_mask


class RelativeMultiHeadAttention(nn.Module):
    def __init__(self, d_model: int, n_heads: int, dropout_rate: float = 0.0):
        super().__init__()
        if d_model % n_heads != 0:
            raise AssertionError("d_model must be divisible by n_heads")
        self.d_model = d_model
        self.n_heads = n_heads
        self.head_samples = d_model // n_heads
        self.sqrt_dim = float(self.head_samples) ** 0.5

        self.query_proj = nn.Linear(d_model, d_model)
        self.key_proj = nn.Linear(d_model, d_model)
        self.value_proj = nn.Linear(d_model, d_model)
        self.pos_proj = nn.Linear(d_model, d_model)
        self.dropout = nn.Dropout(p=dropout_rate)
        self.content_bias = nn.Parameter(torch.empty(n_heads, self.head_samples))
        self.position_bias = nn.Parameter(torch.empty(n_heads, self.head_samples))
        self.out_proj = nn.Linear(d_model, d_model)
        self.mask_value = None          # kept for attribute parity; the fused kernel skips masked keys
        nn.init.xavier_uniform_(self.content_bias)
        nn.init.xavier_uniform_(self.position_bias)
        self._packs = PackCache()

    # ---- fused path -------------------------------------------------------------------------
    def _qkv_params(self):
        ws = (self.query_proj.weight, self.key_proj.weight, self.value_proj.weight)
        bs = (self.query_proj.bias, self.key_proj.bias, self.value_proj.bias)
        w = self._packs.get("qkv_w", ws, lambda: torch.cat([t.detach() for t in ws], dim=0).contiguous())
        b = self._packs.get("qkv_b", bs, lambda: torch.cat([t.detach() for t in bs], dim=0).contiguous())
        return w, b

    def context(self, x: torch.Tensor, pos_table: torch.Tensor, lengths: Optional[torch.Tensor],
                pos_projected: Optional[torch.Tensor] = None, for_gemm: bool = False, ln_fold=None) -> torch.Tensor:
        """x: (B,T,d) already layer-normed; pos_table: (2T-1,d) un-projected; returns concat-head context.
        `pos_projected` (a (2T-1,d) view, any row stride) is this layer's slice of the encoder-wide batched
        pos_proj GEMM (Encoder._projected_positions); without it the projection runs here."""
        w, b = self._qkv_params()
        # inference under autocast: the projections are written in the 16-bit type (what torch.autocast's nn.Linear returns; the
        # attention core rounds its operands to that type anyway -- q after the bias add, as the reference does)
        if ln_fold is not None:
            # x is NOT layer-normed: (stats, LayerNorm module) -- the LayerNorm of attention.py:15 folded into the fused q|k|v GEMM
            stats, ln = ln_fold
            wf, bf, cs = self._packs.get("qkv_ln_fold", (w, b, ln.weight, ln.bias),
                                         lambda: ops.fold_layernorm(w, b, ln.weight, ln.bias))
            qkv = ops.linear_lnfold(x, stats, wf, bf, cs, ln.eps)
        else:
            qkv = ops.linear(x, w, b, for_gemm=for_gemm)
        pos = pos_projected if pos_projected is not None else \
            ops.linear(pos_table, self.pos_proj.weight, self.pos_proj.bias)
        return ops.relpos_attention(qkv, pos, self.content_bias, self.position_bias, lengths, self.n_heads, for_gemm=for_gemm)

    def fused(self, x, pos_table, lengths, residual: Optional[torch.Tensor] = None,
              pos_projected: Optional[torch.Tensor] = None, ln_fold=None, emit_stats: bool = False):
        ctx = self.context(x, pos_table, lengths, pos_projected, for_gemm=True, ln_fold=ln_fold)   # (the context only feeds out_proj)
        if residual is None:
            if emit_stats:
                return ops.linear(ctx, self.out_proj.weight, self.out_proj.bias, emit_stats=True)
            return ops.linear(ctx, self.out_proj.weight, self.out_proj.bias)
        return ops.linear_residual(ctx, self.out_proj.weight, self.out_proj.bias, residual, 1.0, emit_stats=emit_stats)

    # ---- reference-compatible entry (attention.py:74) ------------------------------------------
    def forward(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, pos_embedding: torch.Tensor,
                mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        if not (k is q and v is q):
            raise NotImplementedError("the fused gfx950 kernel implements SELF-attention (q is k is v), which is "
                                      "the only way the reference calls it (attention.py:16)")
        table = pos_embedding[0] if pos_embedding.dim() == 3 else pos_embedding
        lengths = None if mask is None else lengths_from_key_padding_mask(mask)
        if ag.needs_grad(self, q, table):
            # differentiable path of the bare module (the blocks go through MultiHeadSelfAttentionModule, which folds the
            # LayerNorm and the residual into SelfAttentionFn): projections -> attention core -> out_proj, each an autograd
            # Function on the gfx950 kernels; the weight concatenation is ordinary autograd
            wqkv = torch.cat([self.query_proj.weight, self.key_proj.weight, self.value_proj.weight], dim=0)
            bqkv = torch.cat([self.query_proj.bias, self.key_proj.bias, self.value_proj.bias], dim=0)
            qkv = ag.LinearFn.apply(q, wqkv, bqkv)
            pos = ag.LinearFn.apply(table, self.pos_proj.weight, self.pos_proj.bias)
            att = ag.RelPosAttentionFn.apply(qkv, pos, self.content_bias, self.position_bias, lengths, self.n_heads,
                                             active_dropout(self.dropout))
            return ag.LinearFn.apply(att, self.out_proj.weight, self.out_proj.bias)
        refuse_dropout(self, "RelativeMultiHeadAttention")
        return self.fused(q, table, lengths)


class MultiHeadSelfAttentionModule(nn.Module):
    def __init__(self, d_model: int, n_heads: int, dropout_rate: float = 0.0) -> None:
        super().__init__()
        self.layer_norm = nn.LayerNorm(normalized_shape=d_model)
        self.attention = RelativeMultiHeadAttention(d_model=d_model, n_heads=n_heads, dropout_rate=dropout_rate)
        self.dropout = nn.Dropout(p=dropout_rate)

    def fused(self, x, pos_table, lengths, residual: Optional[torch.Tensor] = None,
              pos_projected: Optional[torch.Tensor] = None, stats: Optional[torch.Tensor] = None, emit_stats: bool = False):
        """stats / emit_stats: see FeedForwardModule.fused (the LayerNorm of attention.py:15 folds into the q|k|v GEMM)."""
        refuse_dropout(self, "MultiHeadSelfAttentionModule")
        if ag.needs_grad(self, x, pos_projected):
            a = self.attention
            if residual is not None and residual is not x:
                raise NotImplementedError("MultiHeadSelfAttentionModule.fused: pass residual=x (block.py:21)")
            if pos_projected is None:
                pos_projected = ag.LinearFn.apply(pos_table, a.pos_proj.weight, a.pos_proj.bias)
            out = ag.SelfAttentionFn.apply(x, self.layer_norm.weight, self.layer_norm.bias, a.query_proj.weight,
                                           a.query_proj.bias, a.key_proj.weight, a.key_proj.bias, a.value_proj.weight,
                                           a.value_proj.bias, pos_projected, a.content_bias, a.position_bias,
                                           a.out_proj.weight, a.out_proj.bias, lengths, a.n_heads, self.layer_norm.eps,
                                           active_dropout(self.dropout))
            return out if residual is not None else out - x
        if stats is not None:
            return self.attention.fused(x, pos_table, lengths, residual, pos_projected, ln_fold=(stats, self.layer_norm),
                                        emit_stats=emit_stats)
        xn = ops.layernorm(x, self.layer_norm.weight, self.layer_norm.bias, self.layer_norm.eps, for_gemm=True)
        return self.attention.fused(xn, pos_table, lengths, residual, pos_projected, emit_stats=emit_stats)

    def forward(self, x: torch.Tensor, pos_embedding: torch.Tensor, mask: Optional[torch.Tensor] = None):
        table = pos_embedding[0] if pos_embedding.dim() == 3 else pos_embedding
        lengths = None if mask is None else lengths_from_key_padding_mask(mask)
        return self.fused(x, table, lengths)
