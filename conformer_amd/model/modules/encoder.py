"""Encoder (surface of model/modules/encoder.py:9-37): stem -> input Linear -> N Conformer blocks, all on
gfx950 kernels.  `forward(x (B,n_mel,T), lengths (B,) int64 | None) -> (y (B,T',d), lengths')`.

`lengths` stays on the device and is handed to the attention kernel; T' comes from the tensor shape, so
the two host syncs of the reference (masking.py:10) are gone.  Setting CONFORMER_AMD_STRICT=1 restores the
reference's failure when lengths.max() != T' (masking.py:9-12 + encoder.py:30) at the cost of one sync.
"""
from typing import Optional, Tuple

import torch
import torch.nn as nn

from ... import autograd as ag
from ... import ops
from ..utils._guard import STRICT, PackCache, active_dropout, refuse_dropout
from ..utils.block import ConformerBlock
from ..utils.convolution import ConvolutionSubsampling
from ..utils.position import RelativePositionalEncoding


class Encoder(nn.Module):
    def __init__(self, n_mel_channels: int, n_blocks: int, d_model: int, n_heads: int, kernel_size: int,
                 dropout_rate: float = 0.0) -> None:
        super().__init__()
        self.n_freq_out = ((n_mel_channels - 1) // 2 - 1) // 2
        self.downsampling_conv = ConvolutionSubsampling(channels=d_model)
        self.linear = nn.Linear(in_features=d_model * self.n_freq_out, out_features=d_model)
        self.dropout = nn.Dropout(p=dropout_rate)
        self.rel_pe = RelativePositionalEncoding(d_model=d_model)
        self.layers = nn.ModuleList([ConformerBlock(d_model=d_model, n_heads=n_heads, kernel_size=kernel_size,
                                                    dropout_rate=dropout_rate) for _ in range(n_blocks)])
        self._packs = PackCache()
        self.cache_projected_positions = False

    def forward(self, x: torch.Tensor, lengths: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        refuse_dropout(self, "Encoder")
        d = self.linear.out_features
        h = self.downsampling_conv.channel_last(x)                                   # (B, T', F'*C)
        st = None
        out_len = ConvolutionSubsampling.out_lengths(lengths)
        if ag.needs_grad(self.linear):
            # differentiable re-layout of the weight (columns c*F'+f -> f*C+c) so .grad lands in the reference layout
            wlp = self.linear.weight.view(d, d, self.n_freq_out).transpose(1, 2).reshape(d, -1)
            h = ag.LinearFn.apply(h, wlp, self.linear.bias, active_dropout(self.dropout))     # encoder.py:23-25
        else:
            wlp = self._packs.get("wlp", (self.linear.weight,),
                                  lambda: ops.pack_linear_weight(self.linear.weight, d, self.n_freq_out))
            fold = ops.ln_fold_ok(d) and len(self.layers) > 0 and h.dtype == torch.float32 and self.layers[0]._ln_fold(h)
            if fold:      # the statistics of the rows the input Linear stores feed the first block's folded LayerNorm
                h, st = ops.linear(h, wlp, self.linear.bias, emit_stats=True)
            else:
                h = ops.linear(h, wlp, self.linear.bias)                             # (B, T', d)
        n_frames = h.shape[1]
        if out_len is not None:
            if out_len.device != h.device:
                out_len = out_len.to(h.device)
            if STRICT and int(out_len.max()) != n_frames:
                raise RuntimeError(f"lengths.max() after subsampling must equal T'={n_frames}")
        table = self.rel_pe.table(n_frames)
        pos_all = self._projected_positions(table)
        for i, layer in enumerate(self.layers):
            h, st = layer.fused_chain(h, table, out_len, None if pos_all is None else pos_all[:, i * d:(i + 1) * d],
                                      x_stats=st, want_stats=i + 1 < len(self.layers))
        return h, out_len

    def _projected_positions(self, table: torch.Tensor) -> Optional[torch.Tensor]:
        """pos_proj of every layer applied to the (2T'-1, d) table in ONE MFMA GEMM: (2T'-1, d) x (L*d, d)^T.
        Layer i reads columns [i*d, (i+1)*d) through the attention kernel's `ldp` row stride -- no per-layer pos
        GEMM, no copies.  The result depends only on T' and the pos_proj parameters; it is nevertheless recomputed
        on every forward (the weight concatenation alone is cached) unless `cache_projected_positions` is set, which
        an inference service with frozen weights may do (then it is keyed on the parameters' identity/version)."""
        if len(self.layers) == 0:
            return None
        ws = [l.attention.attention.pos_proj.weight for l in self.layers]
        bs = [l.attention.attention.pos_proj.bias for l in self.layers]

        if ag.needs_grad(self.layers):
            return ag.LinearFn.apply(table, torch.cat(ws, dim=0), torch.cat(bs, dim=0))   # cat is differentiable glue
        w = self._packs.get("pos_w", ws, lambda: torch.cat([t.detach() for t in ws], dim=0).contiguous())
        b = self._packs.get("pos_b", bs, lambda: torch.cat([t.detach() for t in bs], dim=0).contiguous())
        if self.cache_projected_positions:
            return self._packs.get("pos_all", (w, b, table), lambda: ops.linear(table, w, b))
        return ops.linear(table, w, b)
