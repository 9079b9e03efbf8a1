"""RelativePositionalEncoding (surface of model/utils/position.py:5-27).

`div_term` stays an nn.Parameter(requires_grad=False) so it appears in the state_dict exactly as in the
reference (position.py:9).  The hot path uses `table(T)`: one (2T-1, d) sinusoid table built by
cfm_relpos_table_f32 and cached per T; `forward` reproduces the reference's batch-repeated (B, 2T-1, d)
output for API parity only.
"""
import math

import torch
import torch.nn as nn

from ... import ops


class RelativePositionalEncoding(nn.Module):
    def __init__(self, d_model: int) -> None:
        super().__init__()
        self.d_model = d_model
        freq = torch.exp(torch.arange(0, d_model, 2) * -(math.log(10000.0) / d_model))
        self.div_term = nn.Parameter(freq.unsqueeze(0), requires_grad=False)
        self._cache = {}

    def table(self, n_frames: int) -> torch.Tensor:
        key = (n_frames, self.div_term.data_ptr(), self.div_term._version)
        hit = self._cache.get("t")
        if hit is None or hit[0] != key:
            hit = (key, ops.relpos_table(self.div_term, n_frames))
            self._cache["t"] = hit
        return hit[1]

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.table(x.size(1)).unsqueeze(0).expand(x.size(0), -1, -1).contiguous()
