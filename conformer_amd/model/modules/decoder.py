"""Decoder (surface of model/modules/decoder.py:7-27): packed LSTM -> Swish -> BatchNorm1d -> Linear.

SURVEY 8f row N1.  Inference (eval mode, no gradient wanted) runs on gfx950 kernels: one GEMM for the input projection,
the LSTM recurrence kernel (csrc/lstm.hip), a fused Swish+BatchNorm(eval) pass and the vocabulary GEMM -- no host
round-trip for `lengths` (the reference's `lengths.cpu()` at decoder.py:17 synchronises the stream every call).
Training (gradients through the LSTM, train-mode BatchNorm) stays on the stock PyTorch-ROCm modules this round: the
decoder is outside the round-1 hot path and its backward kernels are not built.  The state_dict keys are the reference's.
"""
from typing import Optional

import torch
import torch.nn as nn

from ... import ops
from ..utils._guard import PackCache
from ..utils.activation import Swish


class Decoder(nn.Module):
    def __init__(self, vocab_size: int, d_model: int, hidden_dim: int, n_layers: int) -> None:
        super().__init__()
        self.lstm = nn.LSTM(input_size=d_model, hidden_size=hidden_dim, num_layers=n_layers, batch_first=True)
        self.activation = Swish()
        self.norm = nn.BatchNorm1d(num_features=hidden_dim)
        self.linear = nn.Linear(in_features=hidden_dim, out_features=vocab_size)
        self._packs = PackCache()

    def _hip_eligible(self, x: torch.Tensor) -> bool:
        wants_grad = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))
        return (x.is_cuda and not self.training and not wants_grad and self.lstm.hidden_size % 4 == 0
                and not self.lstm.bidirectional and self.lstm.proj_size == 0)

    def forward(self, x: torch.Tensor, lengths: Optional[torch.Tensor] = None) -> torch.Tensor:
        if self._hip_eligible(x):
            return self.fused(x, lengths)
        packed = lengths is not None
        if packed:
            x = nn.utils.rnn.pack_padded_sequence(x, lengths.cpu(), batch_first=True, enforce_sorted=self.training)
        self.lstm.flatten_parameters()
        y, _ = self.lstm(x)
        if packed:
            y, _ = nn.utils.rnn.pad_packed_sequence(y, batch_first=True)
        y = self.activation(y)
        y = self.norm(y.transpose(1, 2)).transpose(1, 2)
        return self.linear(y)

    def fused(self, x: torch.Tensor, lengths: Optional[torch.Tensor]) -> torch.Tensor:
        """Eval forward on the gfx950 kernels.  Note pad_packed_sequence (decoder.py:22) trims the time axis to
        max(lengths); the encoder guarantees max(lengths) == T' (encoder.py / masking.py:4-13), so no trim happens here."""
        h = x.float() if x.dtype != torch.float32 else x
        lens = None if lengths is None else lengths.to(device=h.device, dtype=torch.int64)
        for k in range(self.lstm.num_layers):
            w_ih, w_hh = getattr(self.lstm, f"weight_ih_l{k}"), getattr(self.lstm, f"weight_hh_l{k}")
            b_ih, b_hh = getattr(self.lstm, f"bias_ih_l{k}"), getattr(self.lstm, f"bias_hh_l{k}")
            bias = self._packs.get(f"bias{k}", (b_ih, b_hh), lambda: (b_ih + b_hh).detach().contiguous())
            h = ops.lstm_forward(h, w_ih.detach(), w_hh.detach(), bias, lens)
        n = self.norm
        z = ops.swish_bn_eval(h, n.running_mean, n.running_var, n.weight.detach(), n.bias.detach(), n.eps)
        return ops.linear(z, self.linear.weight.detach(), self.linear.bias.detach())
