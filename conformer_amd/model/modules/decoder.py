"""Decoder (surface of model/modules/decoder.py:7-27): packed LSTM -> Swish -> BatchNorm1d -> Linear.

SURVEY 8f row N1.  On a HIP device everything runs on gfx950 kernels: one GEMM for the input projection, the LSTM
recurrence (one launch per frame, csrc/lstm.hip), Swish+BatchNorm (running statistics in eval, batch statistics + running
update in train) and the vocabulary GEMM; training goes through torch.autograd.Functions whose backward is explicit
kernels too (back-propagation through time + GEMMs).  No host round-trip for `lengths` (the reference's `lengths.cpu()`
at decoder.py:17 synchronises the stream every call).  CPU tensors, bidirectional / projected LSTMs and hidden sizes
that are not a multiple of 4 use the stock modules (the reference never builds those).  State_dict keys are the
reference's; `enforce_sorted` (decoder.py:17) is not needed: the kernels take ragged lengths in any order.
"""
from typing import Optional

import torch
import torch.nn as nn

from ... import autograd as A
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
        return (x.is_cuda and self.lstm.hidden_size % 4 == 0 and not self.lstm.bidirectional and self.lstm.proj_size == 0
                and self.lstm.dropout == 0.0)

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
        """Forward on the gfx950 kernels.  Note pad_packed_sequence (decoder.py:22) trims the time axis to max(lengths);
        the encoder guarantees max(lengths) == T' (masking.py:4-13), so no trim happens here."""
        h = x.float() if x.dtype != torch.float32 else x
        lens = None if lengths is None else lengths.to(device=h.device, dtype=torch.int64)
        n = self.norm
        wants_grad = torch.is_grad_enabled() and (h.requires_grad or any(p.requires_grad for p in self.parameters()))
        if wants_grad:
            for k in range(self.lstm.num_layers):
                h = A.LstmFn.apply(h, getattr(self.lstm, f"weight_ih_l{k}"), getattr(self.lstm, f"weight_hh_l{k}"),
                                   getattr(self.lstm, f"bias_ih_l{k}"), getattr(self.lstm, f"bias_hh_l{k}"), lens)
            train_bn = n.training or n.running_mean is None
            momentum = 0.1 if n.momentum is None else n.momentum
            z = A.SwishBatchNormFn.apply(h, n.weight, n.bias, n.running_mean, n.running_var, train_bn, momentum, n.eps)
            if train_bn and n.num_batches_tracked is not None:
                n.num_batches_tracked += 1
            return A.LinearFn.apply(z, self.linear.weight, self.linear.bias)
        for k in range(self.lstm.num_layers):
            w_ih, w_hh = getattr(self.lstm, f"weight_ih_l{k}"), getattr(self.lstm, f"weight_hh_l{k}")
            b_ih, b_hh = getattr(self.lstm, f"bias_ih_l{k}"), getattr(self.lstm, f"bias_hh_l{k}")
            bias = self._packs.get(f"bias{k}", (b_ih, b_hh), lambda: (b_ih + b_hh).detach().contiguous())
            h = ops.lstm_forward(h, w_ih.detach(), w_hh.detach(), bias, lens)
        if n.training:
            mean, var = ops.swish_bn_batch_stats(h, n.running_mean, n.running_var, 0.1 if n.momentum is None else n.momentum)
            if n.num_batches_tracked is not None:
                n.num_batches_tracked += 1
        else:
            mean, var = n.running_mean, n.running_var
        z = ops.swish_bn_eval(h, mean, var, n.weight.detach(), n.bias.detach(), n.eps)
        return ops.linear(z, self.linear.weight.detach(), self.linear.bias.detach())
