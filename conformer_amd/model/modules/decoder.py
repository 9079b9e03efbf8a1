"""Decoder (surface of model/modules/decoder.py:7-27): packed LSTM -> Swish -> BatchNorm1d -> Linear.

Out of the HIP scope this round (1.4 % of the reference's CPU time): the arithmetic stays on
stock PyTorch-ROCm modules (MIOpen LSTM), only the module surface and state_dict keys are reproduced.
"""
from typing import Optional

import torch
import torch.nn as nn

from ..utils.activation import Swish


class Decoder(nn.Module):
    def __init__(self, vocab_size: int, d_model: int, hidden_dim: int, n_layers: int) -> None:
        super().__init__()
        self.lstm = nn.LSTM(input_size=d_model, hidden_size=hidden_dim, num_layers=n_layers, batch_first=True)
        self.activation = Swish()
        self.norm = nn.BatchNorm1d(num_features=hidden_dim)
        self.linear = nn.Linear(in_features=hidden_dim, out_features=vocab_size)

    def forward(self, x: torch.Tensor, lengths: Optional[torch.Tensor] = None) -> torch.Tensor:
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
