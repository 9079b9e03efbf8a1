"""Swish / GLU modules (surface of model/utils/activation.py:4-17 in the reference).

On the hot path both are fused into GEMM / depthwise-conv epilogues (gemm_f32.hip, dwconv.hip); these
stand-alone modules exist for API parity (the stock-torch Decoder uses Swish) and run as stock tensor ops.
"""
import torch
import torch.nn as nn


class Swish(nn.Module):
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return torch.sigmoid(x).mul(x)


class GLU(nn.Module):
    def __init__(self, dim: int) -> None:
        super().__init__()
        self.dim = dim

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        value, gate = torch.chunk(x, 2, dim=self.dim)       # first half is the value, second the gate
        return value * torch.sigmoid(gate)
