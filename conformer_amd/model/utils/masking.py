"""generate_padding_mask (surface of model/utils/masking.py:4-13).  Integer/bool bookkeeping only; the
fused attention kernel consumes `lengths` directly and never needs this mask."""
from typing import Optional

import torch


def generate_padding_mask(lengths: torch.Tensor, max_length: Optional[int] = None) -> torch.Tensor:
    """(B,) lengths -> (B, max_length) bool, True at valid frames.  `max_length=None` uses lengths.max()
    (a host sync, as in the reference)."""
    if max_length is None:
        max_length = int(lengths.max())
    steps = torch.arange(max_length, dtype=lengths.dtype, device=lengths.device)
    return steps[None, :] < lengths[:, None]


def lengths_from_key_padding_mask(mask: torch.Tensor) -> torch.Tensor:
    """Inverse of the encoder's `(~generate_padding_mask(lengths))[:, None, None, :]` (encoder.py:30):
    a (B,1,1,T) bool mask, True at padded keys -> (B,) int64 valid-key counts.  Only suffix (key-padding)
    masks are representable; anything else is rejected by shape."""
    if mask.dtype != torch.bool or mask.dim() != 4 or mask.shape[1] != 1 or mask.shape[2] != 1:
        raise NotImplementedError(
            f"attention mask of shape {tuple(mask.shape)} / dtype {mask.dtype}: the fused gfx950 attention kernel "
            "supports key-padding masks (B,1,1,T) bool only")
    return (~mask).sum(dim=-1).reshape(-1).to(torch.int64)
