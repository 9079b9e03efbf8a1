"""Shared guards/caches for the module surface (not part of the reference API)."""
from __future__ import annotations

import os
from typing import Callable, Dict, Sequence, Tuple

import torch

STRICT = os.environ.get("CONFORMER_AMD_STRICT", "0") == "1"


def require_inference(module: torch.nn.Module, what: str, *tensors: torch.Tensor) -> None:
    """Round-1 scope: the HIP path implements the forward pass.  Refuse (loudly) anything that would
    silently produce non-differentiable or train-mode-incorrect results."""
    if torch.is_grad_enabled() and (any(t.requires_grad for t in tensors if isinstance(t, torch.Tensor))
                                    or any(p.requires_grad for p in module.parameters())):
        raise NotImplementedError(
            f"{what}: the gfx950 backward kernels are not part of this build yet; call under torch.no_grad() "
            "/ torch.inference_mode().  (No autograd fallback exists on purpose.)")


class PackCache:
    """Derived device tensors (fused QKV weight, re-laid-out conv/linear weights) keyed on the identity and
    in-place version of their source parameters, so eval pays for the re-layout once and training would
    re-pack after every optimizer step."""

    def __init__(self) -> None:
        self._store: Dict[str, Tuple[tuple, torch.Tensor]] = {}

    def get(self, name: str, srcs: Sequence[torch.Tensor], make: Callable[[], torch.Tensor]) -> torch.Tensor:
        key = tuple((s.data_ptr(), s._version, s.device) for s in srcs)
        hit = self._store.get(name)
        if hit is not None and hit[0] == key:
            return hit[1]
        val = make()
        self._store[name] = (key, val)
        return val

    def clear(self) -> None:
        self._store.clear()
