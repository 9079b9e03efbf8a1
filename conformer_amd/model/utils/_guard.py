"""Shared guards/caches for the module surface (not part of the reference API)."""
from __future__ import annotations

import os
from typing import Callable, Dict, Sequence, Tuple

import torch

STRICT = os.environ.get("CONFORMER_AMD_STRICT", "0") == "1"


def active_dropout(drop: torch.nn.Dropout) -> float:
    """The probability an nn.Dropout sub-module applies right now (0 in eval mode) -- nn.Dropout semantics."""
    return float(drop.p) if drop.training else 0.0


def refuse_dropout(module: torch.nn.Module, what: str) -> None:
    """Dropout is only wired through the autograd Functions (training with gradients).  `.train()` under no_grad with
    p > 0 has no kernel path: refuse instead of silently skipping the masks."""
    if module.training and not torch.is_grad_enabled():
        for m in module.modules():
            if isinstance(m, torch.nn.Dropout) and m.p > 0:
                raise NotImplementedError(f"{what}: dropout p={m.p} in training mode under no_grad is not built "
                                          "(call .eval() for inference)")


def refuse_grad(module: torch.nn.Module, what: str, *tensors: torch.Tensor) -> None:
    """For pieces whose backward kernels do not exist yet (the conv-subsampling stem): refuse a call that would
    need gradients rather than return a silently non-differentiable result."""
    if torch.is_grad_enabled() and (any(t.requires_grad for t in tensors if isinstance(t, torch.Tensor))
                                    or any(p.requires_grad for p in module.parameters())):
        raise NotImplementedError(
            f"{what}: backward kernels for this piece are not built yet; freeze it (requires_grad_(False)) or call "
            "under torch.no_grad().  (No autograd fallback exists on purpose.)")


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
