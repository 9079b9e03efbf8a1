"""Shared guards/caches for the module surface (not part of the reference API)."""
from __future__ import annotations

import os
import weakref
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


def _ver(t: torch.Tensor) -> int:
    """In-place version of a source tensor; inference tensors (parameters created or loaded under
    torch.inference_mode(), as the reference's infer.py / test.py run) carry no version counter and cannot be modified
    in place outside inference mode, so 0 identifies them."""
    return 0 if t.is_inference() else t._version


_EPOCH = [0]


def invalidate_weight_caches() -> None:
    """Drop every derived weight tensor (fused QKV, packed conv / linear weights, projected position tables, 16-bit and
    split-plane weight copies) so the next forward rebuilds them.  Needed only after writes that bypass PyTorch's version
    counter -- `param.data.copy_(...)`, `param.data = ...` keeps working through the identity check -- because nothing
    on the host can see those; optimizer steps, `load_state_dict` and ordinary in-place ops are detected automatically."""
    _EPOCH[0] += 1
    from conformer_amd import ops
    ops._W16_CACHE.clear()
    ops._WSPLIT_CACHE.clear()


class PackCache:
    """Derived device tensors (fused QKV weight, re-laid-out conv/linear weights) keyed on the IDENTITY (weak reference:
    a freed tensor may hand its address to a new one), storage address and in-place version of their source tensors, so
    eval pays for the re-layout once and training re-packs after every optimizer step."""

    def __init__(self) -> None:
        self._store: Dict[str, Tuple[tuple, tuple, torch.Tensor]] = {}

    def get(self, name: str, srcs: Sequence[torch.Tensor], make: Callable[[], torch.Tensor]) -> torch.Tensor:
        key = (_EPOCH[0],) + tuple((s.data_ptr(), _ver(s), s.device, tuple(s.shape)) for s in srcs)
        hit = self._store.get(name)
        if hit is not None and hit[0] == key and all(r() is s for r, s in zip(hit[1], srcs)):
            return hit[2]
        val = make()
        self._store[name] = (key, tuple(weakref.ref(s) for s in srcs), val)
        return val

    def clear(self) -> None:
        self._store.clear()
