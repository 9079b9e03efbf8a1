"""Fused multi-tensor Adam on gfx950 (reference: torch.optim.Adam(model.parameters(), lr) at
train.py:188).  One kernel launch updates every parameter (param, grad, exp_avg, exp_avg_sq streamed once: 16 B read +
12 B written per element) instead of torch's per-dtype foreach chains.  The optimizer state keeps torch.optim.Adam's
layout ({'step', 'exp_avg', 'exp_avg_sq'} per parameter), so `manager.py`-style checkpoints interchange with the stock
optimizer.  Defaults only: no weight decay, no amsgrad, no maximize."""
from __future__ import annotations

import ctypes
import math
from typing import List

import torch

from . import _lib, ops

class _AdamTensor(ctypes.Structure):      # cfm_adam_tensor, include/conformer_hip.h
    _fields_ = [("p", ctypes.c_void_p), ("g", ctypes.c_void_p), ("m", ctypes.c_void_p), ("v", ctypes.c_void_p),
                ("n", ctypes.c_int64)]


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8) -> None:
        # the extra keys are torch.optim.Adam's own defaults, so a state_dict saved here loads into the stock optimizer
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False, maximize=False,
                                      foreach=None, capturable=False, differentiable=False, fused=None,
                                      decoupled_weight_decay=False))
        self._tables = {}
        self._steps = {}            # group index -> step count (mirrored into state[p]['step'] by state_dict())

    def _table(self, group_idx: int, plist: List[torch.Tensor]):
        """Host descriptor array; parameter/state pointers are stable, only the gradient pointers are refreshed."""
        hit = self._tables.get(group_idx)
        if hit is None or hit[0] != [id(p) for p in plist]:
            arr = (_AdamTensor * len(plist))()
            for i, p in enumerate(plist):
                st = self.state[p]
                arr[i] = _AdamTensor(p.data_ptr(), 0, st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel())
            hit = ([id(p) for p in plist], arr)
            self._tables[group_idx] = hit
        arr = hit[1]
        for i, p in enumerate(plist):
            arr[i].g = p.grad.data_ptr()
            arr[i].p = p.data_ptr()
        return arr

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            plist = [p for p in group["params"] if p.grad is not None]
            if not plist:
                continue
            if group["weight_decay"] or group["amsgrad"] or group["maximize"]:
                raise _lib.ConformerHipError("FusedAdam: weight_decay / amsgrad / maximize are not built (train.py:188 uses none)")
            for p in plist:
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad.is_contiguous()):
                    raise _lib.ConformerHipError("FusedAdam: parameters and gradients must be contiguous fp32 HIP tensors")
                st = self.state[p]
                if not st:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros(p.shape, device=p.device, dtype=torch.float32)
                    st["exp_avg_sq"] = torch.zeros(p.shape, device=p.device, dtype=torch.float32)
            # all parameters of a group step together
            if gi not in self._steps:
                self._steps[gi] = int(self.state[plist[0]]["step"])
            step = self._steps[gi] = self._steps[gi] + 1
            b1, b2 = group["betas"]
            arr = self._table(gi, plist)
            _lib.check(_lib.load().cfm_adam_step_f32(ctypes.addressof(arr), len(plist), float(group["lr"]), b1, b2, group["eps"],
                                                     1.0 - b1 ** step, math.sqrt(1.0 - b2 ** step), ops._stream()),
                       "cfm_adam_step_f32")
            # the kernel wrote through raw pointers: tell autograd (and every cache keyed on `_version`: the packed / fused /
            # 16-bit weight copies of the modules and of ops.weight16) that the parameters changed in place
            torch.autograd.graph.increment_version(plist)
            ops.refresh_weight16(plist)                  # the cached 16-bit weight copies of the autocast path, in one batched launch
        return loss

    def _sync_steps(self) -> None:
        for gi, group in enumerate(self.param_groups):
            if gi in self._steps:
                for p in group["params"]:
                    if p in self.state and self.state[p]:
                        self.state[p]["step"] = torch.tensor(float(self._steps[gi]))

    def state_dict(self):
        self._sync_steps()
        return super().state_dict()

    def load_state_dict(self, state_dict) -> None:
        super().load_state_dict(state_dict)
        self._steps.clear()
        self._tables.clear()
