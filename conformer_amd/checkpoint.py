"""Checkpoint compatibility (reference: checkpoint.py:7-35, manager.py:23-47).

The reference saves {'model', 'optimizer', 'scheduler', 'n_steps', 'n_epochs'} with `torch.save` and reloads with a
strict `load_state_dict`, adding / stripping DistributedDataParallel's "module." prefix as needed.  The module mirror
keeps the reference's state_dict keys and shapes and `FusedAdam` keeps torch.optim.Adam's state layout, so reference
`.pt` files load unchanged; these helpers do it with `weights_only=True` (nothing in the file is executed)."""
from __future__ import annotations

from collections import OrderedDict
from typing import Mapping, Optional, Tuple, Union

import torch

_PREFIX = "module."


def is_ddp_state_dict(state_dict: Mapping[str, torch.Tensor]) -> bool:
    """True when every key carries DistributedDataParallel's prefix (checkpoint.py:7-11)."""
    return len(state_dict) > 0 and all(k.startswith(_PREFIX) for k in state_dict)


def convert_prefix(state_dict: Mapping[str, torch.Tensor], ddp: bool) -> "OrderedDict[str, torch.Tensor]":
    """Return the state dict keyed for a DDP-wrapped (ddp=True) or a bare (ddp=False) model."""
    has = is_ddp_state_dict(state_dict)
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for k, v in state_dict.items():
        if ddp and not has:
            k = _PREFIX + k
        elif not ddp and has:
            k = k[len(_PREFIX):]
        out[k] = v
    return out


def load_model(source: Union[str, Mapping[str, torch.Tensor]], model: torch.nn.Module, world_size: int = 1) -> None:
    """checkpoint.py:27-35: `source` is a checkpoint path (its 'model' entry is used) or a state dict; strict load."""
    if isinstance(source, str):
        source = torch.load(source, map_location="cpu", weights_only=True)["model"]
    wrapped = isinstance(model, torch.nn.parallel.DistributedDataParallel) or world_size > 1
    model.load_state_dict(convert_prefix(source, ddp=wrapped), strict=True)


def save_checkpoint(path: str, model: torch.nn.Module, optimizer: torch.optim.Optimizer, scheduler=None, n_steps: int = 0,
                    n_epochs: int = 0) -> None:
    """manager.py:34-44 layout."""
    torch.save({"model": model.state_dict(), "optimizer": optimizer.state_dict(),
                "scheduler": scheduler.state_dict() if scheduler is not None else {}, "n_steps": n_steps,
                "n_epochs": n_epochs}, path)


def load_checkpoint(path: str, model: torch.nn.Module, optimizer: Optional[torch.optim.Optimizer] = None, scheduler=None,
                    world_size: int = 1) -> Tuple[int, int]:
    """manager.py:23-32: restores model (+ optimizer, scheduler) and returns (n_steps, n_epochs)."""
    data = torch.load(path, map_location="cpu", weights_only=True)
    load_model(data["model"], model, world_size=world_size)
    if optimizer is not None:
        optimizer.load_state_dict(data["optimizer"])
    if scheduler is not None and data.get("scheduler"):
        scheduler.load_state_dict(data["scheduler"])
    return int(data["n_steps"]), int(data["n_epochs"])
