"""Data parallelism for the hot path: one process per GPU, torch.distributed on RCCL (backend "nccl" on ROCm) over xGMI.

Utterances shard over the batch axis.  Inference needs no exchange at all (independent replicas); training has exactly one
exchange per step, the gradient all-reduce, which `wrap_ddp` delegates to DistributedDataParallel: its bucketed
all-reduce is launched from autograd hooks as the explicit backward kernels (conformer_amd/autograd.py) produce each
gradient, on RCCL's own stream, so it overlaps with the rest of the backward (train.py:186 semantics: plain
BatchNorm, per-replica statistics; buffers broadcast from rank 0).
"""
from __future__ import annotations

import os
import time
from dataclasses import dataclass
from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


@dataclass
class DistEnv:
    rank: int
    world: int
    local_rank: int

    @property
    def is_main(self) -> bool:
        return self.rank == 0


def env_from_os() -> DistEnv:
    return DistEnv(int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
                   int(os.environ.get("LOCAL_RANK", "0")))


def init_distributed(env: DistEnv, device: Optional[torch.device] = None, backend: Optional[str] = None) -> bool:
    """Initialise the default process group when WORLD_SIZE > 1 (MASTER_ADDR/PORT from the launcher).  Returns True
    when a group was created."""
    if env.world <= 1:
        return False
    if backend is None:
        backend = "nccl" if (device is not None and device.type == "cuda") else "gloo"
    kwargs = {}
    if backend == "nccl" and device is not None:
        kwargs["device_id"] = device
    dist.init_process_group(backend, rank=env.rank, world_size=env.world, **kwargs)
    return True


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) slice of a global batch for `rank` (sizes differ by at most one, earlier ranks larger) --
    the DistributedSampler-equivalent partition of train.py:203 for an in-memory batch."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def timed_steps(step: Callable[[], None], steps: int, warmup: int, sync: Callable[[], None],
                device: Optional[torch.device] = None) -> float:
    """`warmup` untimed steps, then EXACTLY `steps` timed ones bracketed by barrier + device sync on both sides;
    returns the MAX over ranks of the elapsed seconds (the bench.py contract)."""
    for _ in range(warmup):
        step()
    sync()
    if dist.is_initialized():
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    if dist.is_initialized():
        dist.barrier()
    sync()
    dt = time.perf_counter() - t0
    if dist.is_initialized():
        t = torch.tensor([dt], dtype=torch.float64, device=device if device is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def wrap_ddp(model: torch.nn.Module, device: Optional[torch.device] = None, bucket_cap_mb: int = 25) -> torch.nn.Module:
    """DistributedDataParallel with the reference's settings (train.py:186): default 25 MB buckets, buffers broadcast
    from rank 0 every forward, gradients averaged over ranks."""
    if not dist.is_initialized():
        return model
    ids = [device.index] if device is not None and device.type == "cuda" else None
    return torch.nn.parallel.DistributedDataParallel(model, device_ids=ids, bucket_cap_mb=bucket_cap_mb)
