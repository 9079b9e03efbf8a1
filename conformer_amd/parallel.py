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


def wrap_ddp(model: torch.nn.Module, device: Optional[torch.device] = None, bucket_cap_mb: int = 25,
             gradient_as_bucket_view: bool = False, static_graph: bool = False) -> torch.nn.Module:
    """DistributedDataParallel with the reference's settings (train.py:186): default 25 MB buckets, buffers broadcast
    from rank 0 every forward, gradients averaged over ranks.  The three knobs are DDP's own (bench.py exposes them)."""
    if not dist.is_initialized():
        return model
    ids = [device.index] if device is not None and device.type == "cuda" else None
    return torch.nn.parallel.DistributedDataParallel(model, device_ids=ids, bucket_cap_mb=bucket_cap_mb,
                                                     gradient_as_bucket_view=gradient_as_bucket_view, static_graph=static_graph)


class DdpCommProbe:
    """Measures what the gradient exchange costs a training step WITHOUT changing it: a DDP communication hook that performs
    the default exchange (divide by the world size, all-reduce the bucket: torch's `allreduce_hook`) and time-stamps it.

    Per bucket: `ready` = the bucket's gradients are all computed (the hook fires on the backward stream right behind the
    kernel that produced the last of them), `done` = its all-reduce has finished (stamped from the future's completion
    callback: on the collective's stream for RCCL, on the host for gloo).  Per step:
        exposed_comm_ms  = done(last bucket to finish) - ready(last bucket to become ready)
                           -- the tail of the exchange that no backward kernel is left to hide (0 when the ring keeps up);
        comm_span_ms     = done(last) - ready(first): how long the exchange was in flight beside the backward;
        allreduce_bytes  = bytes handed to the collective (= 4 x parameters for fp32 gradients), allreduce_buckets.
    Device stamps are HIP events (resolved in `summary()`, after a sync); gloo runs use host clocks."""

    def __init__(self, ddp: torch.nn.Module) -> None:
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.backend = dist.get_backend() if dist.is_initialized() else None
        self.steps = []            # one list of bucket records per step
        self._cur = None
        self.enabled = isinstance(ddp, torch.nn.parallel.DistributedDataParallel)
        if self.enabled:
            ddp.register_comm_hook(self, DdpCommProbe._hook)

    def begin_step(self) -> None:
        self._cur = []
        self.steps.append(self._cur)

    @staticmethod
    def _hook(self, bucket):
        t = bucket.buffer()
        on_dev = t.is_cuda and self.backend == "nccl"
        rec = {"bytes": t.numel() * t.element_size(), "t_ready": time.perf_counter(), "device": on_dev}
        if on_dev:
            rec["ev_ready"] = torch.cuda.Event(enable_timing=True)
            rec["ev_ready"].record()
        if self._cur is not None:
            self._cur.append(rec)
        t.div_(self.world)
        fut = dist.all_reduce(t, async_op=True).get_future()

        def done(f):
            rec["t_done"] = time.perf_counter()
            if on_dev:                                   # (the callback runs with the collective's stream current)
                rec["ev_done"] = torch.cuda.Event(enable_timing=True)
                rec["ev_done"].record()
            return f.value()[0]

        return fut.then(done)

    def summary(self, skip: int = 0) -> dict:
        """Means over the recorded steps after the first `skip` (call after a device sync)."""
        rows = []
        for recs in self.steps[skip:]:
            recs = [r for r in recs if "t_done" in r]
            if not recs:
                continue
            if recs[0]["device"]:
                first = recs[0]["ev_ready"]
                last_ready = recs[-1]["ev_ready"]                       # hooks fire in the order the buckets become ready
                span = max(first.elapsed_time(r["ev_done"]) for r in recs)
                exposed = max(last_ready.elapsed_time(r["ev_done"]) for r in recs)
            else:
                t0, tl = recs[0]["t_ready"], max(r["t_ready"] for r in recs)
                td = max(r["t_done"] for r in recs)
                span, exposed = (td - t0) * 1e3, (td - tl) * 1e3
            rows.append((max(exposed, 0.0), span, sum(r["bytes"] for r in recs), len(recs)))
        if not rows:
            return {"exposed_comm_ms": 0.0 if not self.enabled else None, "comm_span_ms": 0.0 if not self.enabled else None,
                    "allreduce_bytes_per_step": 0, "allreduce_buckets": 0}
        n = len(rows)
        return {"exposed_comm_ms": sum(r[0] for r in rows) / n, "comm_span_ms": sum(r[1] for r in rows) / n,
                "allreduce_bytes_per_step": rows[-1][2], "allreduce_buckets": rows[-1][3]}


def gather_to_rank0(value):
    """Python object from every rank, as a list on every rank ([value] without a process group)."""
    if not dist.is_initialized():
        return [value]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, value)
    return out
