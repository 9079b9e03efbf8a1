#!/usr/bin/env python3
"""One full training step with the reference's semantics (train.py:226-243), on the gfx950 path, fp32:

    Conformer(x, lengths) -> CTC loss on log_softmax (evaluation.py:12-16) -> backward -> Adam(lr 2e-5) step

    python tools/train_step.py [--batch 32] [--steps 5]                       # single GPU
    python -m torch.distributed.run --nproc-per-node N tools/train_step.py    # DDP over RCCL, per-GPU batch = --batch

Synthetic cfg-3 shapes (T=1000 mel frames, 40 target tokens per utterance), dropout 0.0, BatchNorm in train mode.
Prints one JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conformer_amd import parallel  # noqa: E402
from model.conformer import Conformer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--optim", choices=["torch", "fused"], default="fused")
    ap.add_argument("--ctc", choices=["hip", "torch"], default="hip", help="CTC loss: gfx950 lattice kernels or torch.nn.CTCLoss")
    ap.add_argument("--dtype", choices=["f32", "bf16", "f16"], default="f32",
                    help="bf16/f16: model under torch.autocast (f16 with GradScaler), as train.py:217,232-243")
    args = ap.parse_args()
    env = parallel.env_from_os()
    torch.cuda.set_device(env.local_rank)
    dev = torch.device("cuda", env.local_rank)
    parallel.init_distributed(env, dev)
    torch.manual_seed(0)
    model = Conformer(370, 80, 16, 512, 8, 31, 640, 1, 0.0).to(dev).train()
    ddp = parallel.wrap_ddp(model, dev)
    if args.optim == "fused":
        from conformer_amd.optim import FusedAdam
        opt = FusedAdam(model.parameters(), lr=2e-5)
    else:
        opt = torch.optim.Adam(model.parameters(), lr=2e-5)
    g = torch.Generator().manual_seed(100 + env.rank)
    x = torch.randn(args.batch, 80, args.frames, generator=g).to(dev)
    lengths = torch.full((args.batch,), args.frames, dtype=torch.int64, device=dev)
    targets = torch.randint(1, 370, (args.batch, 40), generator=g).to(dev)
    tlen = torch.full((args.batch,), 40, dtype=torch.int64, device=dev)
    from conformer_amd.evaluation import ConformerCriterion
    crit = ConformerCriterion(blank_id=0)
    torch_ctc = torch.nn.CTCLoss(blank=0, zero_infinity=True)
    last = {}

    amp_dtype = {"f32": None, "bf16": torch.bfloat16, "f16": torch.float16}[args.dtype]
    scaler = torch.amp.GradScaler("cuda", enabled=args.dtype == "f16")

    def step():
        with torch.autocast("cuda", dtype=amp_dtype, enabled=amp_dtype is not None):
            logits, out_len = ddp(x, lengths)
        if args.ctc == "hip":
            loss = crit.ctc_loss(logits, targets, out_len, tlen)
        else:
            loss = torch_ctc(logits.float().log_softmax(-1).transpose(0, 1), targets, out_len, tlen)
        opt.zero_grad(set_to_none=True)
        scaler.scale(loss).backward()
        scaler.unscale_(opt)
        scaler.step(opt)
        scaler.update()
        last["loss"] = loss

    dt = parallel.timed_steps(step, args.steps, args.warmup, torch.cuda.synchronize, dev)
    # host-side issue time of one step (no sync inside): tells whether the step is launch-bound
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        step()
    host_ms = (time.perf_counter() - t0) / 3 * 1e3
    torch.cuda.synchronize()
    if env.is_main:
        ms = dt / args.steps * 1e3
        print(json.dumps({"what": "Conformer-L training step fwd+CTC+bwd+Adam, dropout 0, BN train", "dtype": args.dtype,
                          "optimizer": args.optim, "ctc": args.ctc, "n_gpus": env.world, "per_gpu_batch": args.batch, "mel_frames": args.frames,
                          "ms_per_step": ms, "host_issue_ms_per_step": host_ms, "frames_per_sec": env.world * args.batch * args.frames * args.steps / dt,
                          "loss": float(last["loss"]), "max_mem_gib": torch.cuda.max_memory_allocated() / 2 ** 30}),
              flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
