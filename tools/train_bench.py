#!/usr/bin/env python3
"""Times one training step of the encoder hot path (forward + backward through the gfx950 kernels), fp32.

    python tools/train_bench.py [--batch 32] [--steps 5] [--blocks 16]

loss = sum(enc * w), so the timing isolates the encoder's forward + backward (all parameters, stem included).  Prints ms/step and mel-frames/s.
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from model.modules.encoder import Encoder  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--blocks", type=int, default=16)
    ap.add_argument("--frames", type=int, default=1000)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    enc = Encoder(80, args.blocks, 512, 8, 31, 0.0).to(dev).train()
    x = torch.randn(args.batch, 80, args.frames, device=dev)
    L = torch.full((args.batch,), args.frames, dtype=torch.int64, device=dev)
    w = None

    def step():
        nonlocal w
        for p in enc.parameters():
            p.grad = None
        y, _ = enc(x, L)
        if w is None:
            w = torch.randn_like(y)
        (y * w).sum().backward()

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    with torch.no_grad():
        enc.eval()
        for _ in range(2):
            enc(x, L)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            enc(x, L)
        torch.cuda.synchronize()
        df = (time.perf_counter() - t0) / args.steps
    print(f"train step (fwd+bwd, B={args.batch}, T={args.frames}, {args.blocks} blocks, fp32): {dt * 1e3:.1f} ms "
          f"= {args.batch * args.frames / dt:,.0f} frames/s;  eval forward {df * 1e3:.1f} ms;  "
          f"max mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")


if __name__ == "__main__":
    main()
