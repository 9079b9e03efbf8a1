#!/usr/bin/env python3
"""Row N3 measured: is batch n+1's front end (H2D copy, reflect pad, DFT GEMM, mel, length sort -- side stream) inside batch n's
model step (main stream)?  HIP events on both streams against one origin; prints one JSON line with, per step, the front-end
span, the model-step span and the EXPOSED front-end time (what the model step had to wait for at the hand-over).

    python tools/frontend_timeline.py [--batch 32] [--seconds 10] [--steps 8] [--serial]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conformer_amd.frontend import ConformerAudioFrontend  # noqa: E402
from conformer_amd.graph import GraphedEncoder  # noqa: E402
from conformer_amd.pipeline import FrontendPipeline  # noqa: E402
from model.modules.encoder import Encoder  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--seconds", type=float, default=9.99)          # 999 hops -> T = 1000 mel frames (BASELINE cfg-2)
    ap.add_argument("--steps", type=int, default=8)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    n = int(args.seconds * 16000) // 160 * 160
    g = torch.Generator().manual_seed(1)
    batches = [[torch.randn(n, generator=g) * 0.1 for _ in range(args.batch)] for _ in range(args.steps + 2)]
    fe = ConformerAudioFrontend(device=dev)
    enc = Encoder(80, 16, 512, 8, 31, 0.0).to(dev).eval()
    T = n // 160 + 1
    x0 = torch.randn(args.batch, 80, T, device=dev)
    L0 = torch.full((args.batch,), T, dtype=torch.int64, device=dev)
    model = GraphedEncoder(enc, x0, L0)
    # the front end alone (serial reference): device time of one batch
    wave = torch.stack(batches[0]).to(dev)
    for _ in range(3):
        fe.mel_spectrogram(wave)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fe.mel_spectrogram(wave)
    e1.record()
    torch.cuda.synchronize()
    logmel_us = e0.elapsed_time(e1) / 10 * 1e3

    pipe = FrontendPipeline(batches, fe, depth=2)
    pipe.timeline = []
    origin = torch.cuda.Event(enable_timing=True)
    origin.record()
    steps = []
    for i, (mels, frames, order) in enumerate(pipe):
        s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0.record()                                   # (after the hand-over wait: what the model step really starts at)
        with torch.no_grad():
            model(mels, frames)
        s1.record()
        steps.append((s0, s1))
    torch.cuda.synchronize()
    rows = []
    for i, ((f0, f1), (s0, s1)) in enumerate(zip(pipe.timeline, steps)):
        fe_start, fe_done = origin.elapsed_time(f0), origin.elapsed_time(f1)
        st_start, st_done = origin.elapsed_time(s0), origin.elapsed_time(s1)
        prev_done = origin.elapsed_time(steps[i - 1][1]) if i else 0.0
        rows.append({"batch": i, "frontend_ms": [round(fe_start, 3), round(fe_done, 3)], "model_step_ms": [round(st_start, 3), round(st_done, 3)],
                     "frontend_span_ms": round(fe_done - fe_start, 3),
                     "exposed_frontend_ms": round(max(0.0, fe_done - max(prev_done, fe_start)) if i else fe_done - fe_start, 3)})
    steady = rows[2:] or rows
    print(json.dumps({"what": "front-end / model-step overlap (FrontendPipeline, depth 2)", "batch": args.batch, "mel_frames": T,
                      "logmel_device_us_serial": round(logmel_us, 1),
                      "model_step_ms": round(sum(r["model_step_ms"][1] - r["model_step_ms"][0] for r in steady) / len(steady), 3),
                      "frontend_span_ms": round(sum(r["frontend_span_ms"] for r in steady) / len(steady), 3),
                      "exposed_frontend_ms": round(sum(r["exposed_frontend_ms"] for r in steady) / len(steady), 4),
                      "note": "exposed = front-end time of batch n+1 that ends after model step n has ended (the first two batches fill the pipe)",
                      "per_batch": rows}))


if __name__ == "__main__":
    main()
