#!/usr/bin/env python3
"""BASELINE cfg-5: streaming / chunked inference, T=20000 mel frames in 640-frame chunks with cached K/V + depthwise state,
B=8, Conformer-L encoder on one MI355X.  Reports whole-stream throughput, per-chunk latency (first / median / last: the
attention cost grows with the cache) and, for scale, the full-context Encoder.forward of the same batch."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conformer_amd.streaming import StreamingEncoder  # noqa: E402
from model.modules.encoder import Encoder  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--frames", type=int, default=20000)
    ap.add_argument("--chunk", type=int, default=640)
    ap.add_argument("--repeats", type=int, default=3)
    ap.add_argument("--graphs", action="store_true", help="replay one captured hipGraph per chunk step (StreamingEncoder(graphs=True))")
    ap.add_argument("--dtype", choices=["f32", "bf16", "f16"], default="f32", help="bf16 / f16: the whole stream under torch.autocast")
    args = ap.parse_args()
    import contextlib
    amp = contextlib.nullcontext if args.dtype == "f32" else \
        (lambda: torch.autocast("cuda", dtype=torch.bfloat16 if args.dtype == "bf16" else torch.float16))
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    enc = Encoder(80, 16, 512, 8, 31, 0.0).to(dev).eval()
    x = torch.randn(args.batch, 80, args.frames, device=dev)
    st = StreamingEncoder(enc, args.batch, args.frames, graphs=args.graphs)
    best, lat = None, None
    for _ in range(args.repeats + 1):                     # first pass = warm-up (packs, caches; --graphs: the chunk graphs are captured)
        st.reset()
        torch.cuda.synchronize()
        per = []
        t0 = time.perf_counter()
        for t in range(0, args.frames, args.chunk):
            c0 = time.perf_counter()
            with amp():
                st.step(x[:, :, t:t + args.chunk])
            torch.cuda.synchronize()                      # a streaming service hands each chunk's frames on
            per.append((time.perf_counter() - c0) * 1e3)
        dt = time.perf_counter() - t0
        if best is None or dt < best:
            best, lat = dt, per
    with torch.no_grad(), amp():
        enc(x, None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        enc(x, None)
        torch.cuda.synchronize()
        full = time.perf_counter() - t0
    lat_sorted = sorted(lat)
    print(json.dumps({"what": "cfg-5 streaming encoder: cached K/V + depthwise state", "launch": "hipGraph per chunk step" if args.graphs else "eager", "dtype": args.dtype, "batch": args.batch,
                      "mel_frames": args.frames, "chunk": args.chunk, "chunks": len(lat), "encoder_frames": st.frames,
                      "stream_ms": best * 1e3, "frames_per_sec": args.batch * args.frames / best,
                      "realtime_factor_per_stream": (args.frames * 0.010) / best,
                      "chunk_latency_ms": {"first": lat[0], "median": lat_sorted[len(lat) // 2], "last_full": lat[-2], "max": max(lat)},
                      "full_context_forward_ms": full * 1e3,
                      "kv_cache_gib": sum(t.numel() for t in st.qkv) * 4 / 2 ** 30}))


if __name__ == "__main__":
    main()
