#!/usr/bin/env python3
"""Forward / backward 16-bit-MFMA GEMM sites of one Conformer-L layer at cfg-2 (M = 7968) under bf16 autocast:
hipGraph-replay device time per call and TFLOP/s."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conformer_amd import ops  # noqa: E402
from tools.kernel_table import time_us  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 7968
    g = torch.Generator(device=dev).manual_seed(0)
    R = lambda *s: torch.randn(*s, device=dev, generator=g)
    rows = []
    with torch.autocast("cuda", dtype=torch.bfloat16):
        for name, N, K, kind in [("FFN hidden swish", 2048, 512, "swish"), ("FFN out residual", 512, 2048, "resid"),
                                 ("QKV", 1536, 512, "bias"), ("attn out / pw2", 512, 512, "resid"), ("pw1 GLU", 1024, 512, "glu")]:
            a, w, b, r = R(M, K), R(N * (2 if kind == "glu" else 1), K), R(N * (2 if kind == "glu" else 1)), R(M, N)
            fn = {"swish": lambda: ops.linear(a, w, b, act="swish"), "resid": lambda: ops.linear_residual(a, w, b, r, 0.5),
                  "bias": lambda: ops.linear(a, w, b), "glu": lambda: ops.linear_glu(a, w, b)}[kind]
            us = time_us(fn, 20)
            n_eff = N * (2 if kind == "glu" else 1)
            rows.append((f"fwd {name} {M}x{n_eff}x{K}", us, 2.0 * M * n_eff * K / us / 1e6))
        for name, N, K in [("FFN hidden", 2048, 512), ("FFN out", 512, 2048), ("QKV", 1536, 512), ("out/pw2", 512, 512)]:
            x, w, dy = R(M, K), R(N, K), R(M, N)
            us = time_us(lambda: ops.linear_bwd(x, w, dy), 20)
            rows.append((f"bwd dX+dW+db {name} (N={N},K={K})", us, 4.0 * M * N * K / us / 1e6))
    for n, us, tf in rows:
        print(f"{n:48s} {us:8.1f} us  {tf:7.1f} TFLOP/s")


if __name__ == "__main__":
    main()
