#!/usr/bin/env python3
"""Forward / backward 16-bit-MFMA GEMM sites of one Conformer-L layer at cfg-2 (M = 7968) under bf16 autocast:
hipGraph-replay device time per call and TFLOP/s."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conformer_amd import ops  # noqa: E402
from tools.kernel_table import time_us  # noqa: E402


def trace(M, N, K, kind):
    """Per-K-tile timeline of two workgroups of one forward 16-bit GEMM (100 MHz stamps)."""
    from conformer_amd import _lib
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    R = lambda *s: torch.randn(*s, device=dev, generator=g)
    a, w, b, r = R(M, K), R(N, K), R(N), R(M, N)
    a16 = a.to(torch.bfloat16)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        for src in (a, a16):
            fn = {"swish": lambda: ops.linear_train("swish", src, w, b, save_z=True), "resid": lambda: ops.linear_residual(src, w, b, r, 0.5),
                  "bias": lambda: ops.linear(src, w, b)}[kind]
            for _ in range(3):
                fn()
            tr = torch.zeros(128, dtype=torch.int64, device=dev)
            _lib.load().cfm_debug_gemm_mfma16_trace(tr.data_ptr())
            fn()
            torch.cuda.synchronize()
            _lib.load().cfm_debug_gemm_mfma16_trace(None)
            t = tr.cpu().view(2, 64)
            nkt = (K + 63) // 64
            for wgi in range(2):
                row = t[wgi]
                t0 = int(row[0])
                ks = [(int(row[2 + i]) - (int(row[1 + i]))) * 10 for i in range(min(nkt, 60))]
                print(f"A {str(src.dtype):15s} wg{wgi}: prologue {(int(row[1]) - t0) * 10} ns | per K-tile {ks} | epilogue issue "
                      f"{(int(row[62]) - int(row[1 + min(nkt, 60)])) * 10} drain {(int(row[63]) - int(row[62])) * 10} | total {(int(row[63]) - t0) * 10} ns")
            print(f"   whole launch: {time_us(fn, 20):.1f} us")


def sweep():
    """Forward 16-bit GEMM sites x forced block tile (1 = 128x128 family, 2 = 256x128, 3 = 256x256), bf16 A operand."""
    from conformer_amd import _lib
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    R = lambda *s: torch.randn(*s, device=dev, generator=g)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        for M in (7968, 15936):
            for name, N, K, kind in [("FFN hidden swish", 2048, 512, "swish"), ("FFN out residual", 512, 2048, "resid"),
                                     ("QKV", 1536, 512, "bias"), ("attn out / pw2", 512, 512, "resid"), ("input linear", 512, 9728, "bias")]:
                a, w, b, r = R(M, K).to(torch.bfloat16), R(N, K), R(N), R(M, N)
                fn = {"swish": lambda: ops.linear(a, w, b, act="swish"), "resid": lambda: ops.linear_residual(a, w, b, r, 0.5),
                      "bias": lambda: ops.linear(a, w, b)}[kind]
                line = f"{name:18s} {M}x{N}x{K}: "
                for tile in (0, 1, 2, 3):
                    _lib.load().cfm_debug_gemm_mfma16_force_tile(tile)
                    line += f"| t{tile} {time_us(fn, 20):6.1f} us "
                _lib.load().cfm_debug_gemm_mfma16_force_tile(0)
                print(line, flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "sweep":
        return sweep()
    if len(sys.argv) > 1 and sys.argv[1] == "trace":
        return trace(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5])
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
