#!/usr/bin/env python3
"""Times the fp32 MFMA GEMM at the hot-path shapes for every block-tile shape (GPU box only).

    python tools/gemm_tune.py            # table: shape x cfg -> us, TFLOP/s, and what the heuristic picks
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conformer_amd import _lib  # noqa: E402

SHAPES = [(7968, 2048, 512), (7968, 512, 2048), (7968, 1536, 512), (7968, 512, 512), (7968, 1024, 512),
          (7968, 512, 9728), (497, 512, 512), (15936, 2048, 512), (15936, 512, 2048), (98, 144, 144), (98, 576, 144)]
CFG = ["128x128", "128x64", "64x128", "64x64", "auto"]
ENG = ["reg", "dma", "dma-norefill"]


def one(M, N, K, cfg, iters):
    """Run one shape/cfg `iters` times (for rocprofv3 --pmc / --kernel-trace runs)."""
    lib = _lib.load()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / K ** 0.5
    b = torch.randn(N, device=dev); r = torch.randn(M, N, device=dev); c = torch.empty(M, N, device=dev)
    for _ in range(iters):
        assert lib.cfm_debug_gemm_cfg_f32(cfg, a.data_ptr(), w.data_ptr(), b.data_ptr(), r.data_ptr(), 0.5,
                                          c.data_ptr(), M, N, K, st) == 0
    torch.cuda.synchronize()


def ablate():
    """Prices the refill path of the LDS-DMA engine: real kernel vs the same kernel without refills."""
    lib = _lib.load()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    for (M, N, K) in [(7968, 2048, 512), (7968, 512, 2048), (31872, 2048, 2048)]:
        a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / K ** 0.5
        b = torch.randn(N, device=dev); r = torch.randn(M, N, device=dev); c = torch.empty(M, N, device=dev)
        for tile in (0, 1, 3):
            line = f"{M}x{N}x{K} tile {CFG[tile]:>7s}: "
            for eng in (0, 1, 2):
                cfg = tile + 4 * eng

                def run():
                    assert lib.cfm_debug_gemm_cfg_f32(cfg, a.data_ptr(), w.data_ptr(), b.data_ptr(), r.data_ptr(), 0.5,
                                                      c.data_ptr(), M, N, K, st) == 0
                for _ in range(3):
                    run()
                torch.cuda.synchronize()
                best = 1e9
                for _ in range(5):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(20):
                        run()
                    e1.record()
                    torch.cuda.synchronize()
                    best = min(best, e0.elapsed_time(e1) / 20)
                line += f"| {ENG[eng]:>12s} {best * 1e3:7.1f}us {2.0 * M * N * K / best / 1e9:6.1f}TF "
            print(line, flush=True)


def main():
    if len(sys.argv) == 2 and sys.argv[1] == "ablate":
        return ablate()
    if len(sys.argv) >= 5:
        return one(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]),
                   int(sys.argv[5]) if len(sys.argv) > 5 else 10)
    lib = _lib.load()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    for (M, N, K) in SHAPES:
        a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / K ** 0.5
        b = torch.randn(N, device=dev); r = torch.randn(M, N, device=dev)
        ref = None
        line = f"{M:6d}x{N:5d}x{K:5d} "
        cfgs = [0, 1, 2, 3, -1]
        for ci, cfg in enumerate(cfgs):
            c = torch.empty(M, N, device=dev)

            def run():
                s = lib.cfm_debug_gemm_cfg_f32(cfg, a.data_ptr(), w.data_ptr(), b.data_ptr(), r.data_ptr(), 0.5,
                                               c.data_ptr(), M, N, K, st)
                assert s == 0, s
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            reps, inner = 5, 20
            best = 1e9
            for _ in range(reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(inner):
                    run()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / inner)
            if ref is None:
                ref = c.clone()
            else:
                assert torch.allclose(ref, c, rtol=1e-4, atol=1e-4), "configs must agree"
            name = "auto" if cfg < 0 else f"{ENG[cfg >> 2]}:{CFG[cfg & 3]}"
            line += f"| {name:>11s} {best * 1e3:7.1f}us {2.0 * M * N * K / best / 1e9:6.1f} "
        print(line, flush=True)


if __name__ == "__main__":
    main()
