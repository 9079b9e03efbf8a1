#!/usr/bin/env python3
"""N = 512 products of the B = 32 forward on the 16-bit pipe: 64x64 vs 128x64 (4-wave) tiles."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conformer_amd import _lib, ops  # noqa: E402
from tools.kernel_table import time_us  # noqa: E402

lib = _lib.load(); dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
R = lambda *s: torch.randn(*s, device=dev, generator=g)
with torch.autocast("cuda", dtype=torch.bfloat16):
    for M in (7968, 3984):
        for name, N, K, a16 in [("FFN out (16-bit A)", 512, 2048, True), ("attn out (fp32 A)", 512, 512, False), ("pw2 (16-bit A)", 512, 512, True)]:
            a, w, b, r = R(M, K), R(N, K), R(N), R(M, N)
            if a16:
                a = a.to(torch.bfloat16)
            fn = lambda: ops.linear_residual(a, w, b, r, 0.5)
            line, outs = f"{name:20s} {M}x{N}x{K}: ", []
            for tile, lab in ((5, "64x64"), (4, "128x64"), (0, "auto")):
                lib.cfm_debug_gemm_mfma16_force_tile(tile)
                outs.append(fn().clone())
                line += f"| {lab} {time_us(fn, 20):6.1f} us "
            lib.cfm_debug_gemm_mfma16_force_tile(0)
            print(line, "| equal:", bool(torch.equal(outs[0], outs[1])), flush=True)
