#!/usr/bin/env python3
"""Stem conv2 forward on the 16-bit pipe: 128x128 (4-wave) vs 256x256 (8-wave) tiles, 16-bit h1, B = 32 and 64."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conformer_amd import _lib  # noqa: E402
from tools.kernel_table import time_us  # noqa: E402

lib = _lib.load(); dev = torch.device("cuda:0")
C, F1, T1 = 512, 39, 499
T2, F2 = (T1 - 1) // 2, (F1 - 1) // 2
for B in (32, 64):
    h1 = torch.randn(B, T1, F1, C, device=dev).relu().to(torch.bfloat16)
    w = (torch.randn(C, 9 * C, device=dev) / 68).to(torch.bfloat16)
    b2 = torch.randn(C, device=dev)
    outs = {}
    for h2_16 in (0, 1):
        for tile in (1, 0):
            h2 = torch.empty(B, T2, F2, C, device=dev, dtype=torch.bfloat16 if h2_16 else torch.float32)
            lib.cfm_debug_gemm_mfma16_force_tile(tile)
            st = lambda: torch.cuda.current_stream().cuda_stream
            fn = lambda: lib.cfm_subsample_conv2_relu_mfma16_f32(1, h1.data_ptr(), 1, w.data_ptr(), 1, b2.data_ptr(), h2.data_ptr(), h2_16, B, F1, T1, C, st())
            assert fn() == 0
            us = time_us(fn, 10)
            outs[(h2_16, tile)] = h2.float().clone()
            fl = 2.0 * B * T2 * F2 * C * 9 * C
            print(f"B={B} h2 {'bf16' if h2_16 else 'fp32'} tile {'128x128' if tile == 1 else 'auto (256x256)'}: {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s")
        lib.cfm_debug_gemm_mfma16_force_tile(0)
        d = (outs[(h2_16, 0)] - outs[(h2_16, 1)]).abs().max() / outs[(h2_16, 1)].abs().max()
        print(f"   max relative difference between the tilings: {float(d):.2e}")
