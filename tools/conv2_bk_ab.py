#!/usr/bin/env python3
"""The stem's implicit-GEMM conv2 (714 GFLOP at cfg-2: 26 % of the forward) with K-tile 16 vs 32, interleaved rounds in one process."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conformer_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
B, T1, F1, C = 32, 499, 39, 512
T2, F2 = (T1 - 1) // 2, (F1 - 1) // 2
lib = _lib.load()
h1 = torch.randn(B, T1, F1, C, device=dev).relu_()
w2 = torch.randn(C, C, 3, 3, device=dev) / (9 * C) ** 0.5
b2 = torch.randn(C, device=dev) * 0.1
w2p = ops.pack_conv2_weight(w2)
outs, times = {}, {16: [], 32: [], 0: []}
st = torch.cuda.current_stream().cuda_stream


def run(out):
    _lib.check(lib.cfm_subsample_conv2_relu_f32(h1.data_ptr(), w2p.data_ptr(), b2.data_ptr(), out.data_ptr(), B, F1, T1, C, st), "conv2")


for bk in (16, 32, 0):                                      # 0 = K-tile 16 with K walked in storage (tap-major) order
    lib.cfm_debug_set_conv2_bk(1); lib.cfm_debug_set_conv2_bk(16)
    lib.cfm_debug_set_conv2_bk(bk)
    outs[bk] = torch.empty(B, T2, F2 * C, device=dev)
    run(outs[bk]); run(outs[bk])
torch.cuda.synchronize()
print("rel-L2 difference 16 vs 32:", float((outs[16] - outs[32]).norm() / outs[16].norm()), " channel-chunk-major vs storage order:",
      float((outs[16] - outs[0]).norm() / outs[0].norm()))
for rnd in range(7):
    for bk in (16, 32, 0):
        lib.cfm_debug_set_conv2_bk(1); lib.cfm_debug_set_conv2_bk(16)
        lib.cfm_debug_set_conv2_bk(bk)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            run(outs[bk])
        e1.record()
        torch.cuda.synchronize()
        times[bk].append(e0.elapsed_time(e1) / 4)
lib.cfm_debug_set_conv2_bk(16); lib.cfm_debug_set_conv2_bk(1)
fl = 2.0 * B * T2 * F2 * C * 9 * C
for bk in (16, 32, 0):
    med = statistics.median(times[bk])
    print(f"{'K-tile 16, storage-order K walk' if bk == 0 else f'K-tile {bk}'}: median {med:.3f} ms  min {min(times[bk]):.3f} ms  {fl / med / 1e9:.1f} TFLOP/s = {fl / med / 1e9 / 157.3:.3f} of the fp32 MFMA peak")
