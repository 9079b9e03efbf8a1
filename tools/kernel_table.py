#!/usr/bin/env python3
"""Per-kernel roofline table at BASELINE cfg-2 shapes (B=32, T'=249, d=512, H=8, K=31): every kernel class of the hot
path timed in isolation with HIP events on the launch stream, priced against its own roof -- HBM (8 TB/s) for the
streaming kernels with ALGORITHMIC bytes (each operand read once, each result written once), fp32 MFMA (157.3 TFLOP/s)
for attention.  SURVEY.md 8(d): "LN, dwconv+BN+Swish, softmax/attention, rel-PE, log-mel, SpecAugment are HBM-bound and
are graded on GB/s".  Prints a markdown table and one JSON line.

    python tools/kernel_table.py [--iters 50] > profiles/rNN_kernel_table.md
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conformer_amd import ops  # noqa: E402

HBM, MFMA32 = 8000.0, 157.3       # GB/s, TFLOP/s (MI355X_MICROARCH.md)


def time_us(fn, iters):
    """Mean device time of one call: `iters` calls are captured into a hipGraph and the graph is replayed, so the Python /
    ctypes issue cost (~10 us per call, more than several of these kernels take) is not what gets measured."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        fn()
        side.synchronize()
        with torch.cuda.graph(graph, stream=side):
            for _ in range(iters):
                fn()
    torch.cuda.synchronize()
    graph.replay()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        graph.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    B, T, d, H, K = 32, 249, 512, 8, 31
    N, P, dh = B * T, 2 * T - 1, 64
    f = 4.0
    g = torch.Generator(device=dev).manual_seed(0)
    R = lambda *s: torch.randn(*s, device=dev, generator=g)
    x, dy, res = R(B, T, d), R(B, T, d), R(B, T, d)
    w, b = R(d), R(d)
    qkv, pos, u, v = R(B, T, 3 * d), R(P, d), R(H, dh) * 0.1, R(H, dh) * 0.1
    L = torch.full((B,), T, dtype=torch.int64, device=dev)
    wd, bd = R(d, 1, K) * 0.2, R(d) * 0.1
    bnw, bnb, bnm, bnv = R(d), R(d), R(d) * 0.1, torch.rand(d, device=dev, generator=g) + 0.5
    z2 = R(B, T, 2 * d)
    y_ln, mean, rstd = ops.layernorm_train(x, w, b)
    ctx, lse = ops.relpos_attention_train(qkv, pos, u, v, L, H)
    div = torch.exp(torch.arange(0, d, 2, device=dev, dtype=torch.float32) * (-9.210340371976184 / d)).view(1, -1)
    mel = R(B, 80, 1000)
    wave = R(B, 159840)
    from conformer_amd.frontend import ConformerAudioFrontend
    fe = ConformerAudioFrontend(device=dev)
    bands = torch.tensor([[2, 100, 135], [2, 400, 410], [1, 10, 25]], dtype=torch.int32, device=dev)
    from conformer_amd import _lib
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    params = [torch.nn.Parameter(R(2048, 512)) for _ in range(16)]
    for p in params:
        p.grad = torch.randn_like(p)
    from conformer_amd.optim import FusedAdam
    adam = FusedAdam(params, lr=1e-5)
    logits = R(B, T, 370)
    from conformer_amd.decode import greedy_ctc_decode
    ctc_tg = torch.randint(1, 370, (B, 40), device=dev)
    ctc_il = torch.full((B,), T, dtype=torch.int64, device=dev)
    ctc_tl = torch.full((B,), 40, dtype=torch.int64, device=dev)
    ctc_state = ops.ctc_loss_forward(logits, ctc_tg, ctc_il, ctc_tl, 0)[1]
    ctc_g = torch.ones((), device=dev)

    def specaug():
        _lib.check(lib.cfm_specaugment_apply_f32(mel.data_ptr(), B, 80, 1000, bands.data_ptr(), 3, 0.0, ops._stream()), "specaug")

    rows = [
        # name, fn, algorithmic bytes, flops (0 = HBM-priced)
        ("layernorm fwd (inference)", lambda: ops.layernorm(x, w, b), 2 * N * d * f, 0),
        ("layernorm fwd (+mean,rstd)", lambda: ops.layernorm_train(x, w, b), 2 * N * d * f, 0),
        ("layernorm bwd dx (+residual grad)", lambda: ops.layernorm_bwd(x, w, dy, mean, rstd, dres=res)[0], 0, 0),
        ("colsum (bias grad) 7968x2048", lambda: ops.colsum(z2.view(N, 2 * d)[:, :], out=torch.zeros(2 * d, device=dev)), N * 2 * d * f, 0),
        ("dwconv+BN+swish fwd", lambda: ops.dwconv_bn_swish(x, wd, bd, bnw, bnb, bnm, bnv), 2 * N * d * f, 0),
        ("dwconv BN batch stats", lambda: ops.dwconv_bn_batch_stats(x, wd, bd, None, None), N * d * f, 0),
        ("dwconv+BN+swish bwd (eval stats)", lambda: ops.dwconv_bn_swish_bwd(x, dy, wd, bd, bnw, bnb, bnm, bnv), 3 * N * d * f, 0),
        ("GLU fwd", lambda: ops.glu_fwd(z2), 3 * N * d * f, 0),
        ("GLU bwd", lambda: ops.glu_bwd(z2, dy), 5 * N * d * f, 0),
        ("rel-pos table (2T'-1, d)", lambda: ops.relpos_table(div, T), P * d * f, 0),
        ("rel-pos attention fwd", lambda: ops.relpos_attention(qkv, pos, u, v, L, H), 4 * N * d * f, 6.0 * B * T * T * d),
        ("rel-pos attention fwd (+lse)", lambda: ops.relpos_attention_train(qkv, pos, u, v, L, H), 4 * N * d * f, 6.0 * B * T * T * d),
        ("rel-pos attention bwd (fused flash kernel)", lambda: ops.relpos_attention_bwd(qkv, pos, u, v, L, H, ctx, lse, dy), 8 * N * d * f,
         15.0 * B * T * T * d),
        ("log-mel front end (B=32, 10 s)", lambda: fe.mel_spectrogram(wave), (B * 159840 + B * 80 * 1000) * f, 0),
        ("SpecAugment apply (3 bands)", specaug, 0, 0),
        ("fused Adam (16 x 1M params)", lambda: adam.step(), 16 * 2048 * 512 * 7 * f, 0),
        ("greedy CTC decode (B,T',370)", lambda: greedy_ctc_decode(logits, 0, 1), B * T * 370 * f, 0),
        ("CTC loss fwd: lse + alpha lattice (40 labels)", lambda: ops.ctc_loss_forward(logits, ctc_tg, ctc_il, ctc_tl, 0),
         B * T * 370 * f, 0),
        ("CTC loss bwd: beta lattice + logits grad", lambda: ops.ctc_loss_backward(ctc_state, ctc_g), 2 * B * T * 370 * f, 0),
    ]
    # layernorm bwd dx: x, dy, dres read + dx written
    rows[2] = (rows[2][0], rows[2][1], 4 * N * d * f, 0)
    # SpecAugment touches only the banded elements: 45 frames x 80 + 15 bins x 1000, written once
    rows[14] = (rows[14][0], rows[14][1], B * (45 * 80 + 15 * 1000) * f, 0)
    out = []
    print("| kernel (cfg-2 shapes) | time µs | algorithmic MB | GB/s | of HBM 8 TB/s | TFLOP/s | of fp32 MFMA |")
    print("|---|---|---|---|---|---|---|")
    for name, fn, nbytes, flops in rows:
        us = time_us(fn, args.iters)
        gbs = nbytes / us / 1e3
        tf = flops / us / 1e6
        out.append(dict(kernel=name, us=us, alg_bytes=nbytes, gbs=gbs, hbm_frac=gbs / HBM, tflops=tf, mfma_frac=tf / MFMA32))
        print(f"| {name} | {us:.1f} | {nbytes / 1e6:.1f} | {gbs:.0f} | {gbs / HBM:.2f} | "
              f"{(f'{tf:.1f}' if flops else '-')} | {(f'{tf / MFMA32:.2f}' if flops else '-')} |")
    print()
    print("Timing: 20 back-to-back calls captured in a hipGraph, replayed 3x, best mean.  `colsum`, `layernorm bwd`, `dwconv bwd` "
          "and `attention bwd` rows include the zero-fill of their accumulators (and every helper kernel of the op).")
    print()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
