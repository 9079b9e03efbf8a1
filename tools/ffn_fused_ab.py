#!/usr/bin/env python3
"""The feed-forward sub-layer at cfg-2 shapes (7968 rows, d = 512, hidden 2048): one fused kernel vs hidden GEMM + residual GEMM
(folded LayerNorm on both sides), interleaved rounds in one process."""
import math
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conformer_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
M, d = (int(sys.argv[1]) if len(sys.argv) > 1 else 7968), 512
hid = 4 * d
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(M, d, device=dev, generator=g) + 0.3
lw, lb = 1 + 0.1 * torch.randn(d, device=dev, generator=g), 0.1 * torch.randn(d, device=dev, generator=g)
w1, b1 = torch.randn(hid, d, device=dev, generator=g) / math.sqrt(d), 0.1 * torch.randn(hid, device=dev, generator=g)
w2, b2 = torch.randn(d, hid, device=dev, generator=g) / math.sqrt(hid), 0.1 * torch.randn(d, device=dev, generator=g)
_, st = ops.layernorm(x, lw, lb, emit_stats=True)          # any producer: here one partial per row of x itself
xs = x.view(M, d // 32, 32)
st = torch.stack([xs.sum(-1), ((xs - xs.mean(-1, keepdim=True)) ** 2).sum(-1)], dim=-1).contiguous()
wf, bf, cs = ops.fold_layernorm(w1, b1, lw, lb)
wp = ops.ffn_pack(wf, w2)


def two():
    h = ops.linear_lnfold(x, st, wf, bf, cs, 1e-5, act="swish")
    return ops.linear_residual(h, w2, b2, x, 0.5, emit_stats=True)


def one():
    return ops.ffn_fused(x, st, wp, bf, cs, b2, 0.5, 1e-5, emit_stats=True)


def one_ln():
    return ops.ffn_fused(x, st, wp, bf, cs, b2, 0.5, 1e-5, emit_stats=True, closing_ln=(lw, lb, 1e-5))


ya, sa = two()
yb, sb = one()
print("rel-L2 fused vs two-GEMM:", float((ya - yb).norm() / ya.norm()), " stats:", float((sa - sb).norm() / sa.norm()))
from conformer_amd import _lib as _l  # noqa: E402


def dbg(v):
    def f():
        _l.load().cfm_debug_ffn_variant(v)
        r = one()
        _l.load().cfm_debug_ffn_variant(0)
        return r
    return f


def layout(pad, rotate):
    """variant with its own packing: pad between tiles (16-byte units), slice rotation per workgroup"""
    _l.load().cfm_debug_ffn_layout(pad, rotate)
    wpl = ops.ffn_pack(wf, w2)
    _l.load().cfm_debug_ffn_layout(0, 0)

    def f():
        _l.load().cfm_debug_ffn_layout(pad, rotate)
        r = ops.ffn_fused(x, st, wpl, bf, cs, b2, 0.5, 1e-5, emit_stats=True)
        _l.load().cfm_debug_ffn_layout(0, 0)
        return r
    return f


variants = {"two GEMMs": two, "fused": one, "fused + closing LN": one_ln, "fused, no weight loads (dbg)": dbg(1)}
for pad, rotate in ((0, 0), (0, 1), (272, 0), (16, 1), (64, 1), (256, 1), (1040, 1)):
    variants[f"fused, pad {pad * 16} B, rotate {rotate}"] = layout(pad, rotate)
for k, fn in variants.items():
    if k.startswith("fused, pad"):
        print(k, "vs default:", float((fn()[0] - yb).norm() / yb.norm()))
times = {k: [] for k in variants}
for rnd in range(9):
    for k, fn in variants.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        times[k].append(e0.elapsed_time(e1) / 10 * 1e3)
fl = 4.0 * M * d * hid
for k in variants:
    med = statistics.median(times[k])
    print(f"{k:34s}: median {med:7.1f} us  min {min(times[k]):7.1f} us  {fl / med / 1e6:6.1f} TFLOP/s = {fl / med / 1e6 / 157.3:.3f} of the fp32 MFMA peak")
if len(sys.argv) > 2 and sys.argv[2] == "trace":
    # s_memrealtime stamps (100 MHz) of wave 0 of workgroups 0 and 128: prologue, (stage 1 + Swish, stage 2) per slice, tile sum, epilogue
    from conformer_amd import _lib
    lib = _lib.load()
    for v, detail in ((0, 0), (1, 0), (0, 1)):
        tr = torch.zeros(128, dtype=torch.int64, device=dev)
        fn = dbg(v)
        for _ in range(3):
            fn()
        lib.cfm_debug_ffn_trace(tr.data_ptr(), detail)
        fn()
        torch.cuda.synchronize()
        lib.cfm_debug_ffn_trace(None, 0)
        ns = hid // 128
        print("variant", v, "(no weight loads in the main loop)" if v else "", "per-slice stamps" if detail else "")
        for blk in (0, 1):
            t = tr[64 * blk:64 * blk + 4 + 2 * ns].tolist()
            ck = tr[64 * blk + 60:64 * blk + 63].tolist()
            loop_ns = 10 * (ck[2] - t[1])
            print(f"workgroup {128 * blk}: prologue {10 * (t[1] - t[0])} ns, main loop {loop_ns} ns = {ck[1] - ck[0]} shader clocks -> "
                  f"{(ck[1] - ck[0]) / loop_ns:.3f} GHz; {(ck[1] - ck[0]) / (16 * ns * 32):.1f} clocks per MFMA; tile sum "
                  f"{10 * (t[2 + 2 * ns] - ck[2])} ns, epilogue {10 * (t[3 + 2 * ns] - t[2 + 2 * ns])} ns, total {10 * (t[3 + 2 * ns] - t[0])} ns")
            if detail:
                u = [10 * (x_ - t[0]) for x_ in t]
                prev = u[1]
                for s_ in range(ns):
                    a_, b_ = u[2 + 2 * s_], u[3 + 2 * s_]
                    if s_ in (0, 1, ns - 1):
                        print(f"   slice {s_:2d}: stage 1 + Swish {a_ - prev:6d} ns | stage 2 {b_ - a_:6d} ns")
                    prev = b_
