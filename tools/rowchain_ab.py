#!/usr/bin/env python3
"""The three row-local chains of a block at cfg-2 shapes (7968 rows, d = 512) against the kernels they replace, interleaved rounds."""
import math
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conformer_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
M, d = (int(sys.argv[1]) if len(sys.argv) > 1 else 7968), 512
g = torch.Generator(device=dev).manual_seed(0)
R = lambda *s: torch.randn(*s, device=dev, generator=g)  # noqa: E731
x, ctx, c = R(M, d) + 0.3, R(M, d), R(M, d)
lnw, lnb = 1 + 0.1 * R(d), 0.1 * R(d)
w1, b1, w2, b2 = R(4 * d, d) / math.sqrt(d), 0.1 * R(4 * d), R(d, 4 * d) / math.sqrt(4 * d), 0.1 * R(d)
wq, bq = R(3 * d, d) / math.sqrt(d), 0.1 * R(3 * d)
wo, bo = R(d, d) / math.sqrt(d), 0.1 * R(d)
wg, bg = R(2 * d, d) / math.sqrt(d), 0.1 * R(2 * d)
xs = x.view(M, d // 32, 32)
st = torch.stack([xs.sum(-1), ((xs - xs.mean(-1, keepdim=True)) ** 2).sum(-1)], dim=-1).contiguous()
wf, bf, cs = ops.fold_layernorm(w1, b1, lnw, lnb)
ffn = (ops.ffn_pack(wf, w2), bf, cs)
wqf, bqf, csq = ops.fold_layernorm(wq, bq, lnw, lnb)
wq_p = ops.rowgemm_pack(wqf)
wgf, bgf, csg = ops.fold_layernorm(wg, bg, lnw, lnb)
wg_p = ops.rowgemm_pack(wgf, glu=True)
wo_p = ops.rowgemm_pack(wo)


def k1():
    return ops.rowchain_ffn_qkv(x, st, ffn, b2, 0.5, 1e-5, wq_p, bqf, csq, 1e-5)


def k1_ref():
    y, s2 = ops.ffn_fused(x, st, ffn[0], bf, cs, b2, 0.5, 1e-5, emit_stats=True)
    return y, ops.linear_lnfold(y, s2, wqf, bqf, csq, 1e-5)


def k2():
    return ops.rowchain_out_glu(ctx, wo_p, bo, x, wg_p, bgf, csg, 1e-5)


def k2_ref():
    y, s2 = ops.linear_residual(ctx, wo, bo, x, 1.0, emit_stats=True)
    return y, ops.linear_lnfold(y, s2, wgf, bgf, csg, 1e-5, glu=True)


def k3():
    return ops.rowchain_pw2_ffn_ln(c, wo_p, bo, x, ffn, b2, 0.5, 1e-5, (lnw, lnb, 1e-5), want_stats=True)


def k3_ref():
    y, s2 = ops.linear_residual(c, wo, bo, x, 1.0, emit_stats=True)
    return ops.ffn_fused(y, s2, ffn[0], bf, cs, b2, 0.5, 1e-5, emit_stats=True, closing_ln=(lnw, lnb, 1e-5))


pairs = {"K1 FFN1 + qkv": (k1, k1_ref, 4.0 * M * d * 4 * d + 2.0 * M * d * 3 * d), "K2 out_proj + GLU": (k2, k2_ref, 2.0 * M * d * d + 2.0 * M * d * 2 * d),
         "K3 pw2 + FFN2 + LN": (k3, k3_ref, 2.0 * M * d * d + 4.0 * M * d * 4 * d)}
for name, (f, r, fl) in pairs.items():
    a, b = f(), r()
    print(name, "chain vs kernels: rel-L2", [float((u - v).norm() / v.norm()) for u, v in zip(a, b) if u is not None])
times = {}
for rnd in range(7):
    for name, (f, r, fl) in pairs.items():
        for tag, fn in (("chain", f), ("kernels", r)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record()
            torch.cuda.synchronize()
            times.setdefault((name, tag), []).append(e0.elapsed_time(e1) / 10 * 1e3)
for name, (f, r, fl) in pairs.items():
    for tag in ("chain", "kernels"):
        med = statistics.median(times[name, tag])
        print(f"{name:22s} {tag:8s}: median {med:7.1f} us  {fl / med / 1e6:6.1f} TFLOP/s = {fl / med / 1e6 / 157.3:.3f} of the fp32 MFMA peak")
