#!/usr/bin/env python3
"""Decoder forward (LSTM 512->640, Swish, BatchNorm eval, Linear 640->370) at cfg-2 shapes: gfx950 kernels vs the stock
PyTorch-ROCm modules (MIOpen LSTM) the reference would run.  Wall time per call incl. host issue, and device time."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from model.modules.decoder import Decoder  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    dec = Decoder(370, 512, 640, 1).to(dev).eval()
    x = torch.randn(B, 249, 512, device=dev)
    L = torch.full((B,), 249, dtype=torch.int64, device=dev)
    L[B // 2:] = 200
    res = {}
    with torch.no_grad():
        for name, fn in (("hip", lambda: dec.fused(x, L)), ("stock_miopen", lambda: torch.nn.Module.__call__(dec, x, L))):
            if name == "stock_miopen":
                dec._hip_eligible = lambda _x: False
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                y = fn()
            e1.record()
            torch.cuda.synchronize()
            res[name] = {"wall_ms": (time.perf_counter() - t0) / 10 * 1e3, "device_ms": e0.elapsed_time(e1) / 10}
            res[name + "_out"] = y
    err = float((res.pop("hip_out") - res.pop("stock_miopen_out")).abs().max())
    # training direction: forward + backward through the decoder (LSTM BPTT, Swish+BN train, vocabulary GEMM)
    dec._hip_eligible = type(dec)._hip_eligible.__get__(dec)
    dec.train()
    xg = x.clone().requires_grad_(True)
    w = torch.randn(B, 249, 370, device=dev)
    for name in ("hip_train", "stock_train"):
        if name == "stock_train":
            dec._hip_eligible = lambda _x: False
        def step():
            for p in dec.parameters():
                p.grad = None
            (dec(xg, L) * w).sum().backward()
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(5):
            step()
        e1.record()
        torch.cuda.synchronize()
        res[name] = {"wall_ms": (time.perf_counter() - t0) / 5 * 1e3, "device_ms": e0.elapsed_time(e1) / 5}
    # CTC loss on the logits (evaluation.py:12-16): lattice kernels vs torch's log_softmax + ctc_loss on the same device
    from conformer_amd.evaluation import ConformerCriterion
    crit = ConformerCriterion(0)
    logits = torch.randn(B, 249, 370, device=dev)
    tg = torch.randint(1, 370, (B, 40), device=dev)
    tl = torch.full((B,), 40, dtype=torch.int64, device=dev)
    def ctc_hip():
        lg = logits.detach().requires_grad_(True)
        crit.ctc_loss(lg, tg, L, tl).backward()
        return lg.grad
    def ctc_torch():
        lg = logits.detach().requires_grad_(True)
        torch.nn.functional.ctc_loss(lg.log_softmax(-1).transpose(0, 1), tg, L, tl, blank=0, zero_infinity=True).backward()
        return lg.grad
    for name, fn in (("ctc_hip_fwd_bwd", ctc_hip), ("ctc_torch_fwd_bwd", ctc_torch)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(10):
            g = fn()
        e1.record()
        torch.cuda.synchronize()
        res[name] = {"wall_ms": (time.perf_counter() - t0) / 10 * 1e3, "device_ms": e0.elapsed_time(e1) / 10}
        res[name + "_g"] = g
    res["ctc_grad_rel_l2_vs_torch"] = float((res["ctc_hip_fwd_bwd_g"] - res["ctc_torch_fwd_bwd_g"]).norm() /
                                            res.pop("ctc_torch_fwd_bwd_g").norm())
    res.pop("ctc_hip_fwd_bwd_g")
    print(json.dumps({"what": f"decoder forward B={B} T'=249 512->640->370, eval", **res, "max_abs_diff": err}))


if __name__ == "__main__":
    main()
