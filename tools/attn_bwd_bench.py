#!/usr/bin/env python3
"""Attention-core backward at cfg-2 / cfg-3 shapes: the fused flash kernel (fp32, and with the autocast operand rounding).
Prints JSON lines; `trace` prints the per-phase timeline of one wave.  (Round-2 record of the replaced materialising form,
measured with this tool before it was deleted: profiles/r02_attn_bwd_flash_vs_materialised.jsonl.)"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conformer_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


if len(sys.argv) > 1 and sys.argv[1] in ("trace", "trace16"):
    from conformer_amd import _lib
    import contextlib
    lp = sys.argv[1] == "trace16"
    amp = (lambda: torch.autocast("cuda", dtype=torch.bfloat16)) if lp else contextlib.nullcontext
    setter = "cfm_debug_attention_bwd_trace_mfma16" if lp else "cfm_debug_attention_bwd_trace_f32"
    B, T, d, H = 32, 249, 512, 8
    g = torch.Generator(device=dev).manual_seed(0)
    qkv = torch.randn(B, T, 3 * d, device=dev, generator=g) * 0.5
    pos = torch.randn(2 * T - 1, d, device=dev, generator=g) * 0.5
    u = torch.randn(H, d // H, device=dev, generator=g) * 0.1
    v = torch.randn(H, d // H, device=dev, generator=g) * 0.1
    L = torch.full((B,), T, dtype=torch.int64, device=dev)
    dy = torch.randn(B, T, d, device=dev, generator=g)
    with amp():
        ctx, lse = ops.relpos_attention_train(qkv, pos, u, v, L, H)
        for _ in range(3):
            ops.relpos_attention_bwd(qkv, pos, u, v, L, H, ctx, lse, dy)
        nq = (T + 31) // 32
        tr = torch.zeros(16 * nq, dtype=torch.int64, device=dev)
        getattr(_lib.load(), setter)(tr.data_ptr())
        ops.relpos_attention_bwd(qkv, pos, u, v, L, H, ctx, lse, dy)
        torch.cuda.synchronize()
        getattr(_lib.load(), setter)(None)
    s = tr.cpu().view(nq, 16)
    names = (["S", "band+skew+dW", "P,dS", "dS tiles", "dV,dK", "dQu+dQv", "dPband+dq+atomics", "wait-bar", "commit", "flush+bar", "prefetch"] if lp else
             ["S", "band+skew+dW", "P,dS", "dV,dK", "dQu", "dQv", "dPband", "wait-bar", "commit", "flush+bar", "prefetch"])
    print("tile | " + " | ".join(f"{n:>9s}" for n in names) + " | total   (ns; s_memrealtime 100 MHz)")
    for it in range(nq):
        dts = [(int(s[it, i + 1]) - int(s[it, i])) * 10 for i in range(11)]
        print(f"{it:4d} | " + " | ".join(f"{x:9d}" for x in dts) + f" | {sum(dts)}")
    sys.exit(0)

for B in (32, 64):
    T, d, H = 249, 512, 8
    g = torch.Generator(device=dev).manual_seed(0)
    qkv = torch.randn(B, T, 3 * d, device=dev, generator=g) * 0.5
    pos = torch.randn(2 * T - 1, d, device=dev, generator=g) * 0.5
    u = torch.randn(H, d // H, device=dev, generator=g) * 0.1
    v = torch.randn(H, d // H, device=dev, generator=g) * 0.1
    L = torch.full((B,), T, dtype=torch.int64, device=dev)
    dy = torch.randn(B, T, d, device=dev, generator=g)
    ctx, lse = ops.relpos_attention_train(qkv, pos, u, v, L, H)
    flops = 10.0 * B * T * T * d + 2.0 * 2 * B * T * T * d        # 5 products + the two positional ones (algorithmic, 2*MAC)
    row = {"B": B, "T": T, "d": d, "H": H}
    row["flash_f32_us"] = timeit(lambda: ops.relpos_attention_bwd(qkv, pos, u, v, L, H, ctx, lse, dy))
    with torch.autocast("cuda", dtype=torch.bfloat16):
        c16, l16 = ops.relpos_attention_train(qkv, pos, u, v, L, H)
        row["autocast_bf16_us"] = timeit(lambda: ops.relpos_attention_bwd(qkv, pos, u, v, L, H, c16, l16, dy))
    row["flash_f32_tflops_algorithmic"] = flops / row["flash_f32_us"] / 1e6
    print(json.dumps(row), flush=True)
