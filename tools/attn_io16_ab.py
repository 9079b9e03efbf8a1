#!/usr/bin/env python3
"""Inference under autocast, one MHSA layer at cfg-2 (B=32, T'=249, d=512, 8 heads): fp32 vs 16-bit projections / context."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conformer_amd import ops  # noqa: E402
from tools.kernel_table import time_us  # noqa: E402
from model.utils.attention import MultiHeadSelfAttentionModule  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
B, T, d, H = 32, 249, 512, 8
m = MultiHeadSelfAttentionModule(d, H).to(dev).eval()
a = m.attention
x = torch.randn(B, T, d, device=dev)
table = ops.relpos_table(torch.exp(torch.arange(0, d, 2, device=dev).float() * (-math.log(10000.0) / d)), T)
with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
    xn = ops.layernorm(x, m.layer_norm.weight, m.layer_norm.bias, m.layer_norm.eps, for_gemm=True)
    w, b = a._qkv_params()
    pos = ops.linear(table, a.pos_proj.weight, a.pos_proj.bias)
    for rep in range(2):
        for fg in (False, True):
            qkv = ops.linear(xn, w, b, for_gemm=fg)
            ctx = ops.relpos_attention(qkv, pos, a.content_bias, a.position_bias, None, H, for_gemm=fg)
            t1 = time_us(lambda: ops.linear(xn, w, b, for_gemm=fg), 20)
            t2 = time_us(lambda: ops.relpos_attention(qkv, pos, a.content_bias, a.position_bias, None, H, for_gemm=fg), 20)
            t3 = time_us(lambda: ops.linear_residual(ctx, a.out_proj.weight, a.out_proj.bias, x, 1.0), 20)
            print(f"16-bit io {fg}: QKV GEMM {t1:6.1f} us | attention {t2:6.1f} us | out-proj {t3:6.1f} us | sum {t1 + t2 + t3:6.1f}", flush=True)
    qkv32, qkv16 = ops.linear(xn, w, b), ops.linear(xn, w, b, for_gemm=True)
    for name, q, fg in (("fp32 qkv, fp32 ctx", qkv32, False), ("fp32 qkv, 16-bit ctx", qkv32, True), ("16-bit qkv, fp32 ctx", qkv16, False),
                        ("16-bit qkv, 16-bit ctx", qkv16, True)) * 2:
        print(f"attention alone, {name}: {time_us(lambda: ops.relpos_attention(q, pos, a.content_bias, a.position_bias, None, H, for_gemm=fg), 20):6.1f} us", flush=True)
