#!/usr/bin/env python3
"""Times the split-operand fp32 GEMM (csrc/gemm_split.hip) against the native fp32 MFMA GEMM at the hot-path shapes."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conformer_amd import ops  # noqa: E402
from tools.kernel_table import time_us  # noqa: E402

SHAPES = [("ffn_up swish", 7968, 2048, 512, "swish"), ("ffn_down resid", 7968, 512, 2048, "resid"),
          ("qkv", 7968, 1536, 512, "none"), ("out_proj resid", 7968, 512, 512, "resid"), ("pw1 glu", 7968, 512, 512, "glu"),
          ("input linear", 7968, 512, 9728, "none")]


def main():
    dev = torch.device("cuda:0")
    modes = sys.argv[1].split(",") if len(sys.argv) > 1 else ["native", "bf16x6", "bf16x3"]
    out = []
    for name, m, n, k, epi in SHAPES:
        a = torch.randn(m, k, device=dev)
        w = torch.randn(2 * n if epi == "glu" else n, k, device=dev) / k ** 0.5
        b = torch.randn(w.shape[0], device=dev)
        r = torch.randn(m, n, device=dev)
        row = {"site": name, "M": m, "N": n, "K": k}
        for mode in modes:
            if mode == "autocast":
                ops.set_fp32_matmul("native")
                fns = {"swish": lambda: ops.linear(a, w, b, "swish"), "none": lambda: ops.linear(a, w, b),
                       "resid": lambda: ops.linear_residual(a, w, b, r, 0.5), "glu": lambda: ops.linear_glu(a, w, b)}
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    us = time_us(fns[epi], 20)
                row[mode] = {"us": round(us, 1), "tflops_equiv": round(2.0 * m * w.shape[0] * k / us / 1e6, 1)}
                continue
            ops.set_fp32_matmul(mode)
            fn = {"swish": lambda: ops.linear(a, w, b, "swish"), "none": lambda: ops.linear(a, w, b),
                  "resid": lambda: ops.linear_residual(a, w, b, r, 0.5), "glu": lambda: ops.linear_glu(a, w, b)}[epi]
            us = time_us(fn, 20)
            row[mode] = {"us": round(us, 1), "tflops_equiv": round(2.0 * m * w.shape[0] * k / us / 1e6, 1)}
        out.append(row)
        print(json.dumps(row), flush=True)
    ops.set_fp32_matmul("native")


if __name__ == "__main__":
    main()
