#!/usr/bin/env python3
"""Weight-gradient GEMM sites of a Conformer-L layer under bf16 autocast: the dedicated kernel (gemm_dw16_impl.h, with the
bias gradient fused) against the general backward kernel + column-sum pass, for fp32 / bf16 stored operands.
CFM_DW16_SPLITS=n forces the split-K factor of the dedicated kernel (tuning)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conformer_amd import _lib, ops  # noqa: E402
from tools.kernel_table import time_us  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    lib = _lib.load()
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 15936
    g = torch.Generator(device=dev).manual_seed(0)
    R = lambda *s: torch.randn(*s, device=dev, generator=g)
    for name, N, K in [("FFN hidden", 2048, 512), ("FFN out", 512, 2048), ("QKV", 1536, 512), ("out/pw2", 512, 512), ("pw1", 1024, 512)]:
        dy, x = R(M, N), R(M, K)
        line = f"{name:10s} dW {N}x{K} (M={M}): "
        for dy16, x16 in [(0, 0), (0, 1), (1, 1)]:
            a = dy.to(torch.bfloat16) if dy16 else dy
            b = x.to(torch.bfloat16) if x16 else x
            dw = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev)

            def new():
                st = torch.cuda.current_stream().cuda_stream
                assert lib.cfm_linear_bwd_weight_mfma16_f32(1, a.data_ptr(), dy16, N, b.data_ptr(), x16, K, dw.data_ptr(), K, db.data_ptr(),
                                                            N, K, M, 1.0, st) == 0
            t_new = time_us(new, 20)
            flops = 2.0 * M * N * K
            line += f"| dY{'16' if dy16 else '32'} X{'16' if x16 else '32'}: {t_new:6.1f} us {flops / t_new / 1e6:5.0f} TF "
            if not dy16:
                def old():
                    ops.gemm_bwd(dy, True, b, True, N, K, M, allow_split=True, out=dw, prec=1, b16=bool(x16))
                    ops.colsum(dy, 1.0, out=db)
                line += f"(general+colsum {time_us(old, 20):6.1f}) "
        print(line, flush=True)


def trace(N, K, M, dy16, x16):
    dev = torch.device("cuda:0")
    lib = _lib.load()
    dy, x = torch.randn(M, N, device=dev), torch.randn(M, K, device=dev)
    a = dy.to(torch.bfloat16) if dy16 else dy
    b = x.to(torch.bfloat16) if x16 else x
    dw = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    run = lambda: lib.cfm_linear_bwd_weight_mfma16_f32(1, a.data_ptr(), dy16, N, b.data_ptr(), x16, K, dw.data_ptr(), K, db.data_ptr(), N, K, M, 1.0, st)
    for _ in range(3):
        run()
    tr = torch.zeros(128, dtype=torch.int64, device=dev)
    lib.cfm_debug_dw16_trace(tr.data_ptr())
    run()
    torch.cuda.synchronize()
    lib.cfm_debug_dw16_trace(None)
    t = tr.cpu().view(2, 64)
    for w in range(2):
        r = [int(v) for v in t[w]]
        ks = [(r[2 + i] - r[1 + i]) * 10 for i in range(58) if r[2 + i] and r[1 + i]]
        print(f"wg{w}: prologue {(r[1] - r[0]) * 10} ns | K-tiles {ks} | epilogue issue {(r[60] - max(r[1:60])) * 10} drain {(r[61] - r[60]) * 10} | total {(r[61] - r[0]) * 10} ns")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "trace":
        trace(*(int(v) for v in sys.argv[2:7]))
    else:
        main()
