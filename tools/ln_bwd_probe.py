#!/usr/bin/env python3
"""LayerNorm backward: fused (dx + dgamma/dbeta in one pass) vs the two-kernel form, at the cfg-2 / cfg-3 row counts."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conformer_amd import _lib  # noqa: E402
from tools.kernel_table import time_us  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    lib = _lib.load()
    st_ = lambda: torch.cuda.current_stream().cuda_stream
    for rows in (7968, 15936):
        d = 512
        x, dy, dres = (torch.randn(rows, d, device=dev) for _ in range(3))
        g = torch.randn(d, device=dev)
        mean, rstd = torch.randn(rows, device=dev), torch.rand(rows, device=dev) + 0.5
        dx = torch.empty_like(x); dw = torch.zeros(d, device=dev); db = torch.zeros(d, device=dev)
        P = lambda t: t.data_ptr()
        two = lambda: (lib.cfm_layernorm_bwd_dx_f32(P(x), P(g), P(dy), P(mean), P(rstd), P(dres), P(dx), rows, d, st_()),
                       lib.cfm_layernorm_bwd_params_f32(P(x), P(dy), P(mean), P(rstd), P(dw), P(db), rows, d, st_()))
        nws = int(lib.cfm_layernorm_bwd_workspace_bytes(rows, d)); ws = torch.empty(nws // 4, device=dev)
        one = lambda: lib.cfm_layernorm_bwd_f32(P(x), P(g), P(dy), P(mean), P(rstd), P(dres), P(dx), P(dw), P(db), rows, d, P(ws), nws, st_())
        print(f"rows {rows}: two kernels {time_us(two, 20):.1f} us, fused {time_us(one, 20):.1f} us "
              f"(4 x {rows * d * 4 / 1e6:.0f} MB of traffic)")


if __name__ == "__main__":
    main()
