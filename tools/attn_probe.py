#!/usr/bin/env python3
"""Launch the rel-pos attention forward a few times at cfg-2 shapes (for rocprofv3 --pmc passes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conformer_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
B, T, d, H = 32, 249, 512, 8
g = torch.Generator(device=dev).manual_seed(0)
qkv = torch.randn(B, T, 3 * d, device=dev, generator=g)
pos = torch.randn(2 * T - 1, d, device=dev, generator=g)
u = torch.randn(H, d // H, device=dev, generator=g) * 0.1
v = torch.randn(H, d // H, device=dev, generator=g) * 0.1
L = torch.full((B,), T, dtype=torch.int64, device=dev)
for _ in range(6):
    ops.relpos_attention(qkv, pos, u, v, L, H)
torch.cuda.synchronize()
if len(sys.argv) > 1 and sys.argv[1] == "trace":
    from conformer_amd import _lib
    lib = _lib.load()
    ntiles = (T + 31) // 32
    tr = torch.zeros(16 * ntiles, dtype=torch.int64, device=dev)
    ctx = torch.empty(B, T, d, device=dev)
    base = qkv.data_ptr()
    _lib.check(lib.cfm_debug_attention_trace_f32(base, base + 4 * d, base + 8 * d, 3 * d, pos.data_ptr(), d, u.data_ptr(),
                                                 v.data_ptr(), L.data_ptr(), ctx.data_ptr(), d, B, T, H, d // H, tr.data_ptr(),
                                                 torch.cuda.current_stream().cuda_stream), "trace")
    torch.cuda.synchronize()
    s = tr.cpu().view(ntiles, 16)[:, :9]
    names = ["stage next K/V", "content mfma", "band mfma+spill", "skew reads", "select+add", "softmax", "PV", "barrier"]
    print("tile | " + " | ".join(names) + " | total   (ns, 100 MHz clock)")
    for kt in range(ntiles):
        dts = [(int(s[kt, i + 1]) - int(s[kt, i])) * 10 for i in range(8)]
        print(f"{kt:4d} | " + " | ".join(f"{x:6d}" for x in dts) + f" | {sum(dts)}")
print("done")
