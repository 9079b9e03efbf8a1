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
    raw = tr.cpu().view(ntiles, 16)
    t_in, t_pro, t_loop, t_out = (int(raw[0, i]) for i in (9, 10, 11, 12))
    print(f"wave 0 of workgroup (0,0): prologue {(t_pro - t_in) * 10} ns | key loop {(t_loop - t_pro) * 10} ns | "
          f"normalise + store {(t_out - t_loop) * 10} ns | total {(t_out - t_in) * 10} ns")
    s = raw[:, :9]
    names = ["stage next K/V", "content mfma", "band mfma+spill", "skew reads", "select+add", "softmax", "PV", "barrier"]
    print("tile | " + " | ".join(names) + " | total   (ns, 100 MHz clock)")
    for kt in range(ntiles):
        dts = [(int(s[kt, i + 1]) - int(s[kt, i])) * 10 for i in range(8)]
        print(f"{kt:4d} | " + " | ".join(f"{x:6d}" for x in dts) + f" | {sum(dts)}")
if len(sys.argv) > 1 and sys.argv[1] == "trace8":
    # 8-wave staggered form: per interval, when wave 0 (group A) / wave 4 (group B) start and finish their phase
    from conformer_amd import _lib
    lib = _lib.load()
    lib.cfm_debug_set_attention_waves(8)
    ntiles = (T + 31) // 32
    tr = torch.zeros(16 * ntiles, dtype=torch.int64, device=dev)
    ctx = torch.empty(B, T, d, device=dev)
    base = qkv.data_ptr()
    _lib.check(lib.cfm_debug_attention_trace_f32(base, base + 4 * d, base + 8 * d, 3 * d, pos.data_ptr(), d, u.data_ptr(),
                                                 v.data_ptr(), L.data_ptr(), ctx.data_ptr(), d, B, T, H, d // H, tr.data_ptr(),
                                                 torch.cuda.current_stream().cuda_stream), "trace")
    torch.cuda.synchronize()
    lib.cfm_debug_set_attention_waves(0)
    t = tr.cpu().tolist()
    t0 = t[0]
    print("interval | A: start  phase-end (what) | B: start  phase-end (what)   [ns from the first stamp]")
    for iv in range(2 * ntiles + 1):
        a0, a1, b0, b1 = (10 * (t[4 * iv + k] - t0) for k in range(4))
        wa = ("P1" if iv % 2 == 0 else "P2") + f"(t{iv // 2})" if iv < 2 * ntiles else "-"
        wb = ("P1" if (iv - 1) % 2 == 0 else "P2") + f"(t{(iv - 1) // 2})" if 1 <= iv else "-"
        print(f"{iv:3d} | {a0:7d} {a1:7d} {a1 - a0:6d} {wa:8s} | {b0:7d} {b1:7d} {b1 - b0:6d} {wb}")
if len(sys.argv) > 1 and sys.argv[1] == "ab":
    # 4-wave (rounds 1-2) vs 8-wave staggered workgroups, interleaved rounds in one process; results must be bit-identical
    import statistics
    from conformer_amd import _lib
    lib = _lib.load()
    variants = [(4, 0), (8, 0), (9, 0)]                      # (waves, -); 9 = the software-pipelined 8-wave form
    outs, times = {}, {vv: [] for vv in variants}
    Lr = torch.randint(1, T + 1, (B,), device=dev, generator=g)
    Lr[0] = T
    for nw in (4, 8, 9):
        lib.cfm_debug_set_attention_waves(nw)
        outs[nw] = ops.relpos_attention(qkv, pos, u, v, L, H).clone()
        outs[nw, "ragged"] = ops.relpos_attention(qkv, pos, u, v, Lr, H).clone()
    print("bit-identical:", bool(torch.equal(outs[4], outs[8])), " rel-L2 difference:", float((outs[4] - outs[8]).norm() / outs[4].norm()))
    for key in (9, (9, "ragged")):
        ref = outs[4] if key == 9 else outs[4, "ragged"]
        print(f"pipelined {key}: rel-L2 vs 4 waves {float((outs[key] - ref).norm() / ref.norm()):.3e}  max abs {float((outs[key] - ref).abs().max()):.3e}")
    for rnd in range(9):
        for vv in variants:
            lib.cfm_debug_set_attention_waves(vv[0])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.relpos_attention(qkv, pos, u, v, L, H)
            e1.record()
            torch.cuda.synchronize()
            times[vv].append(e0.elapsed_time(e1) / 20 * 1e3)
    lib.cfm_debug_set_attention_waves(0)
    fl = 6.0 * B * T * T * d
    for vv in variants:
        med = statistics.median(times[vv])
        print(f"{vv[0]} ({'pipelined 8' if vv[0] == 9 else 'waves'}): median {med:.1f} us  min {min(times[vv]):.1f} us  {fl / med / 1e6:.1f} TFLOP/s = {fl / med / 1e6 / 157.3:.3f} of the fp32 MFMA peak")
print("done")
