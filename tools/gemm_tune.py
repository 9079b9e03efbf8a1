#!/usr/bin/env python3
"""Times the fp32 MFMA GEMM at the hot-path shapes for every block-tile shape (GPU box only).

    python tools/gemm_tune.py            # table: shape x cfg -> us, TFLOP/s, and what the heuristic picks
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conformer_amd import _lib  # noqa: E402

SHAPES = [(7968, 2048, 512), (7968, 512, 2048), (7968, 1536, 512), (7968, 512, 512), (7968, 1024, 512),
          (7968, 512, 9728), (497, 512, 512), (15936, 2048, 512), (15936, 512, 2048), (98, 144, 144), (98, 576, 144)]
CFG = ["128x128", "128x64", "64x128", "64x64", "auto"]
ENG = ["reg", "dma", "dma-norefill"]


def one(M, N, K, cfg, iters):
    """Run one shape/cfg `iters` times (for rocprofv3 --pmc / --kernel-trace runs)."""
    lib = _lib.load()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / K ** 0.5
    b = torch.randn(N, device=dev); r = torch.randn(M, N, device=dev); c = torch.empty(M, N, device=dev)
    for _ in range(iters):
        assert lib.cfm_debug_gemm_cfg_f32(cfg, a.data_ptr(), w.data_ptr(), b.data_ptr(), r.data_ptr(), 0.5,
                                          c.data_ptr(), M, N, K, None, st) == 0
    torch.cuda.synchronize()


def trace(M, N, K, cfg):
    """Per-block timeline of one launch: when blocks start / finish their main loop, per CU."""
    import numpy as np
    lib = _lib.load()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / K ** 0.5
    b = torch.randn(N, device=dev); r = torch.randn(M, N, device=dev); c = torch.empty(M, N, device=dev)
    bm, bn = [(128, 128), (128, 64), (64, 128), (64, 64)][cfg & 15]
    nblk = ((M + bm - 1) // bm) * ((N + bn - 1) // bn)
    tr = torch.zeros(nblk, 8, dtype=torch.int64, device=dev)
    for it in range(3):
        assert lib.cfm_debug_gemm_cfg_f32(cfg, a.data_ptr(), w.data_ptr(), b.data_ptr(), r.data_ptr(), 0.5,
                                          c.data_ptr(), M, N, K, tr.data_ptr(), st) == 0
        torch.cuda.synchronize()
    t = tr.cpu().numpy()
    t0 = t[:, 0].min()
    start = (t[:, 0] - t0) / 100.0      # us
    end = (t[:, 1] - t0) / 100.0
    hw = t[:, 2]; xcc = t[:, 3] & 0xF
    cu = ((hw >> 8) & 0xF) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5)   # CU_ID | SH_ID | SE_ID
    key = xcc * 1024 + cu
    print(f"{M}x{N}x{K} cfg {cfg}: {nblk} blocks; main loops end at {end.max():.1f} us after the first block start")
    print(f"  block start: p0 {start.min():.1f} p50 {np.median(start):.1f} p90 {np.percentile(start, 90):.1f} max {start.max():.1f} us")
    epi_i = (t[:, 4] - t0) / 100.0 - end
    epi_d = (t[:, 5] - t0) / 100.0 - end
    print(f"  epilogue (wave 0): issued after p50 {np.median(epi_i):.1f} p90 {np.percentile(epi_i, 90):.1f} max {epi_i.max():.1f} us; "
          f"drained after p50 {np.median(epi_d):.1f} p90 {np.percentile(epi_d, 90):.1f} max {epi_d.max():.1f} us")
    dur = end - start
    print(f"  main-loop duration per block: min {dur.min():.1f} p50 {np.median(dur):.1f} p90 {np.percentile(dur, 90):.1f} max {dur.max():.1f} us")
    hist, edges = np.histogram(end, bins=12)
    print("  main-loop END histogram (us):", " ".join(f"{edges[i]:.0f}:{hist[i]}" for i in range(len(hist))))
    hist, edges = np.histogram(start, bins=12)
    print("  block START histogram (us):  ", " ".join(f"{edges[i]:.0f}:{hist[i]}" for i in range(len(hist))))
    ucu, cnt = np.unique(key, return_counts=True)
    print(f"  distinct (xcc,cu) ids seen: {len(ucu)}; blocks per CU: min {cnt.min()} max {cnt.max()}")
    # concurrency over time: how many blocks are in their main loop
    ts = np.linspace(0, end.max(), 16)
    print("  blocks in main loop at t(us):", " ".join(f"{x:.0f}:{int(((start <= x) & (end > x)).sum())}" for x in ts))


def occ_sweep():
    lib = _lib.load()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    for (M, N, K) in [(7968, 2048, 512), (7968, 512, 2048), (7968, 1536, 512), (7968, 512, 512), (7968, 1024, 512)]:
        a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / K ** 0.5
        b = torch.randn(N, device=dev); r = torch.randn(M, N, device=dev); c = torch.empty(M, N, device=dev)
        for tile in (0, 1, 3):
            bm, bn = [(128, 128), (128, 64), (64, 128), (64, 64)][tile]
            tiles = ((M + bm - 1) // bm) * ((N + bn - 1) // bn)
            line = f"{M}x{N}x{K} {CFG[tile]:>7s} ({tiles} tiles): "
            for cap in (0, 2, 3, 4, 5, 6, 8):
                cfg = tile + 256 * cap

                def run():
                    assert lib.cfm_debug_gemm_cfg_f32(cfg, a.data_ptr(), w.data_ptr(), b.data_ptr(), r.data_ptr(), 0.5,
                                                      c.data_ptr(), M, N, K, None, st) == 0
                for _ in range(3):
                    run()
                torch.cuda.synchronize()
                best = 1e9
                for _ in range(4):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(20):
                        run()
                    e1.record()
                    torch.cuda.synchronize()
                    best = min(best, e0.elapsed_time(e1) / 20)
                line += f"| cap{cap} {best * 1e3:6.1f}us {2.0 * M * N * K / best / 1e9:5.1f} "
            print(line, flush=True)


def bwd_sweep(prec=0):
    """Backward-GEMM products of the hot path at every block tile (prec 0 = fp32 MFMA, 1 = bf16, 2 = fp16)."""
    from conformer_amd import ops
    lib = _lib.load()
    dev = torch.device("cuda:0")
    M = int(os.environ.get("TUNE_M", "7968"))
    cases = []
    for (N, K) in [(2048, 512), (512, 2048), (1536, 512), (512, 512), (1024, 512)]:
        cases.append((f"dX  M x{K:5d} (Kc={N:5d})", False, True, M, K, N, False))      # dY (M,N) . W (N,K)
        cases.append((f"dW {N:5d}x{K:5d} (Kc=M)   ", True, True, N, K, M, True))         # dY^T . X
    cases.append(("dZ  M x 2048 (Kc=512) dswish", False, True, M, 2048, 512, False))
    for name, a_col, b_col, I, J, Kc, split in cases:
        A = torch.randn((Kc, I) if a_col else (I, Kc), device=dev)
        Bm = torch.randn((Kc, J) if b_col else (J, Kc), device=dev)
        Z = torch.randn(I, J, device=dev) if "dswish" in name else None
        line = f"{name}: "
        for tile in (0, 1, 3, -1):
            lib.cfm_debug_set_bwd_tile(tile)
            out = torch.zeros(I, J, device=dev)

            def run():
                ops.gemm_bwd(A, a_col, Bm, b_col, I, J, Kc, Z=Z, out=out, allow_split=split, prec=prec)
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    run()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 10)
            tname = {0: "128x128", 1: "128x64", 3: "64x64", -1: "auto"}[tile]
            line += f"| {tname:>7s} {best * 1e3:6.1f}us {2.0 * I * J * Kc / best / 1e9:5.1f} "
        lib.cfm_debug_set_bwd_tile(-1)
        print(line, flush=True)


def bk_sweep():
    """K-tile 16 vs 32 (whole 128-byte lines per staged row) at the hot-path shapes with their own epilogues, every block tile,
    interleaved rounds in one process (median of 7 rounds x 20 launches)."""
    import statistics
    lib = _lib.load()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    M = int(os.environ.get("TUNE_M", "7968"))
    sites = [("FFN hidden swish", M, 2048, 512, 32), ("FFN out resid", M, 512, 2048, 0), ("QKV bias", M, 1536, 512, 16),
             ("attn-out/pw2 resid", M, 512, 512, 0), ("pw1 (bias stand-in)", M, 1024, 512, 16), ("input linear bias", M, 512, 9728, 16)]
    for name, m, n, k, epi in sites:
        a = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev) / k ** 0.5
        b = torch.randn(n, device=dev); r = torch.randn(m, n, device=dev); c = torch.empty(m, n, device=dev)
        variants = [(tile, bk) for tile in (0, 1, 2, 3) for bk in (16, 32)]
        times = {v: [] for v in variants}

        def run(v):
            cfg = v[0] + epi + (64 if v[1] == 32 else 0)
            s_ = lib.cfm_debug_gemm_cfg_f32(cfg, a.data_ptr(), w.data_ptr(), b.data_ptr(), r.data_ptr(), 0.5, c.data_ptr(), m, n, k,
                                            None, st)
            assert s_ == 0, s_
        ref = None
        for v in variants:
            run(v); torch.cuda.synchronize()
            if ref is None:
                ref = c.clone()
            else:
                err = float((c - ref).norm() / ref.norm())
                assert err < 1e-5, (v, err)
        for rnd_ in range(7):
            for v in variants:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    run(v)
                e1.record()
                torch.cuda.synchronize()
                times[v].append(e0.elapsed_time(e1) / 20)
        line = f"{name:>22s} {m}x{n}x{k}: "
        for v in variants:
            med = statistics.median(times[v])
            line += f"| {CFG[v[0]]}/k{v[1]} {med * 1e3:6.1f}us {2.0 * m * n * k / med / 1e9:5.1f} "
        print(line, flush=True)


def main():
    if len(sys.argv) == 2 and sys.argv[1] == "bk":
        return bk_sweep()
    if len(sys.argv) == 2 and sys.argv[1] in ("bwd", "bwd16"):
        return bwd_sweep(1 if sys.argv[1] == "bwd16" else 0)
    if len(sys.argv) == 2 and sys.argv[1] == "occ":
        return occ_sweep()
    if len(sys.argv) == 6 and sys.argv[1] == "trace":
        return trace(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]))
    if len(sys.argv) >= 5:
        return one(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]),
                   int(sys.argv[5]) if len(sys.argv) > 5 else 10)
    lib = _lib.load()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    for (M, N, K) in SHAPES:
        a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / K ** 0.5
        b = torch.randn(N, device=dev); r = torch.randn(M, N, device=dev)
        ref = None
        line = f"{M:6d}x{N:5d}x{K:5d} "
        cfgs = [0, 1, 2, 3, -1]
        for ci, cfg in enumerate(cfgs):
            c = torch.empty(M, N, device=dev)

            def run():
                s = lib.cfm_debug_gemm_cfg_f32(cfg, a.data_ptr(), w.data_ptr(), b.data_ptr(), r.data_ptr(), 0.5,
                                               c.data_ptr(), M, N, K, None, st)
                assert s == 0, s
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            reps, inner = 5, 20
            best = 1e9
            for _ in range(reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(inner):
                    run()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / inner)
            if ref is None:
                ref = c.clone()
            else:
                assert torch.allclose(ref, c, rtol=1e-4, atol=1e-4), "configs must agree"
            name = "auto" if cfg < 0 else f"{ENG[cfg >> 2]}:{CFG[cfg & 3]}"
            line += f"| {name:>11s} {best * 1e3:7.1f}us {2.0 * M * N * K / best / 1e9:6.1f} "
        print(line, flush=True)


if __name__ == "__main__":
    main()
