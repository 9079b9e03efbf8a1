#!/usr/bin/env python3
"""Where one step of the decoder-LSTM recurrence spends its time (GPU box only): per-launch device time by hipGraph-free
event timing for several (B, H), and the in-kernel s_memrealtime stamps of the first and the last workgroup."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conformer_amd import _lib, ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    lib = _lib.load()
    for B, H, T in [(64, 640, 249), (16, 640, 249), (64, 320, 249), (64, 640, 50)]:
        x = torch.randn(B, T, 512, device=dev)
        w_ih = torch.randn(4 * H, 512, device=dev) * 0.05
        w_hh = torch.randn(4 * H, H, device=dev) * 0.05
        bias = torch.zeros(4 * H, device=dev)
        gx = torch.randn(B, T, 4 * H, device=dev)
        y = torch.empty(B, T, H, device=dev); c = torch.empty(B, H, device=dev)
        st = torch.cuda.current_stream().cuda_stream

        def run():
            assert lib.cfm_lstm_fwd_f32(gx.data_ptr(), w_hh.data_ptr(), None, y.data_ptr(), c.data_ptr(), None, None, B, T, H, st) == 0
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            run()
        e1.record(); torch.cuda.synchronize()
        print(f"B={B} H={H} T={T}: {e0.elapsed_time(e1) / 5 / T * 1e3:.2f} us per step (forward recurrence only)")
    # the 16-bit recurrence (lstm_mfma16.hip): forward and backward per-step time
    for B, H, T in [(64, 640, 249), (32, 640, 249)]:
        gx = torch.randn(B, T, 4 * H, device=dev); w_hh = (torch.randn(4 * H, H, device=dev) * 0.05)
        w16 = w_hh.to(torch.bfloat16).view(4, H // 8, 8, H // 16, 2, 8).permute(1, 3, 4, 0, 2, 5).contiguous()
        wt16 = w_hh.t().contiguous().to(torch.bfloat16).view(H // 16, 16, 4 * H // 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous()
        y = torch.empty(B, T, H, device=dev); c = torch.empty(B, H, device=dev)
        h16 = torch.empty(2 * ((B + 31) // 32) * 32 * H, device=dev, dtype=torch.bfloat16); dg16 = torch.empty(2 * ((B + 15) // 16) * 16 * 4 * H, device=dev, dtype=torch.bfloat16)
        gates = torch.rand(B, T, 4 * H, device=dev); cells = torch.randn(B, T, H, device=dev); dy = torch.randn(B, T, H, device=dev)
        dG = torch.empty(B, T, 4 * H, device=dev); dc = torch.empty(B, H, device=dev)
        wt32 = w_hh.t().contiguous()
        wf = w_hh.view(4, H // 4, 4, H // 16, 4, 4).permute(1, 3, 4, 0, 2, 5).contiguous()
        wtf = w_hh.t().reshape(H // 16, 16, 4 * H // 16, 4, 4).permute(0, 2, 3, 1, 4).contiguous()
        hfr = torch.empty(2 * ((B + 15) // 16) * 16 * H, device=dev); dgf = torch.empty(2 * ((B + 15) // 16) * 16 * 4 * H, device=dev)
        y2 = torch.empty_like(y); dG2 = torch.empty_like(dG)
        st = torch.cuda.current_stream().cuda_stream
        fns = {"fwd 16-bit": lambda: lib.cfm_lstm_fwd_mfma16_f32(1, gx.data_ptr(), w16.data_ptr(), None, y.data_ptr(), c.data_ptr(), h16.data_ptr(), None, None, B, T, H, st),
               "bwd 16-bit": lambda: lib.cfm_lstm_bwd_mfma16_f32(1, dy.data_ptr(), gates.data_ptr(), cells.data_ptr(), wt16.data_ptr(), None, dG.data_ptr(), dc.data_ptr(), dg16.data_ptr(), B, T, H, st),
               "fwd fp32": lambda: lib.cfm_lstm_fwd_f32(gx.data_ptr(), w_hh.data_ptr(), None, y.data_ptr(), c.data_ptr(), None, None, B, T, H, st),
               "bwd fp32": lambda: lib.cfm_lstm_bwd_f32(dy.data_ptr(), gates.data_ptr(), cells.data_ptr(), wt32.data_ptr(), None, dG.data_ptr(), dc.data_ptr(), B, T, H, st),
               "fwd fp32 frag": lambda: lib.cfm_lstm_fwd_frag_f32(gx.data_ptr(), wf.data_ptr(), None, y2.data_ptr(), c.data_ptr(), hfr.data_ptr(), None, None, B, T, H, st),
               "bwd fp32 frag": lambda: lib.cfm_lstm_bwd_frag_f32(dy.data_ptr(), gates.data_ptr(), cells.data_ptr(), wtf.data_ptr(), None, dG2.data_ptr(), dc.data_ptr(), dgf.data_ptr(), B, T, H, st)}
        for name, fn in fns.items():
            for _ in range(2):
                assert fn() == 0
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record(); torch.cuda.synchronize()
            print(f"B={B} H={H}: {name}: {e0.elapsed_time(e1) / 5 / T * 1e3:.2f} us per step")
        print(f"B={B} H={H}: fragment-order fp32 kernels bit-identical to row-major: fwd {torch.equal(y, y2)}, bwd {torch.equal(dG, dG2)}")
    B, H, T = 64, 640, 249
    gx = torch.randn(B, T, 4 * H, device=dev); w_hh = torch.randn(4 * H, H, device=dev) * 0.05
    y = torch.empty(B, T, H, device=dev); c = torch.empty(B, H, device=dev)
    tr = torch.zeros(T, 2, 8, dtype=torch.int64, device=dev)
    lib.cfm_debug_lstm_trace(tr.data_ptr())
    assert lib.cfm_lstm_fwd_f32(gx.data_ptr(), w_hh.data_ptr(), None, y.data_ptr(), c.data_ptr(), None, None, B, T, H,
                                torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    lib.cfm_debug_lstm_trace(None)
    t = tr.cpu()
    for step in (100, 101, 102):
        for wg in (0, 1):
            r = t[step, wg]
            base = int(t[step, 0, 0])
            print(f"step {step} wg {'first' if wg == 0 else 'last '}: start +{(int(r[0]) - base) * 10} ns | contraction {(int(r[1]) - int(r[0])) * 10} | "
                  f"lds write {(int(r[2]) - int(r[1])) * 10} | barrier {(int(r[3]) - int(r[2])) * 10} | gates+stores issued {(int(r[4]) - int(r[3])) * 10} | "
                  f"drained {(int(r[5]) - int(r[4])) * 10} | total {(int(r[5]) - int(r[0])) * 10} ns")
        print(f"   step-to-step (first wg start): {(int(t[step + 1, 0, 0]) - int(t[step, 0, 0])) * 10} ns; "
              f"last wg end -> next first wg start: {(int(t[step + 1, 0, 0]) - int(t[step, 1, 5])) * 10} ns")


if __name__ == "__main__":
    main()
