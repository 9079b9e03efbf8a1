"""Per-tensor error table of the bf16-autocast gfx950 path against the reference's fp32 AND the reference's own bf16
autocast results (goldens: tests/golden/autocast_*.npz).  Writes gpurun_out/autocast_report.json and prints a summary.

    python tools/autocast_report.py
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import autocast_cases as AC  # noqa: E402

dev = torch.device("cuda:0")
rep = {}
for case in ("autocast_modules_d32_t48", "autocast_modules_d144_t49", "autocast_modules_d512_t249"):
    rep[case] = AC.module_rows(case, dev)
for case in ("autocast_model_tiny", "autocast_model_cfg1_S"):
    rep[case] = AC.model_rows(case, dev)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(rep, open(os.path.join(ROOT, "gpurun_out", "autocast_report.json"), "w"), indent=1)
for case, rows in rep.items():
    nz = [r for r in rows if not r["zero"]]
    over = [r for r in nz if r["ours"] > 1e-2]
    print(f"{case}: {len(nz)} tensors; ours max {max(r['ours'] for r in nz):.2e} median {sorted(r['ours'] for r in nz)[len(nz)//2]:.2e}; "
          f"reference autocast max {max(r['ref'] for r in nz):.2e} median {sorted(r['ref'] for r in nz)[len(nz)//2]:.2e}; "
          f"ours > 1e-2: {len(over)}; ours > ref: {sum(r['ours'] > r['ref'] for r in nz)}")
    for r in sorted(over, key=lambda r: -r["ours"])[:15]:
        print(f"    {r['tensor']}: ours {r['ours']:.2e} ref {r['ref']:.2e} cross {r['cross']:.2e}")
    for r in rows:
        if r["zero"] and r["ours_abs"] > 1e-3:
            print(f"    ZERO-GRAD {r['tensor']}: ours |max| {r['ours_abs']:.2e} (reference autocast {r['ref_abs']:.2e})")
