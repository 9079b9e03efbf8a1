#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv (+ kernel_trace.csv) per kernel name."""
import csv
import glob
import sys
from collections import defaultdict


def main(d, pat=""):
    cc = glob.glob(f"{d}/*/*counter_collection.csv")[0]
    kt = glob.glob(f"{d}/*/*kernel_trace.csv")
    dur = {}
    if kt:
        for r in csv.DictReader(open(kt[0])):
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(cc)):
        name = r["Kernel_Name"]
        if pat and pat not in name:
            continue
        acc[name][r["Counter_Name"]].append((r["Dispatch_Id"], float(r["Counter_Value"])))
    for name, cs in acc.items():
        print(name[:110])
        ids = sorted({i for v in cs.values() for i, _ in v}, key=int)
        last = ids[-1]
        vals = {c: dict(v)[last] for c, v in cs.items() if last in dict(v)}
        if last in dur:
            print(f"   dispatch {last}: {dur[last]:.1f} us")
        for c, v in sorted(vals.items()):
            print(f"   {c:32s} {v:16.0f}")
        g = vals.get("GRBM_GUI_ACTIVE")
        if g and last in dur:
            print(f"   clock ~ {g / 8 / dur[last] / 1e3:.2f} GHz (GRBM_GUI_ACTIVE/8/dur)" if g > 1e6 else f"   clock ~ {g / dur[last] / 1e3:.2f} GHz (GRBM_GUI_ACTIVE/dur)")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in vals and g:
            for div, lab in ((1, "raw"), (8, "/8")):
                print(f"   MFMA busy / (GUI_ACTIVE{lab} * 1024 SIMDs) = {vals['SQ_VALU_MFMA_BUSY_CYCLES'] / (g / div * 1024):.3f}")
        if "SQ_WAVE_CYCLES" in vals:
            w = vals["SQ_WAVE_CYCLES"]
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                if c in vals:
                    print(f"   {c}/WAVE_CYCLES = {vals[c] / w:.3f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
