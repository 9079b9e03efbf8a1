#!/usr/bin/env python3
"""Per-kernel matrix-pipe utilisation from one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE, SQ_WAVE_CYCLES,
SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY) + --kernel-trace of the same run.

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
              --kernel-trace -d DIR -o p --output-format csv -- python3 bench.py --no-graph --no-cpu-baseline --no-secondary --steps 3
    python tools/pmc_mfma.py DIR profiles/<out>.json

mfma_util = MFMA busy cycles / (1024 SIMDs x shader cycles of the dispatch); shader cycles = GRBM_GUI_ACTIVE / 8 (rocprofv3
sums the 8 XCDs, MI355X_MICROARCH.md "DVFS give-back"), which also gives the effective clock under that kernel's load.
"""
import csv
import glob
import json
import sys
from collections import defaultdict


def main(d, out):
    cc = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)[0]
    kt = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)
    dur = {}
    if kt:
        for r in csv.DictReader(open(kt[0])):
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    per = defaultdict(lambda: defaultdict(dict))
    for r in csv.DictReader(open(cc)):
        per[r["Kernel_Name"]][r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    res = {}
    for name, disp in per.items():
        rows = [v for v in disp.values() if "GRBM_GUI_ACTIVE" in v and v["GRBM_GUI_ACTIVE"] > 0]
        if not rows:
            continue
        n = len(rows)
        mean = lambda c: sum(v.get(c, 0.0) for v in rows) / n                                  # noqa: E731
        cyc = mean("GRBM_GUI_ACTIVE") / 8.0
        us = [dur[i] for i in disp if i in dur]
        entry = {"launches": n, "shader_cycles": cyc, "mfma_busy_cycles": mean("SQ_VALU_MFMA_BUSY_CYCLES"),
                 "mfma_util": mean("SQ_VALU_MFMA_BUSY_CYCLES") / (1024.0 * cyc)}
        if us:
            entry["avg_us"] = sum(us) / len(us)
            entry["effective_clock_ghz"] = cyc / (entry["avg_us"] * 1e3)
        w = mean("SQ_WAVE_CYCLES")
        if w > 0:
            entry["wave_cycle_split"] = {"waiting (s_waitcnt / barrier)": mean("SQ_WAIT_ANY") / w,
                                         "issue stall": mean("SQ_WAIT_INST_ANY") / w, "issuing": mean("SQ_ACTIVE_INST_ANY") / w}
        known = {"GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"}
        extra = sorted({c for v in rows for c in v} - known)
        if extra:                                  # any further counters of the pass: mean per dispatch (and per wave-cycle)
            entry["other_counters"] = {c: mean(c) for c in extra}
            if w > 0:
                entry["other_counters_per_wave_cycle"] = {c: mean(c) / w for c in extra}
        res[name] = entry
    json.dump(res, open(out, "w"), indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1].get("avg_us", 0) * kv[1]["launches"])[:14]:
        print(f"{v.get('avg_us', 0):9.1f} us x{v['launches']:4d}  mfma_util {v['mfma_util']:.3f}  clk {v.get('effective_clock_ghz', 0):.2f} GHz  {k[:90]}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
