#!/usr/bin/env python3
"""Which kernels surround the ATen fill kernels of a traced run (rocprofv3 --kernel-trace CSV directory as argv[1]): tells one-time
state initialisation (long runs of consecutive fills: FusedAdam's moments on the first step) from per-step zero fills."""
import csv, glob, sys, collections
path = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
ctx = collections.Counter()
for i, n in enumerate(names):
    if "FillFunctor<float>" in n:
        nxt = next((names[j] for j in range(i + 1, min(i + 4, len(names))) if "FillFunctor" not in names[j]), "?")
        prv = next((names[j] for j in range(i - 1, max(i - 4, -1), -1) if "FillFunctor" not in names[j]), "?")
        dur = int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])
        ctx[(prv[:70], nxt[:70], "big" if dur > 10000 else "small")] += 1
for k, v in ctx.most_common(25):
    print(v, k)
