#!/usr/bin/env python3
"""Idle time between consecutive kernels of one forward (hipGraph replay) from a rocprofv3 --kernel-trace CSV.

    rocprofv3 --kernel-trace --output-format csv -d DIR -o bench -- python3 bench.py --no-cpu-baseline --no-secondary --no-roofline
    python tools/graph_gaps.py DIR

Prints, for the LAST `n` kernels (n = launches per forward, found as the period of the kernel-name sequence): wall time from the
first start to the last end, the sum of kernel durations, the sum / count / distribution of the gaps, and the largest gaps with
the kernels either side of them."""
import csv
import glob
import os
import sys

d = sys.argv[1]
path = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# period of the tail of the sequence
tail = names[-4000:] if len(names) > 4000 else names
per = next((p for p in range(20, len(tail) // 3) if tail[-p:] == tail[-2 * p:-p] == tail[-3 * p:-2 * p]), None)
if per is None:
    sys.exit("no periodic tail found")
print(f"{path}: {len(rows)} kernels, period {per} launches per forward")
for k in (1, 2):
    seg = rows[len(rows) - k * per:len(rows) - (k - 1) * per]
    st = [int(r["Start_Timestamp"]) for r in seg]
    en = [int(r["End_Timestamp"]) for r in seg]
    wall = en[-1] - st[0]
    busy = sum(e - s for s, e in zip(st, en))
    gaps = [st[i + 1] - en[i] for i in range(per - 1)]
    pos = [g for g in gaps if g > 0]
    print(f"forward -{k}: wall {wall / 1e6:.3f} ms, kernels {busy / 1e6:.3f} ms, gaps {sum(pos) / 1e6:.3f} ms over {len(pos)} boundaries "
          f"(median {sorted(gaps)[len(gaps) // 2] / 1e3:.2f} us, max {max(gaps) / 1e3:.2f} us), overlapped boundaries {sum(1 for g in gaps if g < 0)}")
    if k == 1:
        order = sorted(range(per - 1), key=lambda i: -gaps[i])[:8]
        for i in order:
            print(f"   gap {gaps[i] / 1e3:7.2f} us after {seg[i]['Kernel_Name'][:70]}")
        bykernel = {}
        for r, s, e in zip(seg, st, en):
            key = r["Kernel_Name"][:90]
            c = bykernel.setdefault(key, [0, 0])
            c[0] += 1; c[1] += e - s
        for key, (n, t) in sorted(bykernel.items(), key=lambda kv: -kv[1][1])[:16]:
            print(f"   {t / 1e6:7.3f} ms  {n:4d} x {t / n / 1e3:8.1f} us  {key}")
