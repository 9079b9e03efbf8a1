#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the same command.

Corrections per /opt/skills/guides/MI355X_MICROARCH.md section HBM: counters are in KiB; on gfx950 FETCH_SIZE reports exactly
half of the bytes of a wide (16 B/lane) coalesced streaming read -> doubled; WRITE_SIZE is exact for 16 B/lane stores.
Writes profiles/<out>.json: {kernel: {launches, fetch_bytes_per_launch, write_bytes_per_launch, hbm_bytes_per_launch}}.
"""
import csv
import glob
import json
import sys
from collections import defaultdict


def per_kernel(d, counter):
    acc = defaultdict(list)
    for r in csv.DictReader(open(glob.glob(f"{d}/*/*counter_collection.csv")[0])):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


def main(fetch_dir, write_dir, out):
    f = per_kernel(fetch_dir, "FETCH_SIZE")
    w = per_kernel(write_dir, "WRITE_SIZE")
    res = {}
    for k in sorted(set(f) | set(w)):
        fv, wv = f.get(k, [0.0]), w.get(k, [0.0])
        fb = 2.0 * 1024.0 * sum(fv) / len(fv)          # gfx950: FETCH_SIZE counts 64 B per 128-B request -> x2
        wb = 1024.0 * sum(wv) / len(wv)
        res[k] = dict(launches=len(fv), fetch_bytes_per_launch=fb, write_bytes_per_launch=wb,
                      hbm_bytes_per_launch=fb + wb)
    json.dump(res, open(out, "w"), indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:14]:
        print(f"{k[:90]:90s} n={v['launches']:4d} fetch {v['fetch_bytes_per_launch'] / 1e6:9.1f} MB  write "
              f"{v['write_bytes_per_launch'] / 1e6:9.1f} MB")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3])
