#!/usr/bin/env python3
"""Median kernel duration per consecutive group of N launches of `pattern` kernels, for several rocprofv3 trace dirs.
usage: cmp_traces.py <pattern> <launches per shape> <skip> <dir> [<dir> ...]"""
import csv, glob, sys
pat, per, skip = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
for d in sys.argv[4:]:
    fs = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    if not fs:
        print(d, "missing"); continue
    rows = [r for r in csv.DictReader(open(fs[0])) if pat in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    out = []
    for i in range(len(rows) // per):
        t = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows[i * per + skip:(i + 1) * per])
        out.append(f"{t[len(t) // 2]:6.1f}")
    print(f"{d.split('/')[-1]:>10s}", " ".join(out))
