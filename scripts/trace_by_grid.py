#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel_trace.csv per (kernel, grid, workgroup): calls, mean / min duration (us)."""
import csv, glob, sys, collections
files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
acc = collections.defaultdict(list)
for f in files:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        short = name.split("(")[0][-70:]
        key = (short, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""), r.get("Workgroup_Size_X", ""))
        acc[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
pat = sys.argv[2] if len(sys.argv) > 2 else ""
rows = sorted(acc.items(), key=lambda kv: -sum(kv[1]))
for k, v in rows:
    if pat in k[0]:
        v2 = sorted(v)[: max(1, len(v) * 3 // 4)]
        print(f"{sum(v)/1e3:8.3f} ms  n={len(v):5d} mean={sum(v)/len(v):8.2f} us  q75mean={sum(v2)/len(v2):8.2f}  min={min(v):7.2f}  grid=({k[1]},{k[2]}) wg={k[3]}  {k[0]}")
