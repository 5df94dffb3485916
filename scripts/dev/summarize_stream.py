#!/usr/bin/env python3
"""counter passes of scripts/dev/pmc_stream.sh -> one line per (kernel, grid): every counter averaged over launches 2..6 of that grid.
usage: python scripts/dev/summarize_stream.py <tag> [out.json]"""
import csv, glob, json, os, sys
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tag = sys.argv[1]
vals = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))
dur = defaultdict(list)
for path in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", tag, "st*", "**", "*counter_collection.csv"), recursive=True)):
    for row in csv.DictReader(open(path, newline="")):
        name = row["Kernel_Name"]
        if "tramba::" not in name:
            continue
        short = name.split("tramba::")[1].split("(")[0][:60]
        key = f"{short} grid={row.get('Grid_Size', '?')}"
        vals[key][row["Counter_Name"]][int(row["Dispatch_Id"])] += float(row["Counter_Value"])
for path in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", tag, "st1", "**", "*kernel_trace.csv"), recursive=True)):
    for row in csv.DictReader(open(path, newline="")):
        name = row["Kernel_Name"]
        if "tramba::" in name:
            short = name.split("tramba::")[1].split("(")[0][:60]
            dur[f"{short} grid={row.get('Grid_Size', '?')}"].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
out = {}
for key, cs in vals.items():
    avg = {}
    for c, by in cs.items():
        ls = [by[d] for d in sorted(by)][1:]
        avg[c] = sum(ls) / max(len(ls), 1)
    if key in dur and len(dur[key]) > 1:
        avg["us_under_counters"] = sum(dur[key][1:]) / (len(dur[key]) - 1)
    out[key] = avg
    w = avg.get("SQ_WAVE_CYCLES", 0.0)
    print(key)
    for c, v in sorted(avg.items()):
        print(f"    {c:32s} {v:16.1f}" + (f"   {v / w:8.4f} of wave cycles" if w and c.startswith("SQ_") else ""))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
