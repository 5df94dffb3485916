#!/usr/bin/env python3
"""Average rocprofv3 counter_collection.csv values per (kernel, grid) -> table of counters."""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "tramba" not in r["Kernel_Name"]:
                continue
            key = (r["Kernel_Name"].split("(")[0].replace("void tramba::", "")[:80], r.get("Grid_Size", ""))
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, cs in acc.items():
    print(key)
    for c, v in sorted(cs.items()):
        print(f"    {c:32s} {sum(v)/len(v):16.1f}  (n={len(v)})")
