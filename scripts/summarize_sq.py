"""SQ-counter passes of scripts/pmc_sq.sh -> profiles/<tag>_sq_counters.json.

usage: python scripts/summarize_sq.py <tag>
Reads gpurun_out/<tag>/sq*/**/*counter_collection.csv, keeps the two kernels of the Helix-SS2D pair, averages every counter over
launches 2..5 (the first launch of a process pays code upload and cold caches) and states each next to SQ_WAVE_CYCLES.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = {"ss2d_scan_dma_kernel": "ss2d_scan_dma_kernel (Helix 96x96)",
           "ss2d_merge_norm_deep_kernel": "ss2d_merge_norm_deep_kernel (Helix 96x96)"}


def main(tag):
    vals = {v: defaultdict(lambda: defaultdict(float)) for v in KERNELS.values()}   # kernel -> counter -> dispatch -> value
    for path in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", tag, "sq*", "**", "*counter_collection.csv"), recursive=True)):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                for key, name in KERNELS.items():
                    if key in row["Kernel_Name"]:
                        vals[name][row["Counter_Name"]][int(row["Dispatch_Id"])] += float(row["Counter_Value"])
    out = {"what": "rocprofv3 --pmc passes (scripts/pmc_sq.sh) on the Helix-SS2D pair at 96x96, K=8, D=256, B=4, bf16; averages over "
                   "launches 2..5; SQ_* cycle counters are in quad-cycles summed over waves", "kernels": {}}
    for name, counters in vals.items():
        avg = {}
        for c, by_dispatch in counters.items():
            launches = [by_dispatch[d] for d in sorted(by_dispatch)][1:5]
            avg[c] = sum(launches) / max(len(launches), 1)
        wave = avg.get("SQ_WAVE_CYCLES", 0.0)
        out["kernels"][name] = {c: {"value": int(round(v)), "of_wave_cycles": round(v / wave, 4) if wave else None}
                                for c, v in sorted(avg.items())}
    dst = os.path.join(ROOT, "profiles", f"{tag}_sq_counters.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    for name, cs in out["kernels"].items():
        print(name)
        for c, v in cs.items():
            print(f"  {c:28s} {v['value']:>12d}  {v['of_wave_cycles']}")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r02")
