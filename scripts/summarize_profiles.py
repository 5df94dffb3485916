#!/usr/bin/env python3
"""Distil gpurun_out/<tag>/ (written by scripts/collect_profiles.sh on the MI355X box) into profiles/.

  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (MIOpen find-mode kernels dropped)
  profiles/<tag>_train_kernel_stats.csv  the same for 3 training steps (scripts/profile_train.py)
  profiles/<tag>_bench.json         the bench.py line (and the --train line)
  profiles/<tag>_traffic.json       HBM bytes per launch of the two scan kernels from the PMC passes:
                                    traffic = (2*FETCH_SIZE + WRITE_SIZE) KiB -- on gfx950 FETCH_SIZE counts
                                    half of a wide coalesced read stream (MI355X_MICROARCH.md, HBM section)
"""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", tag), os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)

stats = glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv"))[0]
rows = [r for r in csv.DictReader(open(stats)) if "naive_conv" not in r["Name"]]
with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
    w.writeheader()
    w.writerows(rows)

tstats = glob.glob(os.path.join(src, "trace_train", "*", "*kernel_stats.csv"))
if tstats:   # 3 training steps (scripts/profile_train.py): fwd + bwd + Adam, batch 8
    trows = [r for r in csv.DictReader(open(tstats[0])) if "naive_conv" not in r["Name"]]
    with open(os.path.join(dst, f"{tag}_train_kernel_stats.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(trows[0].keys()))
        w.writeheader()
        w.writerows(trows)

lines = {}
for name in ("bench", "bench_train"):
    p = os.path.join(src, name + ".json")
    if os.path.exists(p):
        txt = [l for l in open(p).read().splitlines() if l.startswith("{")]
        if txt:
            lines[name] = json.loads(txt[-1])
json.dump(lines, open(os.path.join(dst, f"{tag}_bench.json"), "w"), indent=1)

traffic = {}
for cname, d in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    fs = glob.glob(os.path.join(src, d, "*", "*counter_collection.csv"))
    if not fs:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"] == cname and ("selective_scan_fwd" in r["Kernel_Name"] or "ss2d_scan_cl" in r["Kernel_Name"]):
            kern = "selective_scan_fwd_kernel" if "selective_scan_fwd" in r["Kernel_Name"] else "ss2d_scan_cl_kernel"
            agg[(kern, int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    for (kern, grid), v in agg.items():
        v = v[1:] if len(v) > 1 else v  # first launch is cold
        traffic.setdefault(f"{kern}@grid{grid}", {})[cname + "_KiB"] = sum(v) / len(v)
shapes = {"selective_scan_fwd_kernel@grid262144": ("L0 scan (4,1024,9216) bf16->f32", 4 * 1024 * 9216 * 8 + 2 * 4 * 4 * 9216 * 2),
          "ss2d_scan_cl_kernel@grid131072": ("fused scan Helix 96x96 K=8 D=256 B=4, ys f32", 4 * 9216 * 256 * 2 + 4 * 9216 * 8 * 12 * 4 + 4 * 8 * 9216 * 256 * 4),
          "ss2d_scan_cl_kernel@grid65536": ("fused scan raster 96x96 K=4 D=256 B=4, ys f32", 4 * 9216 * 256 * 2 + 4 * 9216 * 4 * 12 * 4 + 4 * 4 * 9216 * 256 * 4)}
for k, v in traffic.items():
    if "FETCH_SIZE_KiB" in v and "WRITE_SIZE_KiB" in v:
        v["traffic_bytes"] = (2 * v["FETCH_SIZE_KiB"] + v["WRITE_SIZE_KiB"]) * 1024
    if k in shapes:
        v["what"], v["algorithmic_bytes"] = shapes[k]
json.dump(traffic, open(os.path.join(dst, f"{tag}_traffic.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1))

# per (kernel, grid) durations of the scan kernels in the single-stream trace: the Helix 96x96 launch that bench.py's
# `roofline` object times is the grid=(4096,8) wg=512 row
import subprocess
tb = os.path.join(root, "scripts", "trace_by_grid.py")
with open(os.path.join(dst, f"{tag}_scan_by_grid.txt"), "w") as f:
    for pat in ("ss2d_s", "selective_scan"):
        f.write(subprocess.run([sys.executable, tb, os.path.join(src, "trace"), pat], capture_output=True, text=True).stdout)
