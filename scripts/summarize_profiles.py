#!/usr/bin/env python3
"""Distil gpurun_out/<tag>/ (written by scripts/collect_profiles.sh on the MI355X box) into profiles/.

  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (MIOpen find-mode kernels dropped)
  profiles/<tag>_train_kernel_stats.csv  per-kernel time of ONE steady-state training step (the last step of
                                    scripts/profile_train.py, cut out of the kernel trace by scripts/train_trace_steps.py)
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

import subprocess
if glob.glob(os.path.join(src, "trace_train", "**", "*kernel_trace.csv"), recursive=True):
    # ONE steady-state training step (the last of scripts/profile_train.py's; fwd + bwd + Adam, batch 8): the first steps of a
    # process also create the optimizer state and fill caches -- averaged over the whole process (as up to r02) they showed
    # ~650 fill launches per step that a steady-state step does not have.  scripts/train_trace_steps.py cuts the trace at the
    # weight-shadow refresh that ends every step.
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "train_trace_steps.py"), os.path.join(src, "trace_train"),
                        os.path.join(dst, f"{tag}_train_kernel_stats.csv")], capture_output=True, text=True)
    open(os.path.join(dst, f"{tag}_train_step_summary.txt"), "w").write(r.stdout.splitlines()[0] + "\n" if r.stdout else r.stderr)

lines = {}
for name in ("bench", "bench_train"):
    p = os.path.join(src, name + ".json")
    if os.path.exists(p):
        txt = [l for l in open(p).read().splitlines() if l.startswith("{")]
        if txt:
            lines[name] = json.loads(txt[-1])
json.dump(lines, open(os.path.join(dst, f"{tag}_bench.json"), "w"), indent=1)

traffic = {}
KERNELS = {"selective_scan_fwd": "selective_scan_fwd_kernel", "ss2d_scan_dma": "ss2d_scan_dma_kernel",
           "ss2d_scan_cl": "ss2d_scan_cl_kernel", "ss2d_seg": "ss2d_seg_kernel", "merge_norm_deep": "ss2d_merge_norm_deep_kernel",
           "merge_norm_stream": "ss2d_merge_norm_stream_kernel"}
for cname, d in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    fs = glob.glob(os.path.join(src, d, "*", "*counter_collection.csv"))
    if not fs:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"] != cname:
            continue
        kern = next((v for k, v in KERNELS.items() if k in r["Kernel_Name"]), None)
        if kern:
            agg[(kern, int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    for (kern, grid), v in agg.items():
        v = v[1:] if len(v) > 1 else v  # first launch is cold
        traffic.setdefault(f"{kern}@grid{grid}", {})[cname + "_KiB"] = sum(v) / len(v)
B, L, D = 4, 9216, 256
x_b, ys8, ys4 = B * L * D * 2, B * 8 * L * D * 2, B * 4 * L * D * 2
shapes = {  # key: (alias, what, algorithmic bytes at the kernel boundary)
    "selective_scan_fwd_kernel@grid262144": ("selective_scan_fwd_kernel@grid262144", "L0 scan (4,1024,9216) bf16->f32",
                                             4 * 1024 * 9216 * 8 + 2 * 4 * 4 * 9216 * 2),
    "ss2d_scan_dma_kernel@grid262144": ("ss2d_scan_dma_kernel@helix96", "fused scan Helix 96x96 K=8 D=256 B=4, ys bf16",
                                        x_b + B * L * 8 * 12 * 4 + ys8),
    "ss2d_scan_dma_kernel@grid131072": ("ss2d_scan_dma_kernel@raster96", "fused scan raster 96x96 K=4 D=256 B=4, ys bf16",
                                        x_b + B * L * 4 * 12 * 4 + ys4),
    "ss2d_merge_norm_deep_kernel@grid2359296": ("ss2d_merge_norm_deep_kernel@helix96",
                                                "merge + out_norm + GELU Helix 96x96 K=8 D=256 B=4, ys bf16", ys8 + x_b),
    "ss2d_merge_norm_stream_kernel@grid147456": ("ss2d_merge_norm_stream_kernel@raster96",
                                                 "merge + out_norm + GELU raster 96x96 K=4 D=256 B=4, ys bf16", ys4 + x_b),
}
out = {}
for k, v in traffic.items():
    if "FETCH_SIZE_KiB" in v and "WRITE_SIZE_KiB" in v:
        v["traffic_bytes"] = (2 * v["FETCH_SIZE_KiB"] + v["WRITE_SIZE_KiB"]) * 1024
    alias = k
    if k in shapes:
        alias, v["what"], v["algorithmic_bytes"] = shapes[k]
    out[alias] = v
traffic = out
json.dump(traffic, open(os.path.join(dst, f"{tag}_traffic.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1))

# per (kernel, grid) durations of the scan kernels in the single-stream trace: the Helix 96x96 launch that bench.py's
# `roofline` object times is the grid=(4096,8) wg=512 row
tb = os.path.join(root, "scripts", "trace_by_grid.py")
with open(os.path.join(dst, f"{tag}_scan_by_grid.txt"), "w") as f:
    for pat in ("ss2d_s", "ss2d_merge", "selective_scan"):
        f.write(subprocess.run([sys.executable, tb, os.path.join(src, "trace"), pat], capture_output=True, text=True).stdout)
