#!/usr/bin/env python3
"""scripts/pmc_train.sh passes -> profiles/<tag>_train_traffic.json: HBM bytes per launch ((2 * FETCH_SIZE + WRITE_SIZE) KiB,
the gfx950 correction of MI355X_MICROARCH.md) of the training step's kernel families, from the LAST of the 4 steps of
scripts/profile_train.py (the steps are cut at the weight-shadow refresh that ends each):
  ss2d_scan_bwd_cl_kernel   per launch shape (grid): the Helix 96x96 launch and the 24x24 launches
  wgrad_dma_kernel + sums   all launches of the step
  linear_* (fwd + dgrad)    all launches of the step
usage: python scripts/summarize_train_traffic.py <tag>"""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"


def last_step(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    by_d = collections.OrderedDict()
    for r in rows:                                   # one row per (dispatch, XCD/instance): sum them
        d = int(r["Dispatch_Id"])
        e = by_d.setdefault(d, [r["Kernel_Name"], int(r["Grid_Size"]), int(r["Workgroup_Size"]), 0.0])
        e[3] += float(r["Counter_Value"])
    ds = sorted(by_d)
    cuts = [d for d in ds if "shadow_cast_multi" in by_d[d][0]]
    lo = cuts[-2] if len(cuts) >= 2 else ds[0]
    hi = cuts[-1] if cuts else ds[-1]
    return [by_d[d] for d in ds if lo < d <= hi]


out = {"what": "HBM bytes of one training step (batch 8, bf16, Tramba-V 384x384), last of 4 steps under rocprofv3 --pmc; "
               "traffic = (2 * FETCH_SIZE + WRITE_SIZE) KiB", "kernels": {}}
acc = collections.defaultdict(lambda: {"launches": 0, "FETCH_KiB": 0.0, "WRITE_KiB": 0.0})
for counter, d in (("FETCH_SIZE", "pmc_train_fetch"), ("WRITE_SIZE", "pmc_train_write")):
    fs = glob.glob(os.path.join(ROOT, "gpurun_out", tag, d, "*", "*counter_collection.csv"))
    if not fs:
        continue
    for name, grid, wg, val in last_step(fs[0], counter):
        if "ss2d_scan_bwd_cl_kernel" in name:
            key = f"ss2d_scan_bwd_cl_kernel@grid{grid}x{wg}"
        elif "wgrad_dma_kernel" in name or "wgrad_tn_kernel" in name:
            key = "wgrad (TN GEMMs)"
        elif "multi_sum_kernel" in name or "slab_sum_kernel" in name or "col_sum_kernel" in name:
            key = "partial sums (multi_sum / slab_sum / col_sum)"
        elif "linear_" in name:
            key = "linear (forward + input-gradient GEMMs)"
        elif "ss2d_scan_" in name or "ss2d_seg" in name:
            key = "ss2d scan forward"
        elif "merge_norm" in name:
            key = "ss2d merge (forward + gradient)"
        elif "layernorm" in name or "add_ln" in name:
            key = "layernorm forward / backward"
        elif "dwconv" in name:
            key = "depth-wise forward / backward / weight gradient"
        elif "adam_kernel" in name or "adam_bump" in name:
            key = "adam (optimizer step)"
        elif "sod_loss" in name:
            key = "sod_loss (deep-supervision loss forward + backward)"
        else:
            continue
        a = acc[key]
        a["FETCH_KiB" if counter == "FETCH_SIZE" else "WRITE_KiB"] += val
        if counter == "FETCH_SIZE":
            a["launches"] += 1
for k, a in acc.items():
    a["traffic_bytes"] = (2 * a["FETCH_KiB"] + a["WRITE_KiB"]) * 1024
    a["traffic_bytes_per_launch"] = a["traffic_bytes"] / max(a["launches"], 1)
    out["kernels"][k] = {kk: (round(v, 1) if isinstance(v, float) else v) for kk, v in a.items()}
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_train_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
