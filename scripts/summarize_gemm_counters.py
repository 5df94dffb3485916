"""scripts/pmc_gemm.sh passes -> profiles/<tag>_gemm_counters.json: per GEMM shape (grid size identifies it) the SQ counters of
launches 2..6, the MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) and the achieved
TFLOP/s from the kernel-trace duration.  usage: python scripts/summarize_gemm_counters.py <tag>"""
import csv, glob, json, os, sys
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
SHAPES = [(2304, 512, 2048), (2304, 2048, 512), (2304, 1024, 512), (2304, 512, 1024), (36864, 512, 128)]
vals = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))   # launch group -> counter -> dispatch -> value
dur = defaultdict(dict)
order = {}
for path in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", tag, "gemm_sq*", "**", "*counter_collection.csv"), recursive=True)):
    for row in csv.DictReader(open(path, newline="")):
        if "linear_" not in row["Kernel_Name"]:
            continue
        d = int(row["Dispatch_Id"])
        vals[path][row["Counter_Name"]][d] += float(row["Counter_Value"])
out = {"what": "rocprofv3 --pmc passes (scripts/pmc_gemm.sh) on the five heaviest GEMM shapes of a batch-4 forward, bf16, bias + GELU "
               "epilogue; averages over launches 2..6 of each shape; SQ_* cycle counters in quad-cycles summed over waves except "
               "SQ_VALU_MFMA_BUSY_CYCLES (cycles, summed over SIMDs) and SQ_BUSY_CYCLES (per SE); GRBM_GUI_ACTIVE summed over 8 XCDs",
       "shapes": {}}
merged = defaultdict(dict)
for path, counters in vals.items():
    for c, by_d in counters.items():
        ds = sorted(by_d)
        for si, shp in enumerate(SHAPES):   # six consecutive launches per shape, in SHAPES order
            grp = ds[6 * si:6 * si + 6][1:]
            if grp:
                merged[shp][c] = sum(by_d[d] for d in grp) / len(grp)
for path in glob.glob(os.path.join(ROOT, "gpurun_out", tag, "gemm_sq1", "**", "*kernel_trace.csv"), recursive=True):
    rows = sorted((r for r in csv.DictReader(open(path)) if "linear_" in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
    for si, shp in enumerate(SHAPES):
        grp = rows[6 * si:6 * si + 6][1:]
        if grp:
            merged[shp]["duration_us_under_pmc"] = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in grp) / len(grp) / 1e3
for shp, cs in merged.items():
    m, n, k = shp
    e = {c: round(v, 1) for c, v in sorted(cs.items())}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in cs and "GRBM_GUI_ACTIVE" in cs and cs["GRBM_GUI_ACTIVE"] > 0:
        e["mfma_utilisation"] = round(cs["SQ_VALU_MFMA_BUSY_CYCLES"] / (cs["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0), 4)
    if "SQ_WAIT_ANY" in cs and cs.get("SQ_WAVE_CYCLES"):
        e["wait_any_of_wave_cycles"] = round(cs["SQ_WAIT_ANY"] / cs["SQ_WAVE_CYCLES"], 4)
    if "SQ_LDS_BANK_CONFLICT" in cs and cs.get("SQ_LDS_IDX_ACTIVE"):
        e["lds_bank_conflict_of_lds_cycles"] = round(cs["SQ_LDS_BANK_CONFLICT"] / cs["SQ_LDS_IDX_ACTIVE"], 4)
    if "duration_us_under_pmc" in cs:
        e["tflops_under_pmc"] = round(2.0 * m * n * k / cs["duration_us_under_pmc"] / 1e6, 1)
    out["shapes"][f"M={m} N={n} K={k}"] = e
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_gemm_counters.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
