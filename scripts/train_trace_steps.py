#!/usr/bin/env python3
"""Per-step summary of a rocprofv3 kernel trace of scripts/profile_train.py: the trace is cut at the weight-shadow refresh
(`shadow_cast_multi_kernel`, the last launch of every training step) and only the LAST step is summarised -- the first steps
of a process also create the optimizer state (1346 zero fills) and fill the per-module caches.
usage: python scripts/train_trace_steps.py <trace dir> [out.csv]"""
import collections, csv, glob, sys
files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
cuts = [i for i, r in enumerate(rows) if "shadow_cast_multi_kernel" in r["Kernel_Name"]]
if len(cuts) < 2:
    sys.exit("need at least two training steps in the trace")
step = rows[cuts[-2] + 1:cuts[-1] + 1]
lib = lambda n: "tramba::" in n or n.startswith("void col2im") or "im2col3x3" in n or "upsample_bilinear_bwd" in n  # noqa: E731
acc = collections.OrderedDict()
for r in step:
    n = r["Kernel_Name"]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    e = acc.setdefault(n, [0, 0.0])
    e[0] += 1
    e[1] += d
tot = sum(v[1] for v in acc.values())
tl = sum(v[1] for k, v in acc.items() if lib(k))
nl = sum(v[0] for k, v in acc.items() if lib(k))
na = sum(v[0] for v in acc.values())
wall = (int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"])) / 1e6
print(f"last step of {len(cuts)}: {na} launches, kernel time {tot / 1e3:.2f} ms (first start to last end {wall:.2f} ms); "
      f"library {nl} launches {tl / 1e3:.2f} ms; outside the library {na - nl} launches {(tot - tl) / 1e3:.2f} ms")
out = sorted(acc.items(), key=lambda kv: -kv[1][1])
if len(sys.argv) > 2:
    with open(sys.argv[2], "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Name", "Calls", "TotalUs", "AverageUs", "Percent", "Library"])
        for k, v in out:
            w.writerow([k, v[0], f"{v[1]:.1f}", f"{v[1] / v[0]:.2f}", f"{100 * v[1] / tot:.2f}", int(lib(k))])
for k, v in out[:70]:
    short = k.split("(")[0][-80:] if lib(k) else k[:130]
    print(f"{v[1] / 1e3:7.3f} ms n={v[0]:4d} avg={v[1] / v[0]:7.1f} us {'L' if lib(k) else ' '} {short}")
