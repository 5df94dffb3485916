#!/usr/bin/env python3
"""tramba_wgrad_cl per TRAMBA_TUNE_WGRAD_FORM on the shapes of a batch-8 step, hipGraph-timed, interleaved: 0 = the library's kernel
(all transposed reads waited for before the MFMAs), 2 = r03's counted waits, 1 = register-staged"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tramba_amd import hip
dev = torch.device("cuda")
SHAPES = [(73728, 256, 128), (73728, 128, 256), (73728, 512, 128), (73728, 128, 512), (18432, 512, 256), (18432, 256, 512),
          (18432, 1024, 256), (18432, 256, 1024), (4608, 1024, 512), (4608, 512, 1024), (4608, 2048, 512), (4608, 512, 2048),
          (1152, 2048, 1024), (1152, 1024, 2048), (1152, 4096, 1024), (1152, 1024, 4096)]
FORMS = [0, 2, 1]
tot = {f: 0.0 for f in FORMS}
for m, n, k in SHAPES:
    gy = torch.randn(m, n, device=dev).bfloat16()
    x = torch.randn(m, k, device=dev).bfloat16()
    graphs = {}
    for f in FORMS:
        hip.tune_set(hip.TUNE_WGRAD_FORM, f)
        for _ in range(3):
            hip.wgrad_cl(gy, x, True)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(10):
                keep = hip.wgrad_cl(gy, x, True)
        graphs[f] = (g, keep)
    hip.tune_set(hip.TUNE_WGRAD_FORM, 0)
    best = {f: 1e9 for f in FORMS}
    for _ in range(3):
        for f in FORMS:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            graphs[f][0].replay(); torch.cuda.synchronize()
            e0.record(); graphs[f][0].replay(); graphs[f][0].replay(); e1.record(); torch.cuda.synchronize()
            best[f] = min(best[f], e0.elapsed_time(e1) / 20 * 1e3)
    for f in FORMS:
        tot[f] += best[f]
    same = torch.equal(graphs[0][1][0], graphs[2][1][0])
    print(f"M={m:6d} N={n:5d} K={k:5d}: " + "  ".join(f"f{f} {best[f]:6.1f} us" for f in FORMS) + ("" if same else "  f0 != f2"), flush=True)
print("sum:", {f: round(v, 1) for f, v in tot.items()})
