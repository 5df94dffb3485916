#!/usr/bin/env python3
"""Waves per sequence of the register-ring chained scan on the small maps (A/B through tramba_tune_set)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tramba_amd import hip
from bench_scan import SHAPES
dev = torch.device("cuda"); dtype = torch.bfloat16
for name in (sys.argv[1:] or ["enc2", "enc1", "helix1", "helix2", "enc3"]):
    fam, h, d, r = SHAPES[name]
    order = hip.scan_order(fam, h, h, dev); k, l = order.k, h * h
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4, l, d, generator=g).to(dev, dtype)
    wx = (torch.randn(k, r + 2, d, generator=g) * d ** -0.5).to(dev, dtype)
    xdbl = hip.linear_cl(x, hip.pad_x_proj_weight(wx), out_dtype=torch.float32)
    dt_w = (torch.randn(k, d, r, generator=g) * r ** -0.5).to(dev); dt_b = (torch.randn(k * d, generator=g) * 0.5 - 3).to(dev)
    a = -torch.ones(k * d, device=dev); ds = torch.ones(k * d, device=dev)
    for form, w in ((0, 0), (1, 1), (1, 2), (1, 4), (1, 8), (2, 0)):
        hip.tune_set(hip.TUNE_SCAN_FORM, form); hip.tune_set(hip.TUNE_SCAN_W, w)
        for _ in range(5): hip.ss2d_scan_cl(x, xdbl, order, dt_w, dt_b, a, ds, dtype)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): hip.ss2d_scan_cl(x, xdbl, order, dt_w, dt_b, a, ds, dtype)
        e1.record(); torch.cuda.synchronize()
        print(f"{name:7s} form {form} W {w}: {e0.elapsed_time(e1) / 30 * 1e3:7.1f} us", flush=True)
    hip.tune_set(hip.TUNE_SCAN_FORM, 0); hip.tune_set(hip.TUNE_SCAN_W, 0)
