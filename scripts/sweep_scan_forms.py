#!/usr/bin/env python3
"""Fused scan per model shape at batch 4 and batch 1, hipGraph-timed (30 launches per replay): the library's own choice (f0),
the LDS-DMA form (f3), the wave-segment form (f2) and the register ring at 1 / 2 / 4 / 8 waves per sequence (f1w*) -- the
measurement behind the form / W rules of tramba_ss2d_scan_cl (profiles/r03_scan_form_sweep.txt = the rules of r02 under the
mailbox carry).  The FIRST column of a row is measured right after the host-side set-up of the shape, on a chip whose clock
has dropped: it reads ~10 % high (f0 and f3 are the same launch on the 96x96 maps)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tramba_amd import hip
from scripts.bench_scan import SHAPES
dev = torch.device("cuda")
def t(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): keep = fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (2 * reps) * 1e3
for b in (4, 1):
  for name in ("enc0", "helix0", "enc1", "win1", "helix1", "enc2", "dil2", "helix2", "enc3"):
    fam, h, d, r = SHAPES[name]
    order = hip.scan_order(fam, h, h, dev)
    k, l = order.k, h * h
    g = torch.Generator().manual_seed(0)
    x = torch.randn(b, l, d, generator=g).to(dev, torch.bfloat16)
    wx = (torch.randn(k, r + 2, d, generator=g) * d ** -0.5).to(dev, torch.bfloat16)
    xdbl = hip.linear_cl(x, hip.pad_x_proj_weight(wx), out_dtype=torch.float32)
    dt_w = (torch.randn(k, d, r, generator=g) * r ** -0.5).to(dev)
    dt_b = (torch.randn(k * d, generator=g) * 0.5 - 3).to(dev)
    a = -torch.ones(k * d, device=dev); ds = torch.ones(k * d, device=dev)
    line = f"B={b} {name:7s}"
    for form, w in ((0, 0), (3, 0), (2, 0), (1, 1), (1, 2), (1, 4), (1, 8)):
        hip.tune_set(hip.TUNE_SCAN_FORM, form); hip.tune_set(hip.TUNE_SCAN_W, w)
        try:
            us = t(lambda: hip.ss2d_scan_cl(x, xdbl, order, dt_w, dt_b, a, ds, torch.bfloat16))
            line += f"  f{form}w{w}:{us:6.1f}"
        except Exception as e:
            line += f"  f{form}w{w}: err"
    hip.tune_set(hip.TUNE_SCAN_FORM, 0); hip.tune_set(hip.TUNE_SCAN_W, 0)
    print(line, flush=True)
