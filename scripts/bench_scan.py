#!/usr/bin/env python3
"""Micro-benchmark of the fused channels-last scan (+ merge) on the model's call shapes (B=4, bf16)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tramba_amd import hip  # noqa: E402

SHAPES = {  # name: (family, H, D, R)
    "enc0": ("raster", 96, 256, 8), "helix0": ("helix", 96, 256, 8), "enc1": ("raster", 48, 512, 16),
    "enc2": ("raster", 24, 1024, 32), "enc3": ("raster", 12, 2048, 64), "win1": ("window", 48, 512, 16),
    "helix1": ("helix", 48, 512, 16), "helix2": ("helix", 24, 1024, 32), "dil2": ("dilation", 24, 1024, 32),
}


def run(name, reps=20, b=4, dtype=torch.bfloat16):
    fam, h, d, r = SHAPES[name]
    dev = torch.device("cuda")
    order = hip.scan_order(fam, h, h, dev)
    k, l = order.k, h * h
    g = torch.Generator().manual_seed(0)
    x = torch.randn(b, l, d, generator=g).to(dev, dtype)
    wx = (torch.randn(k, r + 2, d, generator=g) * d ** -0.5).to(dev, dtype)
    xdbl = hip.linear_cl(x, hip.pad_x_proj_weight(wx), out_dtype=torch.float32)
    dt_w = (torch.randn(k, d, r, generator=g) * r ** -0.5).to(dev)
    dt_b = (torch.randn(k * d, generator=g) * 0.5 - 3).to(dev)
    a = -torch.ones(k * d, device=dev)
    ds = torch.ones(k * d, device=dev)
    lw, lb = torch.ones(d, device=dev), torch.zeros(d, device=dev)
    for ys_dtype, seg, mform, sform in ((torch.float32, True, 0, 0), (dtype, True, 0, 0), (dtype, True, 1, 1), (dtype, True, 2, 2),
                                       (dtype, True, 0, 3)):
        hip.tune_set(hip.TUNE_MERGE_FORM, mform)
        hip.tune_set(hip.TUNE_SCAN_FORM, sform)
        for _ in range(3):
            ys = hip.ss2d_scan_cl(x, xdbl, order, dt_w, dt_b, a, ds, ys_dtype, segmented=seg)
            y = hip.ss2d_merge_norm_cl(ys, order, lw, lb, 1e-5, 2, dtype)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ev[0].record()
        for _ in range(reps):
            ys = hip.ss2d_scan_cl(x, xdbl, order, dt_w, dt_b, a, ds, ys_dtype, segmented=seg)
        ev[1].record()
        for _ in range(reps):
            y = hip.ss2d_merge_norm_cl(ys, order, lw, lb, 1e-5, 2, dtype)
        ev[2].record()
        torch.cuda.synchronize()
        ts, tm = ev[0].elapsed_time(ev[1]) / reps * 1e3, ev[1].elapsed_time(ev[2]) / reps * 1e3
        if mform == 0:
            yref = y.float()
        dmax = float((y.float() - yref).abs().max())
        elems = b * k * l * d
        print(f"{name:7s} {'segmented' if seg else 'chained  '} ys={str(ys_dtype)[6:]:8s} scan {ts:8.1f} us ({elems / ts / 1e3:7.2f} Gelem/s)  "
              f"merge[form {mform}] {tm:7.1f} us  |y - y(form 0)| {dmax:.2e}  scan form {sform}", flush=True)
    hip.tune_set(hip.TUNE_MERGE_FORM, 0)
    hip.tune_set(hip.TUNE_SCAN_FORM, 0)


if __name__ == "__main__":
    for n in (sys.argv[1:] or list(SHAPES)):
        run(n)
