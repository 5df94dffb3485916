#!/usr/bin/env python3
"""Per-block latency study of the chained fused scan: time vs batch (blocks per CU) and vs sequence length."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tramba_amd import hip

def run(fam, h, d, r, b, seg, reps=30, dtype=torch.bfloat16):
    dev = torch.device("cuda")
    order = hip.scan_order(fam, h, h, dev)
    k, l = order.k, h * h
    g = torch.Generator().manual_seed(0)
    x = torch.randn(b, l, d, generator=g).to(dev, dtype)
    rg = hip.ss2d_group_stride(r)
    xdbl = torch.randn(b, l, k * rg, generator=g).to(dev) * 0.1
    dt_w = (torch.randn(k, d, r, generator=g) * r ** -0.5).to(dev)
    dt_b = (torch.randn(k * d, generator=g) * 0.5 - 3).to(dev)
    a = -torch.ones(k * d, device=dev)
    ds = torch.ones(k * d, device=dev)
    for _ in range(3):
        ys = hip.ss2d_scan_cl(x, xdbl, order, dt_w, dt_b, a, ds, torch.float32, segmented=seg)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ys = hip.ss2d_scan_cl(x, xdbl, order, dt_w, dt_b, a, ds, torch.float32, segmented=seg)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / reps * 1e3
    blocks = ((d + 31) // 32) * k * b
    print(f"{fam} h={h:3d} D={d:5d} R={r:3d} B={b:2d} {'seg' if seg else 'chn'}: {t:7.1f} us  blocks={blocks:5d} tiles/seq={(l+31)//32:4d}"
          f"  {b*k*l*d/t/1e3:7.1f} Gelem/s", flush=True)

if __name__ == "__main__":
    run("raster", 24, 1024, 32, 4, False)
    run("helix", 24, 1024, 32, 4, False)
    run("raster", 12, 2048, 64, 4, False)
    run("raster", 12, 2048, 64, 4, True)
    run("raster", 48, 512, 16, 4, False)
    run("helix", 48, 512, 16, 4, False)
    run("raster", 96, 256, 8, 4, False)
    run("helix", 96, 256, 8, 4, False)
