#!/usr/bin/env python3
"""Where does the LDS-DMA Helix scan's time go?  Times ss2d_scan_cl (helix 96x96, K=8, D=256, B=4, bf16) with libraries built
with -DTRAMBA_ABLATE=n (results wrong by construction; measurement only):
  0 product   1 stores dropped by the range check   2 no barrier   3 no barrier, no fold   4 no LDS-DMA (stale slot)
  5 no transcendentals   6 no replay (phase 2 skipped)   7 barrier kept, fold skipped
usage: exp_scan_ablate.py            (parent: one child process per variant)
       exp_scan_ablate.py <lib path> (child)"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def child(path, shapes=("helix0", "enc0", "enc2")):
    sys.path.insert(0, ROOT)
    import torch
    from tramba_amd import hip
    if path != "product":
        hip.LIB_PATH = path
    from scripts.bench_scan import SHAPES
    for name in shapes:
        fam, h, d, r = SHAPES[name]
        b, dtype, dev = 4, torch.bfloat16, torch.device("cuda")
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
        for _ in range(5):
            hip.ss2d_scan_cl(x, xdbl, order, dt_w, dt_b, a, ds, dtype)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            hip.ss2d_scan_cl(x, xdbl, order, dt_w, dt_b, a, ds, dtype)
        e1.record()
        torch.cuda.synchronize()
        print(f"  {name}: {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        for n in range(0, 8):
            path = "product" if n == 0 else os.path.join(HERE, "micro", "abl", f"lib_{n}.so")
            if n and not os.path.exists(path):
                continue
            print(f"ablate {n}:", flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), path], check=False)
