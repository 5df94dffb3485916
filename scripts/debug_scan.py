#!/usr/bin/env python3
"""Debug helper: fused scan (both forms) vs the oracle on a few shapes; prints max abs error of ys-merged output."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tramba_amd import hip as H
from oracle import ops as oo
import torch.nn.functional as F

dev = torch.device("cuda")
for dtype in (torch.float32, torch.bfloat16):
    for fam in ("raster", "helix"):
        for cfg in [(1, 16, 40, 3), (1, 24, 32, 8), (1, 48, 128, 8), (1, 12, 96, 40), (1, 24, 32, 64), (2, 96, 64, 8)]:
            b, h, d, r = cfg
            k = 8 if fam == "helix" else 4
            g = torch.Generator().manual_seed(h * d + k)
            x = torch.randn(b, d, h, h, generator=g).to(dtype)
            wx = (torch.randn(k, r + 2, d, generator=g) * d ** -0.5).to(dtype)
            wdt = torch.randn(k, d, r, generator=g) * r ** -0.5
            dtb = torch.randn(k, d, generator=g) * 0.5 - 2.0
            a_logs = torch.log(0.5 + torch.rand(k * d, 1, generator=g))
            ds = 1 + 0.1 * torch.randn(k * d, generator=g)
            lw, lb = torch.ones(d), torch.zeros(d)
            y = oo.ss2d_core(x.double(), wx.double(), wdt.double(), dtb.double(), a_logs.double(), ds.double(), fam)
            want = y.permute(0, 2, 3, 1).reshape(b, h * h, d)
            order = H.scan_order(fam, h, h, dev)
            xc = x.permute(0, 2, 3, 1).contiguous().view(b, h * h, d).to(dev)
            xdbl = H.linear_cl(xc, H.pad_x_proj_weight(wx.to(dev)), out_dtype=torch.float32)
            for seg in (True, False):
                ys = H.ss2d_scan_cl(xc, xdbl, order, wdt.to(dev), dtb.reshape(-1).to(dev),
                                    (-torch.exp(a_logs)).reshape(-1).to(dev), ds.to(dev), torch.float32, segmented=seg)
                inv_ptr, inv_idx = order.inv_ptr.cpu(), order.inv_idx.cpu()
                ysf = ys.double().cpu().permute(0, 2, 1, 3)  # b, l, k, d
                # merge on the host via the inverse table
                got = torch.zeros(b, h * h, d, dtype=torch.float64)
                flat = ys.double().cpu().reshape(b, k * h * h, d)
                for p in range(h * h):
                    for e in range(int(inv_ptr[p]), int(inv_ptr[p + 1])):
                        got[:, p] += flat[:, int(inv_idx[e])]
                err = (got - want).abs()
                bad = (err > 1e-2 * (1 + want.abs())).nonzero()
                print(str(dtype)[6:], fam, cfg, "seg" if seg else "chain", "maxerr %.3e" % err.max().item(),
                      "nbad", len(bad), "first", bad[0].tolist() if len(bad) else None, flush=True)
