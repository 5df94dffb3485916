#!/usr/bin/env python3
"""Debug helper: pairwise concurrency stress of the fp32 kernels on two streams (96x96 maps)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tramba_amd import hip as H

dev = torch.device("cuda")
dtype = torch.float32 if os.environ.get("DTYPE", "f32") == "f32" else torch.bfloat16
g = torch.Generator().manual_seed(5)

def mk_scan(d, r, fam="raster", h=96, b=1):
    order = H.scan_order(fam, h, h, dev)
    k, l = order.k, h * h
    x = torch.randn(b, l, d, generator=g).to(dev, dtype)
    wx = (torch.randn(k, r + 2, d, generator=g) * d ** -0.5).to(dev, dtype)
    xdbl = H.linear_cl(x, H.pad_x_proj_weight(wx), out_dtype=torch.float32)
    dt_w = (torch.randn(k, d, r, generator=g) * r ** -0.5).to(dev)
    dt_b = (torch.randn(k * d, generator=g) * 0.5 - 3).to(dev)
    a = -torch.rand(k * d, generator=g).to(dev) - 0.5
    ds = torch.ones(k * d, device=dev)
    lw, lb = torch.ones(d, device=dev), torch.zeros(d, device=dev)
    ys0 = H.ss2d_scan_cl(x, xdbl, order, dt_w, dt_b, a, ds, torch.float32)
    return {
        "scan": lambda: H.ss2d_scan_cl(x, xdbl, order, dt_w, dt_b, a, ds, torch.float32),
        "merge": lambda: H.ss2d_merge_norm_cl(ys0, order, lw, lb, 1e-5, 2, dtype),
        "xproj": lambda: H.linear_cl(x, H.pad_x_proj_weight(wx), out_dtype=torch.float32),
    }

def mk_misc(c, h=96, b=1):
    x = torch.randn(b, h, h, c, generator=g).to(dev, dtype)
    w = (torch.randn(2 * c, c, generator=g) * c ** -0.5).to(dev, dtype)
    bias = torch.randn(2 * c, generator=g).to(dev)
    lw, lb = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    w3 = torch.randn(c, 1, 3, 3, generator=g).to(dev)
    wt, bt = H.dw_pack(w3, None)
    return {
        "linear": lambda: H.linear_cl(x, w, bias, None, 2),
        "ln": lambda: H.layernorm_cl(x, lw, lb, 1e-5, 0),
        "dwconv": lambda: H.dwconv_cl(x, wt, bt, 1),
    }

main_ops = {**{"M." + k: v for k, v in mk_scan(512, 16).items()}, **{"M." + k: v for k, v in mk_misc(256).items()}}
side_ops = {**{"S." + k: v for k, v in mk_scan(256, 8, "window").items()}, **{"S." + k: v for k, v in mk_misc(128, 192).items()}}
side = torch.cuda.Stream()
torch.cuda.synchronize()
for mn, mf in ([] if os.environ.get('DETAIL') else main_ops.items()):
    ref = mf().clone(); torch.cuda.synchronize()
    for sn, sf in side_ops.items():
        bad = 0
        for rep in range(3):
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                keep = [sf() for _ in range(40)]
            outs = [mf() for _ in range(12)]
            torch.cuda.synchronize()
            bad += sum(int(not torch.equal(o, ref)) for o in outs)
        print(f"{mn:10s} with {sn:10s}: {bad} of 36 runs differ", flush=True)

print("---- detail: M.merge with S.linear")
mf, sf = main_ops["M.merge"], side_ops["S.linear"]
ref = mf().clone(); torch.cuda.synchronize()
for rep in range(3):
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        keep = [sf() for _ in range(40)]
    outs = [mf() for _ in range(12)]
    torch.cuda.synchronize()
    for o in outs:
        d = (o != ref)
        if d.any():
            idx = d.nonzero()
            pix = idx[:, 1].unique()
            print("differing elements", int(d.sum()), "pixels", len(pix), "first pixels", pix[:12].tolist(),
                  "channels of first pixel", idx[idx[:, 1] == pix[0]][:, 2][:8].tolist(), "n", int((idx[:, 1] == pix[0]).sum()),
                  "bad vals", o[0, pix[0]][d[0, pix[0]]][:8].tolist(), "ref vals", ref[0, pix[0]][d[0, pix[0]]][:8].tolist())
            break
