#!/usr/bin/env python3
"""debug aid: are the projection GEMMs (counted lgkmcnt waits on ds_read_b128 fragments) bitwise stable while ANOTHER stream runs a
partner kernel?  usage: debug_linear_concurrent.py [TRAMBA_TUNE_GEMM_TILE for the GEMM under test: 0 library, 18 r03 kernel, 16 pc, 19 ws]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tramba_amd import hip
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
form = int(sys.argv[1]) if len(sys.argv) > 1 else 0
SH = [(2304, 512, 2048), (2304, 2048, 512), (36864, 512, 128), (36864, 128, 256), (9216, 1024, 256), (576, 1024, 4096), (4608, 512, 1024)]
ops = []
for m, n, k in SH:
    x = torch.randn(m, k, generator=g).bfloat16().to(dev)
    w = (torch.randn(n, k, generator=g) * k ** -0.5).bfloat16().to(dev)
    b = torch.randn(n, generator=g).to(dev)
    cs = w.float().sum(dim=1).contiguous()
    ops.append((x, w, b, cs))


def under_test():
    hip.tune_set(hip.TUNE_GEMM_TILE, form)
    try:
        out = []
        for x, w, b, cs in ops:
            out.append(hip.linear_cl(x, w, b, None, 2))
            if x.shape[1] <= 2048:
                out.append(hip.linear_ln_cl(x, w, cs, b, 1e-5, None, 2))
        return out
    finally:
        hip.tune_set(hip.TUNE_GEMM_TILE, 0)


b_, h, d, r, kk = 8, 96, 256, 8, 8
order = hip.scan_order("helix", h, h, dev)
xc = torch.randn(b_, h * h, d, generator=g).bfloat16().to(dev)
wx = (torch.randn(kk, r + 2, d, generator=g) * d ** -0.5).bfloat16().to(dev)
xdbl = hip.linear_cl(xc, hip.pad_x_proj_weight(wx), out_dtype=torch.float32)
sargs = (xc, xdbl, order, (torch.randn(kk, d, r, generator=g) * r ** -0.5).to(dev), (torch.randn(kk * d, generator=g) * 0.5 - 2.0).to(dev),
         (-0.5 - torch.rand(kk * d, generator=g)).to(dev), torch.ones(kk * d).to(dev), torch.bfloat16)
gy = torch.randn(18432, 256, generator=g).bfloat16().to(dev)
xx = torch.randn(18432, 1024, generator=g).bfloat16().to(dev)
ys = hip.ss2d_scan_cl(*sargs)
lw = torch.ones(d, device=dev)
partners = {"none": lambda: None, "scan_dma": lambda: hip.ss2d_scan_cl(*sargs), "wgrad": lambda: hip.wgrad_cl(gy, xx),
            "merge": lambda: hip.ss2d_merge_norm_cl(ys, order, lw, lw, 1e-5, 2, torch.bfloat16),
            "dwconv7": lambda: hip.layernorm_cl(xc, lw, lw)}
side = torch.cuda.Stream()
ref = [t.clone() for t in under_test()]
torch.cuda.synchronize()
for name, fn in partners.items():
    bad = 0
    for rep in range(25):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(5):
                fn()
        outs = under_test()
        torch.cuda.synchronize()
        bad += sum(int(not torch.equal(o, r_)) for o, r_ in zip(outs, ref))
    print(f"GEMM form {form}, partner {name:9s}: {bad} of {25 * len(ref)} results differ from the reference", flush=True)
