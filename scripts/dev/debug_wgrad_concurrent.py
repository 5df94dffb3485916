#!/usr/bin/env python3
"""debug aid: is tramba_wgrad_cl (wgrad_dma_kernel) bitwise stable while ANOTHER stream runs a given partner kernel?
usage: debug_wgrad_concurrent.py [TRAMBA_TUNE_WGRAD_FORM: 0 = the library's kernel, 1 = register-staged, 2 = r03's counted waits]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tramba_amd import hip
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
form = int(sys.argv[1]) if len(sys.argv) > 1 else 0
hip.tune_set(hip.TUNE_WGRAD_FORM, form)

# weight-gradient shapes of a batch-8 step (tokens, N, K)
WG = [(4608, 512, 2048), (4608, 1024, 512), (18432, 256, 1024), (73728, 128, 512), (1152, 1024, 4096), (18432, 256, 2304 // 2)]
ops = []
for m, n, k in WG:
    gy = torch.randn(m, n, generator=g).bfloat16().to(dev)
    x = torch.randn(m, k, generator=g).bfloat16().to(dev)
    ops.append((gy, x))


def partners():
    b, h, d, r, kk = 8, 96, 256, 8, 8
    order = hip.scan_order("helix", h, h, dev)
    xc = torch.randn(b, h * h, d, generator=g).bfloat16().to(dev)
    wx = (torch.randn(kk, r + 2, d, generator=g) * d ** -0.5).bfloat16().to(dev)
    xdbl = hip.linear_cl(xc, hip.pad_x_proj_weight(wx), out_dtype=torch.float32)
    sargs = (xc, xdbl, order, (torch.randn(kk, d, r, generator=g) * r ** -0.5).to(dev), (torch.randn(kk * d, generator=g) * 0.5 - 2.0).to(dev),
             (-0.5 - torch.rand(kk * d, generator=g)).to(dev), torch.ones(kk * d).to(dev), torch.bfloat16)
    a = torch.randn(73728, 128, generator=g).bfloat16().to(dev)
    w = (torch.randn(512, 128, generator=g) * 0.1).bfloat16().to(dev)
    a2 = torch.randn(4608, 2048, generator=g).bfloat16().to(dev)
    w2 = (torch.randn(512, 2048, generator=g) * 0.02).bfloat16().to(dev)
    ln_w = torch.ones(128, device=dev)
    return {
        "none": lambda: None,
        "scan_dma": lambda: hip.ss2d_scan_cl(*sargs),
        "linear_ws": lambda: hip.linear_cl(a, w, None, None, 2),
        "linear_pc": lambda: hip.linear_cl(a2, w2, None, None, 0),
        "layernorm": lambda: hip.layernorm_cl(a, ln_w, ln_w),
        "wgrad": lambda: hip.wgrad_cl(ops[0][0], ops[0][1]),
        "memset": lambda: torch.empty(64 << 20, dtype=torch.uint8, device=dev).fill_(0x7F),
    }


side = torch.cuda.Stream()
ref = [hip.wgrad_cl(gy, x)[0].clone() for gy, x in ops]
torch.cuda.synchronize()
for name, fn in partners().items():
    bad = 0
    nonf = 0
    for rep in range(30):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(6):
                fn()
        outs = [hip.wgrad_cl(gy, x)[0] for gy, x in ops]
        torch.cuda.synchronize()
        for o, r_ in zip(outs, ref):
            if not torch.equal(o, r_):
                bad += 1
                nonf += int(not torch.isfinite(o).all())
    print(f"wgrad form {form}, partner {name:10s}: {bad} of {30 * len(ops)} results differ from the reference ({nonf} non-finite)", flush=True)
