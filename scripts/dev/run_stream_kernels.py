#!/usr/bin/env python3
"""Driver for counter passes on the step's streaming kernels (scripts/dev/pmc_stream.sh): the folded 7x7 depth-wise stencil of
DWMSMlp at the decoder's 96x96 stage (C = 512; batch 4 = the inference forward, batch 8 = the training step: forward with the dual
store, input gradient on flipped taps, weight gradient) and the LayerNorm backward of the two 96x96 shapes.  Six launches each."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tramba_amd import hip
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
c = 512
wt = (torch.randn(49, c, generator=g) * 0.1).to(dev)
bt = torch.randn(c, generator=g).to(dev)
for b in (() if len(sys.argv) > 1 else (4, 8)):
    x = torch.randn(b, 96, 96, c, generator=g).bfloat16().to(dev)
    for _ in range(6):
        hip.dwconv_cl(x, wt, bt, 2)
    if b == 8:
        gy = torch.randn(b, 96, 96, c, generator=g).bfloat16().to(dev)
        for _ in range(6):
            hip.dwconv_dual_cl(x, wt, bt, 2, True, False)
        for _ in range(6):
            hip.dwconv_dual_cl(gy, wt, bt, 0, False, True)
        for _ in range(6):
            hip.dwconv_wgrad_cl(x, gy, 7)
for cc in (() if len(sys.argv) > 1 else (128, 256)):
    x = torch.randn(73728, cc, generator=g).bfloat16().to(dev)
    dy = torch.randn(73728, cc, generator=g).bfloat16().to(dev)
    w = torch.randn(cc, generator=g).to(dev)
    for _ in range(6):
        hip.layernorm_bwd_cl(x, dy, w)
    for _ in range(6):
        hip.layernorm_cl(x, w, w)
torch.cuda.synchronize()
print("done")
# (r04, second use) the fused scan backward at the decoder's top stage and the encoder's first stage, batch 8 (scripts/dev/pmc_stream.sh)
if len(sys.argv) > 1 and sys.argv[1] == "scanbwd":
    for kk, form in ((8, "helix"), (4, "raster")):
        b_, h, d, r = 8, 96, 256, 8
        order = hip.scan_order(form, h, h, dev)
        xc = torch.randn(b_, h * h, d, generator=g).bfloat16().to(dev)
        wx = (torch.randn(kk, r + 2, d, generator=g) * d ** -0.5).bfloat16().to(dev)
        xdbl = hip.linear_cl(xc, hip.pad_x_proj_weight(wx), out_dtype=torch.float32)
        dtw = (torch.randn(kk, d, r, generator=g) * r ** -0.5).to(dev)
        dtb = (torch.randn(kk * d, generator=g) * 0.5 - 2.0).to(dev)
        A = (-0.5 - torch.rand(kk * d, generator=g)).to(dev)
        Ds = torch.ones(kk * d).to(dev)
        gym = torch.randn(b_, h * h, d, generator=g).bfloat16().to(dev)
        for _ in range(6):
            hip.ss2d_scan_bwd_cl(xc, xdbl, order, dtw, dtb, A, Ds, gym, bc_partials=True)
    torch.cuda.synchronize()
    print("done scanbwd")
