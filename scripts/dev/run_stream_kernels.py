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
for b in (4, 8):
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
for cc in (128, 256):
    x = torch.randn(73728, cc, generator=g).bfloat16().to(dev)
    dy = torch.randn(73728, cc, generator=g).bfloat16().to(dev)
    w = torch.randn(cc, generator=g).to(dev)
    for _ in range(6):
        hip.layernorm_bwd_cl(x, dy, w)
    for _ in range(6):
        hip.layernorm_cl(x, w, w)
torch.cuda.synchronize()
print("done")
