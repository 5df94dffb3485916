"""Runs the two scan kernels a few times on the bench shapes (for rocprofv3 --pmc passes)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tramba_amd import hip
dev = torch.device("cuda")
dtype = torch.bfloat16
# boundary op, largest call shape of the model
nb, kd, k, l = 4, 1024, 4, 9216
g = torch.Generator().manual_seed(0)
u = torch.randn(nb, kd, l, generator=g).to(dev, dtype)
delta = (0.5 * torch.randn(nb, kd, l, generator=g)).to(dev, dtype)
A = -torch.ones(kd, 1, device=dev); B = torch.randn(nb, k, 1, l, generator=g).to(dev, dtype); C = torch.randn(nb, k, 1, l, generator=g).to(dev, dtype)
D = torch.ones(kd, device=dev); bias = torch.full((kd,), -3.0, device=dev)
for _ in range(5):
    hip.selective_scan_fwd(u, delta, A, B, C, D, bias, True, True, want_ckpt=False)
# fused op, Helix 96x96 (K=8) and encoder 96x96 (K=4)
for fam in ("helix", "raster"):
    order = hip.scan_order(fam, 96, 96, dev)
    kk, d, r = order.k, 256, 8
    x = torch.randn(4, 9216, d, generator=g).to(dev, dtype)
    wx = (torch.randn(kk, r + 2, d, generator=g) * d ** -0.5).to(dev, dtype)
    xdbl = hip.linear_cl(x, hip.pad_x_proj_weight(wx), out_dtype=torch.float32)
    dt_w = (torch.randn(kk, d, r, generator=g) * r ** -0.5).to(dev); dt_b = (torch.randn(kk * d, generator=g) * 0.5 - 3).to(dev)
    a = -torch.ones(kk * d, device=dev); ds = torch.ones(kk * d, device=dev)
    for _ in range(5):
        hip.ss2d_scan_cl(x, xdbl, order, dt_w, dt_b, a, ds, torch.float32)
torch.cuda.synchronize()
