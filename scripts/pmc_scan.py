"""Runs the scan kernels a few times on the bench shapes (for rocprofv3 --pmc passes): the L0 boundary scan on the model's
largest call shape, and the Helix-SS2D pair at 96x96 (fused scan with 2-byte ys + merge/out_norm), plus the raster scan."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tramba_amd import hip
dev = torch.device("cuda")
dtype = torch.bfloat16
only = sys.argv[1] if len(sys.argv) > 1 else "all"
g = torch.Generator().manual_seed(0)
if only in ("all", "boundary"):
    nb, kd, k, l = 4, 1024, 4, 9216
    u = torch.randn(nb, kd, l, generator=g).to(dev, dtype)
    delta = (0.5 * torch.randn(nb, kd, l, generator=g)).to(dev, dtype)
    A = -torch.ones(kd, 1, device=dev); B = torch.randn(nb, k, 1, l, generator=g).to(dev, dtype); C = torch.randn(nb, k, 1, l, generator=g).to(dev, dtype)
    D = torch.ones(kd, device=dev); bias = torch.full((kd,), -3.0, device=dev)
    for _ in range(5):
        hip.selective_scan_fwd(u, delta, A, B, C, D, bias, True, True, want_ckpt=False)
fams = [f for f in ("helix", "raster") if only in ("all", f)]
for fam in fams:
    order = hip.scan_order(fam, 96, 96, dev)
    kk, d, r = order.k, 256, 8
    x = torch.randn(4, 9216, d, generator=g).to(dev, dtype)
    wx = (torch.randn(kk, r + 2, d, generator=g) * d ** -0.5).to(dev, dtype)
    xdbl = hip.linear_cl(x, hip.pad_x_proj_weight(wx), out_dtype=torch.float32)
    dt_w = (torch.randn(kk, d, r, generator=g) * r ** -0.5).to(dev); dt_b = (torch.randn(kk * d, generator=g) * 0.5 - 3).to(dev)
    a = -torch.ones(kk * d, device=dev); ds = torch.ones(kk * d, device=dev)
    lw, lb = torch.ones(d, device=dev), torch.zeros(d, device=dev)
    for _ in range(5):
        ys = hip.ss2d_scan_cl(x, xdbl, order, dt_w, dt_b, a, ds, dtype)
        hip.ss2d_merge_norm_cl(ys, order, lw, lb, 1e-5, hip.ACT_GELU, dtype)
torch.cuda.synchronize()
