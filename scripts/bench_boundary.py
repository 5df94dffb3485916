import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tramba_amd import hip
dev = torch.device("cuda")
def run(nb, kd, k, l, dtype, softplus=True):
    g = torch.Generator().manual_seed(0)
    u = torch.randn(nb, kd, l, generator=g).to(dev, dtype); delta = (0.5 * torch.randn(nb, kd, l, generator=g)).to(dev, dtype)
    A = -torch.ones(kd, 1, device=dev); B = torch.randn(nb, k, 1, l, generator=g).to(dev, dtype); C = torch.randn(nb, k, 1, l, generator=g).to(dev, dtype)
    D = torch.ones(kd, device=dev); bias = torch.full((kd,), -3.0, device=dev)
    for _ in range(3): hip.selective_scan_fwd(u, delta, A, B, C, D, bias, softplus, True, want_ckpt=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): hip.selective_scan_fwd(u, delta, A, B, C, D, bias, softplus, True, want_ckpt=False)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    es = 2 if dtype != torch.float32 else 4
    byts = nb * kd * l * (2 * es + 4)
    print(f"B={nb:3d} KD={kd} L={l} {str(dtype)[6:]:9s} softplus={softplus}: {us:8.1f} us  {byts / us / 1e3:8.1f} GB/s  {nb*kd*l/us/1e3:7.1f} Gelem/s", flush=True)
run(4, 1024, 4, 9216, torch.bfloat16)
run(4, 1024, 4, 9216, torch.float32)
run(4, 1024, 4, 9216, torch.bfloat16, softplus=False)
run(16, 1024, 4, 9216, torch.bfloat16)
run(8, 2048, 8, 9216, torch.bfloat16)
run(4, 4096, 4, 576, torch.bfloat16)
run(4, 8192, 4, 144, torch.bfloat16)
