import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scripts.bench_gemm import SHAPES
from tramba_amd import hip
dev = torch.device("cuda")
for (m, n, k, c) in SHAPES:
    x = torch.randn(m, k, device=dev).bfloat16(); w = (torch.randn(n, k, device=dev) * k ** -0.5).bfloat16()
    for _ in range(6):
        torch.mm(x, w.t())
    for _ in range(6):
        hip.linear_cl(x, w, None, None, 0)
    torch.cuda.synchronize()
