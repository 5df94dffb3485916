"""Kernel-trace comparison: torch.mm (hipBLASLt) vs tramba_linear_cl on the model's inference GEMM shapes (run under rocprofv3)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tramba_amd import hip
SHAPES = [(2304, 512, 2048), (2304, 512, 1024), (2304, 2048, 512), (2304, 1024, 512), (36864, 128, 512), (36864, 512, 128),
          (36864, 256, 128), (36864, 128, 256), (9216, 256, 1024), (9216, 1024, 256), (9216, 512, 256), (9216, 256, 512),
          (576, 1024, 4096), (576, 4096, 1024), (576, 2048, 1024), (576, 1024, 2048)]
dev = torch.device("cuda")
for m, n, k in SHAPES:
    x = torch.randn(m, k, device=dev).bfloat16()
    w = (torch.randn(n, k, device=dev) * k ** -0.5).bfloat16()
    for _ in range(10):
        hip.linear_cl(x, w, None, None, 0)
    torch.cuda.synchronize()
    for _ in range(10):
        torch.mm(x, w.t())
    torch.cuda.synchronize()
