import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tramba_amd import hip
dev = torch.device("cuda")
for (m, n, k) in ((36864, 512, 128), (36864, 128, 512)):
    x = torch.randn(m, k, device=dev).bfloat16(); w = (torch.randn(n, k, device=dev) * k ** -0.5).bfloat16()
    for _ in range(6):
        hip.linear_cl(x, w, None, None, 0)
torch.cuda.synchronize()
