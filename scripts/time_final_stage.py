"""Times the last decoder stage of Tramba-V 384x384 batch 4 both ways: three kernels vs the single fused kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tramba_amd import hip as H

dev = "cuda"
b, h, cin, p = 4, 96, 128, 4
g = torch.Generator().manual_seed(0)
x = torch.randn(b, h, h, cin, generator=g).to(torch.bfloat16).to(dev)
w = (torch.randn(p * p * 128, cin, generator=g) * cin ** -0.5).to(torch.bfloat16).to(dev)
lw, lb = torch.ones(128, device=dev), torch.zeros(128, device=dev)
hw = (torch.randn(128, generator=g) * 128 ** -0.5).to(dev)


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return a.elapsed_time(e) / n * 1e3


print("unfused us", timeit(lambda: H.shuffle_norm_head_cl(H.linear_cl(x, w), lw, lb, hw, 0.1, p)))
print("gemm only us", timeit(lambda: H.linear_cl(x, w)))
print("fused us", timeit(lambda: H.expand_norm_head_cl(x, w, lw, lb, hw, 0.1, p)))
