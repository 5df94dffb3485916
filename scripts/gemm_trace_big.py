#!/usr/bin/env python3
"""Launches tramba_linear_cl on the model's tall GEMM shapes (96x96 / 48x48 stages) for a rocprofv3 kernel trace."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tramba_amd import hip  # noqa: E402

SHAPES = [(36864, 512, 128, 2, 0), (36864, 256, 128, 0, 0), (36864, 128, 256, 0, 1), (36864, 128, 512, 0, 1),
          (9216, 1024, 256, 2, 0), (9216, 512, 256, 0, 0), (9216, 256, 512, 0, 1), (9216, 256, 1024, 0, 1),
          (36864, 48, 256, 0, 0), (9216, 80, 512, 0, 0)]
dev = torch.device("cuda")
for m, n, k, act, use_res in SHAPES:
    x = torch.randn(m, k, device=dev).bfloat16()
    w = (torch.randn(n, k, device=dev) * k ** -0.5).bfloat16()
    b = torch.randn(n, device=dev)
    r = torch.randn(m, n, device=dev).bfloat16() if use_res else None
    ref = None
    for _ in range(12):
        y = hip.linear_cl(x, w, b, r, act)
    torch.cuda.synchronize()
    want = x.float() @ w.float().T + b
    if act == 2:
        want = torch.nn.functional.gelu(want)
    if r is not None:
        want = want + r.float()
    err = float((y.float() - want).abs().max())
    print(m, n, k, "max err", err, flush=True)
