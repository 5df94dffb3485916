#!/usr/bin/env python3
"""Launches tramba_linear_cl on the model's small-grid GEMM shapes a few times (for a rocprofv3 kernel trace:
kernel durations, not the launch-rate-bound wall time of scripts/bench_gemm.py)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tramba_amd import hip  # noqa: E402

SHAPES = [(2304, 512, 2048), (2304, 512, 1024), (2304, 1024, 512), (2304, 2048, 512), (576, 1024, 4096), (576, 4096, 1024),
          (576, 2048, 1024), (576, 1024, 2048), (2304, 192, 1024), (9216, 256, 1024), (9216, 256, 512)]
dev = torch.device("cuda")
for m, n, k in SHAPES:
    x = torch.randn(m, k, device=dev).bfloat16()
    w = (torch.randn(n, k, device=dev) * k ** -0.5).bfloat16()
    b = torch.randn(n, device=dev)
    r = torch.randn(m, n, device=dev).bfloat16()
    for _ in range(12):
        hip.linear_cl(x, w, b, r, 0)
    torch.cuda.synchronize()
