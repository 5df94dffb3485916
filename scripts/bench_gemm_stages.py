#!/usr/bin/env python3
"""The tall, short-K GEMMs of the model (output-bound: K <= 512, thousands of tiles), hipGraph-timed, per LDS stage count of
linear_dma_kernel: TRAMBA_TUNE_GEMM_TILE 0 = the library's choice, 13 = 2 stages (5 workgroups per CU), 6 = 3 stages."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tramba_amd import hip
dev = torch.device("cuda")
SHAPES = [(36864, 512, 128), (36864, 256, 128), (36864, 128, 256), (36864, 128, 512), (9216, 1024, 256), (9216, 512, 256),
          (9216, 256, 512), (9216, 256, 1024), (73728, 512, 128), (73728, 128, 512), (18432, 1024, 256), (2304, 2048, 512)]
for m, n, k in SHAPES:
    x = torch.randn(m, k, device=dev).bfloat16()
    w = (torch.randn(n, k, device=dev) * k ** -0.5).bfloat16()
    b = torch.randn(n, device=dev)
    line, ref = f"M={m:6d} N={n:5d} K={k:5d}:", None
    for act in (0, 2):
        for form in (0, 13, 6):
            hip.tune_set(hip.TUNE_GEMM_TILE, form)
            fn = lambda: hip.linear_cl(x, w, b, None, act)
            for _ in range(3):
                y = fn()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(20):
                    keep = fn()
            g.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 40 * 1e3
            if form == 0:
                ref = y
            line += f"  act{act} f{form}: {us:5.1f}{'' if torch.equal(y, ref) else ' DIFF'}"
    hip.tune_set(hip.TUNE_GEMM_TILE, 0)
    print(line, flush=True)
