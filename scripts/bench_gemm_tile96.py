#!/usr/bin/env python3
"""The K >= 256 GEMMs of the model at batch 4 and 8, hipGraph-timed, the library's 64 x 64 LDS-DMA tiles (TRAMBA_TUNE_GEMM_TILE 0)
against the 96 x 64 measurement form (15), bias + residual epilogue.  `model` = what a fill-rate-bound K loop would cost: rounds
of the chip x KB filled per tile and K step -- the prediction the measurement refutes (profiles/r03h_gemm_tile96.txt)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tramba_amd import hip
dev = torch.device("cuda")
FORMS = [int(v) for v in sys.argv[1:]] or [0, 15]     # TRAMBA_TUNE_GEMM_TILE values to compare (first = reference)
SHAPES = [(2304, 512, 1024), (2304, 512, 2048), (2304, 1024, 512), (2304, 2048, 512), (2304, 192, 1024), (576, 1024, 2048),
          (576, 1024, 4096), (576, 4096, 1024), (576, 2048, 1024), (9216, 256, 512), (9216, 256, 1024), (9216, 1024, 256),
          (36864, 128, 512), (4608, 512, 1024), (4608, 512, 2048), (4608, 1024, 512), (4608, 2048, 512), (1152, 1024, 2048),
          (1152, 4096, 1024), (18432, 256, 1024), (18432, 512, 256)]


def timed(fn):
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
    return e0.elapsed_time(e1) / 40 * 1e3, y


for m, n, k in SHAPES:
    x = torch.randn(m, k, device=dev).bfloat16()
    w = (torch.randn(n, k, device=dev) * k ** -0.5).bfloat16()
    b = torch.randn(n, device=dev)
    r = torch.randn(m, n, device=dev).bfloat16()
    t64, t96 = -(-m // 64) * -(-n // 64), -(-m // 96) * -(-n // 64)
    line = f"M={m:6d} N={n:5d} K={k:5d} model 64:{-(-t64 // 256) * 16:4d} 96:{-(-t96 // 256) * 20:4d} |"
    for name, fn in (("plain", lambda: hip.linear_cl(x, w, b, r, 0)),):
        ref = None
        for form in FORMS:
            hip.tune_set(hip.TUNE_GEMM_TILE, form)
            us, y = timed(fn)
            ref = y if ref is None else ref
            line += f"  {name} f{form}: {us:5.1f}{'' if torch.equal(y, ref) else ' DIFF'}"
    hip.tune_set(hip.TUNE_GEMM_TILE, 0)
    print(line, flush=True)
