#!/usr/bin/env python3
"""tramba_linear_cl per block-tile choice (TRAMBA_TUNE_GEMM_TILE) on the Tramba-V GEMM shapes at batch 4 and 8:
one HIP-event pair around 20 back-to-back launches, bf16, bias + GELU epilogue."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tramba_amd import hip  # noqa: E402

BASE = [  # (tokens at batch 1, N, K, calls per forward)
    (576, 512, 2048, 17), (576, 512, 1024, 19), (576, 2048, 512, 17), (576, 1024, 512, 19),
    (9216, 128, 512, 5), (9216, 512, 128, 5), (9216, 256, 128, 6), (9216, 128, 256, 8),
    (2304, 256, 1024, 5), (2304, 1024, 256, 5), (2304, 512, 256, 6), (2304, 256, 512, 7),
    (144, 1024, 4096, 2), (144, 4096, 1024, 2), (144, 2048, 1024, 2), (144, 1024, 2048, 2),
]


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    dev = torch.device("cuda")
    names = {0: "auto", 1: "reg64", 6: "dma3", 7: "dma4"}
    for batch in (4, 8):
        tot = {t: 0.0 for t in names}
        best_tot = 0.0
        for t1, n, k, calls in BASE:
            m = t1 * batch
            x = torch.randn(m, k, device=dev).bfloat16()
            w = (torch.randn(n, k, device=dev) * k ** -0.5).bfloat16()
            b = torch.randn(n, device=dev)
            ref = None
            row = []
            for t in names:
                hip.tune_set(hip.TUNE_GEMM_TILE, t)
                y = hip.linear_cl(x, w, b, None, 2)
                if ref is None:
                    ref = y
                err = float((y.float() - ref.float()).abs().max())
                us = timeit(lambda: hip.linear_cl(x, w, b, None, 2))
                tot[t] += us * calls
                row.append((us, err))
            best_tot += min(r[0] for r in row) * calls
            print(f"B={batch} M={m:6d} N={n:5d} K={k:5d}  " + "  ".join(f"{names[t]} {row[i][0]:6.1f}us" for i, t in enumerate(names)) +
                  f"   max|diff| {max(r[1] for r in row):.1e}", flush=True)
        hip.tune_set(hip.TUNE_GEMM_TILE, 0)
        print(f"B={batch} per pass (us, calls-weighted): " + "  ".join(f"{names[t]} {tot[t]:8.1f}" for t in names) +
              f"  best-per-shape {best_tot:8.1f}", flush=True)


if __name__ == "__main__":
    main()
