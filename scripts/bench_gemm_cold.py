#!/usr/bin/env python3
"""tramba_linear_cl with operands that are NOT cache-resident: the launches of a timed run walk over enough (x, w, y) sets
to exceed the 256 MB Infinity Cache, as a GEMM inside the model finds its weights -- per TRAMBA_TUNE_GEMM_TILE choice."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tramba_amd import hip  # noqa: E402

SHAPES = [(2304, 512, 2048), (2304, 512, 1024), (2304, 2048, 512), (2304, 1024, 512), (36864, 128, 512), (36864, 512, 128),
          (9216, 256, 1024), (9216, 1024, 256), (576, 1024, 4096), (576, 4096, 1024), (2304, 144, 1024)]
NAMES = {0: "auto", 1: "reg64", 6: "dma3", 7: "dma4"}


def main():
    dev = torch.device("cuda")
    for m, n, k in SHAPES:
        per = (m * k + n * k + m * n) * 2
        nset = max(4, int(600e6 / per) + 1)
        xs = [torch.randn(m, k, device=dev).bfloat16() for _ in range(nset)]
        ws = [(torch.randn(n, k, device=dev) * k ** -0.5).bfloat16() for _ in range(nset)]
        b = torch.randn(n, device=dev)
        row = []
        for t in NAMES:
            hip.tune_set(hip.TUNE_GEMM_TILE, t)
            for i in range(nset):
                hip.linear_cl(xs[i], ws[i], b, None, 2)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 2
            e0.record()
            for _ in range(reps):
                for i in range(nset):
                    hip.linear_cl(xs[i], ws[i], b, None, 2)
            e1.record()
            torch.cuda.synchronize()
            row.append(e0.elapsed_time(e1) / (reps * nset) * 1e3)
        hip.tune_set(hip.TUNE_GEMM_TILE, 0)
        print(f"M={m:6d} N={n:5d} K={k:5d} sets={nset:4d}  " + "  ".join(f"{NAMES[t]} {row[i]:6.1f}us" for i, t in enumerate(NAMES)), flush=True)


if __name__ == "__main__":
    main()
