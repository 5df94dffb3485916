#!/usr/bin/env python3
"""Times tramba_linear_cl on the Tramba-V (B=4) GEMM shapes; TRAMBA_GEMM_TILE=<bm>x<bn> forces a tile."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tramba_amd import hip  # noqa: E402

SHAPES = [  # (M, N, K, calls per forward)
    (2304, 512, 2048, 17), (2304, 512, 1024, 19), (2304, 2048, 512, 17), (2304, 1024, 512, 19),
    (36864, 128, 512, 5), (36864, 512, 128, 5), (36864, 256, 128, 6), (36864, 128, 256, 8),
    (9216, 256, 1024, 5), (9216, 1024, 256, 5), (9216, 512, 256, 6), (9216, 256, 512, 7),
    (576, 1024, 4096, 2), (576, 4096, 1024, 2), (576, 2048, 1024, 2), (576, 1024, 2048, 2),
    (36864, 2048, 128, 1), (2304, 136, 1024, 17),
]


def main():
    dev = torch.device("cuda")
    tot = 0.0
    for m, n, k, calls in SHAPES:
        x = torch.randn(m, k, device=dev).bfloat16()
        w = (torch.randn(n, k, device=dev) * k ** -0.5).bfloat16()
        b = torch.randn(n, device=dev)
        for _ in range(3):
            hip.linear_cl(x, w, b, None, 2)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            hip.linear_cl(x, w, b, None, 2)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        tot += us * calls

        def timeit(fn):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                fn()
            b_.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b_) / 20 * 1e3
        plain = timeit(lambda: hip.linear_cl(x, w, None, None, 0))
        wt = w.t().contiguous()
        blas = timeit(lambda: torch.mm(x, w.t()))
        print(f"M={m:6d} N={n:5d} K={k:5d}  {us:8.2f} us  {2.0 * m * n * k / us / 1e6:7.1f} TF/s  x{calls}   "
              f"no-epilogue {plain:7.2f} us   hipBLASLt {blas:7.2f} us ({2.0 * m * n * k / blas / 1e6:6.1f} TF/s)", flush=True)
    print(f"weighted total {tot / 1e3:.3f} ms per forward (tile={os.environ.get('TRAMBA_GEMM_TILE', 'auto')})")


if __name__ == "__main__":
    main()
