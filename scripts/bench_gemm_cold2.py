#!/usr/bin/env python3
"""Which operand's residency matters?  tramba_linear_cl (default kernel) with x / w each either one tensor reused by every
launch (hot) or cycled over ~600 MB of copies (cold)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tramba_amd import hip  # noqa: E402

SHAPES = [(2304, 512, 2048), (2304, 512, 1024), (2304, 2048, 512), (2304, 1024, 512), (36864, 128, 512), (9216, 256, 1024),
          (576, 1024, 4096), (2304, 144, 1024)]


def main():
    dev = torch.device("cuda")
    for m, n, k in SHAPES:
        nx = max(4, int(600e6 / (m * k * 2)) + 1)
        nw = max(4, int(600e6 / (n * k * 2)) + 1)
        nset = min(max(nx, 8), 256)
        xs = [torch.randn(m, k, device=dev).bfloat16() for _ in range(min(nx, nset))]
        ws = [(torch.randn(n, k, device=dev) * k ** -0.5).bfloat16() for _ in range(min(nw, nset))]
        b = torch.randn(n, device=dev)
        row = []
        for xc, wc in ((False, False), (True, False), (False, True), (True, True)):
            def run(i):
                hip.linear_cl(xs[i % len(xs)] if xc else xs[0], ws[i % len(ws)] if wc else ws[0], b, None, 2)
            for i in range(nset):
                run(i)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(2 * nset):
                run(i)
            e1.record()
            torch.cuda.synchronize()
            row.append(e0.elapsed_time(e1) / (2 * nset) * 1e3)
        print(f"M={m:6d} N={n:5d} K={k:5d}  x hot, w hot {row[0]:6.1f}us | x cold {row[1]:6.1f} | w cold ({len(ws)} copies = "
              f"{len(ws) * n * k * 2 / 1e6:.0f} MB) {row[2]:6.1f} | both cold {row[3]:6.1f}", flush=True)


if __name__ == "__main__":
    main()
