#!/usr/bin/env python3
"""How much of the Helix merge's time is the tail of its long pixels?  Times ss2d_merge_norm_cl (96x96, K=8, D=256, B=4,
bf16) on the real Helix inverse table and on a table with the SAME number of entries spread evenly (8 per pixel)."""
import copy
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tramba_amd import hip  # noqa: E402


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main(h=96, d=256, b=4, dtype=torch.bfloat16):
    dev = torch.device("cuda")
    order = hip.scan_order("helix", h, h, dev)
    k, l = order.k, h * h
    ys = torch.randn(b, k, l, d, device=dev).to(dtype)
    lw, lb = torch.ones(d, device=dev), torch.zeros(d, device=dev)
    t_real = timeit(lambda: hip.ss2d_merge_norm_cl(ys, order, lw, lb, 1e-5, 2, dtype))
    # even table: pixel p lists (k, p) for k = 0..7 -- 8 entries everywhere, rows of 8 different directions
    even = copy.copy(order)
    even.inv_ptr = torch.arange(0, 8 * l + 1, 8, dtype=torch.int32, device=dev)
    idx = (np.arange(l)[:, None] + np.arange(k)[None, :] * l).astype(np.int32)
    # permute the positions per direction so that rows are gathered, not streamed
    rng = np.random.default_rng(0)
    for kk in range(k):
        idx[:, kk] = kk * l + rng.permutation(l)
    even.inv_idx = torch.from_numpy(idx.reshape(-1)).to(dev)
    t_even = timeit(lambda: hip.ss2d_merge_norm_cl(ys, even, lw, lb, 1e-5, 2, dtype))
    mb = (b * k * l * d * 2 + b * l * d * 2) / 1e6
    print(f"merge helix {h}x{h} D={d} B={b}: real table {t_real:.1f} us ({mb / t_real:.2f} TB/s), "
          f"even table {t_even:.1f} us ({mb / t_even:.2f} TB/s)")


if __name__ == "__main__":
    main()
    main(h=48, d=512)
