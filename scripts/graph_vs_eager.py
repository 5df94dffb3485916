#!/usr/bin/env python3
"""Per-image latency of the evaluation loop's forward (batch 1, as test_TSOD.py runs it): eager launches against
`tramba_amd.GraphedForward` replay.   python3 scripts/graph_vs_eager.py [batch]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tramba_amd as ta  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
torch.manual_seed(1026)
m = ta.prepare_inference(ta.bulid_model(use_pretrain=False, img_size=384).cuda(), torch.bfloat16)
x = torch.randn(batch, 3, 384, 384).cuda()


def timed(fn, n=50):
    for _ in range(5):
        fn(x)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn(x)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


def eager(v):
    with torch.no_grad():
        return m(v)


e = timed(eager)
g = timed(ta.GraphedForward(m, strict=True))
print(f"batch {batch}: eager {e:.2f} ms  graph {g:.2f} ms  ({e / g:.2f}x)  {batch / g * 1e3:.0f} img/s graphed")
