#!/usr/bin/env python3
"""Training step, eager launches against `tramba_amd.GraphedTrainStep` replay (Tramba-V 384x384, batch 8, bf16
activations, stochastic depth on).   python3 scripts/graph_train.py [batch]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tramba_amd as ta  # noqa: E402
from tramba_amd import train  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 8
x = torch.randn(b, 3, 384, 384, generator=torch.Generator().manual_seed(0)).cuda()
y = (torch.rand(b, 1, 384, 384, generator=torch.Generator().manual_seed(1)) > 0.7).float().cuda()


def fresh(capturable):
    torch.manual_seed(1026)
    m = ta.bulid_model(use_pretrain=False, img_size=384).cuda().train()
    m.compute_dtype = torch.bfloat16
    return m, train.get_opt(1e-4, m, capturable=capturable)


def timed(fn, n=20):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        last = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3, float(last)


for defer in (True, False, True, False):       # the partial-sum reductions batched at the end of backward / where they are issued
    train.DEFER_SUMS = defer
    m, opt = fresh(False)
    e, le = timed(lambda: train.train_step(m, opt, x, y))
    del m, opt
    torch.cuda.empty_cache()
    m, opt = fresh(True)
    step = ta.GraphedTrainStep(m, opt)
    g, lg = timed(lambda: step(x, y))
    print(f"batch {b}, deferred sums {defer}: eager {e:.2f} ms (loss {le:.4f})  graph {g:.2f} ms (loss {lg:.4f})  {e / g:.2f}x  "
          f"{b / g * 1e3:.1f} img/s graphed", flush=True)
    del m, opt, step
    torch.cuda.empty_cache()
train.DEFER_SUMS = True
