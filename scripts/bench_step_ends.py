#!/usr/bin/env python3
"""The two ends of the training step outside the network, library kernels against the framework's: Adam over Tramba-V's
parameter list (tramba_adam_step against torch.optim.Adam(fused=True)) and the deep-supervision loss forward + backward at
the benchmarked batch-8 shapes (tramba_sod_loss_* against the resize / BCE / IoU formula under autograd).  Each timed as a
hipGraph replay.   python3 scripts/bench_step_ends.py"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tramba_amd as ta  # noqa: E402
from tramba_amd import train  # noqa: E402


def replay_us(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


m = ta.bulid_model(use_pretrain=False, img_size=384).cuda()
params = list(m.parameters())
nparam = sum(p.numel() for p in params)
for p in params:
    p.grad = torch.randn_like(p) * 1e-3
for name, opt in (("library", train.Adam(params, 1e-4, capturable=True)),
                  ("torch fused", torch.optim.Adam(params, 1e-4, capturable=True, fused=True))):
    us = replay_us(opt.step)
    print(f"Adam, {len(params)} tensors / {nparam / 1e6:.1f} M parameters, {name}: {us:.0f} us = {28 * nparam / us / 1e6:.2f} TB/s "
          f"of 28 B per parameter", flush=True)
    del opt

b = 8
outs = [torch.randn(b, 1, s, s, device="cuda", requires_grad=True) for s in (24, 48, 96, 384)]
lab = (torch.rand(b, 1, 384, 384, device="cuda") > 0.7).float()


def framework_loss():
    total = 0
    for o in outs:
        r = o if o.shape[-1] == 384 else train._UpsampleBilinearHIP.apply(o, (384, 384))
        total = total + F.binary_cross_entropy_with_logits(r, lab) + train.iou_loss(r, lab)
    return total


def step(loss_fn):
    for o in outs:
        o.grad = None
    loss_fn().backward()


print(f"loss forward + backward, batch {b}, 4 outputs: library {replay_us(lambda: step(lambda: train.tramba_loss(outs, lab))):.0f} us, "
      f"framework formula {replay_us(lambda: step(framework_loss)):.0f} us", flush=True)
