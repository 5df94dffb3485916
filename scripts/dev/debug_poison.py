#!/usr/bin/env python3
"""debug aid: does any kernel of the SINGLE-stream training step read memory it has not written?  Every torch.empty /
empty_like result is filled with NaN (0xFF bytes) first; the step must give the bit-identical gradients."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tramba_amd as ta
from tramba_amd import hip, models, train

models.OVERLAP_TRAINING = False
train.DEFER_SUMS = False if (len(sys.argv) > 1 and sys.argv[1] == "nodefer") else True


def build():
    torch.manual_seed(7)
    m = ta.bulid_model(use_pretrain=False, img_size=384).cuda().train()
    for mod in m.modules():
        if isinstance(mod, ta.DropPath):
            mod.drop_prob = 0.0
    m.compute_dtype = torch.bfloat16
    return m, train.get_opt(1e-4, m)


x = torch.randn(8, 3, 384, 384, generator=torch.Generator().manual_seed(0)).cuda()
y = (torch.rand(8, 1, 384, 384, generator=torch.Generator().manual_seed(1)) > 0.7).float().cuda()


def steps(m, opt, n=2):
    out = []
    for _ in range(n):
        loss = float(train.train_step(m, opt, x, y))
        torch.cuda.synchronize()
        out.append((loss, {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}))
    return out


m, opt = build()
base = steps(m, opt)
del m, opt
_empty, _empty_like = torch.empty, torch.empty_like


def poison(t):
    if t.is_cuda and t.numel() and t.is_contiguous():
        if t.is_floating_point():
            t.fill_(float("nan"))
        else:
            t.reshape(-1).view(torch.uint8).fill_(0x7F)
    return t


torch.empty = lambda *a, **k: poison(_empty(*a, **k))
torch.empty_like = lambda *a, **k: poison(_empty_like(*a, **k))
m, opt = build()
got = steps(m, opt)
torch.empty, torch.empty_like = _empty, _empty_like
for i, ((l0, g0), (l1, g1)) in enumerate(zip(base, got)):
    bad = [k for k in g0 if not torch.equal(g0[k], g1[k])]
    nonf = [k for k in g1 if not torch.isfinite(g1[k]).all()]
    print(f"step {i}: loss {l1} (unpoisoned {l0}); gradients differing {len(bad)}, non-finite {len(nonf)}")
    for k in (nonf or bad)[:20]:
        print("    ", k, tuple(g1[k].shape))
