#!/usr/bin/env python3
"""debug aid: training step with the guide branches on a side stream against the single-stream step, gradient by gradient"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tramba_amd as ta
from tramba_amd import hip, models, train


def run(overlap, defer, nsteps=3, sync=False, bs=8):
    models.OVERLAP_TRAINING = overlap
    train.DEFER_SUMS = defer
    torch.manual_seed(7)
    m = ta.bulid_model(use_pretrain=False, img_size=384).cuda().train()
    for mod in m.modules():
        if isinstance(mod, ta.DropPath):
            mod.drop_prob = 0.0
    m.compute_dtype = torch.bfloat16
    x = torch.randn(bs, 3, 384, 384, generator=torch.Generator().manual_seed(0)).cuda()
    y = (torch.rand(bs, 1, 384, 384, generator=torch.Generator().manual_seed(1)) > 0.7).float().cuda()
    opt = train.get_opt(1e-4, m)
    out = []
    orig_flush = hip.flush_sums
    if sync:
        def flush_synced():
            torch.cuda.synchronize()
            orig_flush()
            torch.cuda.synchronize()
        hip.flush_sums = flush_synced
    for i in range(nsteps):
        try:
            loss = float(train.train_step(m, opt, x, y))
        finally:
            pass
        torch.cuda.synchronize()
        out.append((loss, {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None},
                    {n: p.detach().clone() for n, p in m.named_parameters()}))
    hip.flush_sums = orig_flush
    return out


base = run(False, True)
for overlap, defer, sync in ((True, True, False), (True, True, True), (True, False, False)):
    got = run(overlap, defer, sync=sync)
    for i, ((l0, g0, w0), (l1, g1, w1)) in enumerate(zip(base, got)):
        badg = [n for n in g0 if not torch.equal(g0[n], g1[n])]
        badw = [n for n in w0 if not torch.equal(w0[n], w1[n])]
        nonf = [n for n in g1 if not torch.isfinite(g1[n]).all()]
        print(f"overlap={overlap} defer={defer} sync={sync} step {i}: loss {l1} (base {l0}); grads differing {len(badg)}, non-finite {len(nonf)}, weights differing {len(badw)}")
        for n in (nonf or badg)[:6]:
            d = (g1[n].float() - g0[n].float()).abs().max()
            print("     ", n, tuple(g1[n].shape), "max|d|", float(d))
models.OVERLAP_TRAINING = True
train.DEFER_SUMS = True
