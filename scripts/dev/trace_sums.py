#!/usr/bin/env python3
"""Which call sites of one training step run a partial-sum reduction IMMEDIATELY (tramba_slab_sum / the slab sum inside tramba_wgrad_cl)
instead of recording it for the step's batched reductions?  usage: python scripts/dev/trace_sums.py [batch]"""
import collections, os, sys, traceback, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tramba_amd as ta
from tramba_amd import hip, parallel, train

rows = collections.Counter()


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "tramba_amd" in fr.filename and not fr.filename.endswith("hip.py"):
            return f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}"
    return "?"


_slab, _wg = hip.slab_sum, hip._wgrad


def slab_sum(part, defer=False):
    if not (defer and hip._sumq.enabled):
        rows[("slab_sum", tuple(part.shape), site())] += 1
    return _slab(part, defer)


def wgrad(gy, x, m, n, k, groups, nbatch, *a):
    defer = a[-1] if len(a) == 8 else False
    if not (defer and hip._sumq.enabled):
        rows[("wgrad", (m, n, k, groups, nbatch), site())] += 1
    return _wg(gy, x, m, n, k, groups, nbatch, *a)


hip.slab_sum, hip._wgrad = slab_sum, wgrad
torch.manual_seed(1026)
m = ta.bulid_model(use_pretrain=False, img_size=384).cuda().train()
m.compute_dtype = torch.bfloat16
opt = train.get_opt(1e-4, m)
b = int(sys.argv[1]) if len(sys.argv) > 1 else 8
x = torch.randn(b, 3, 384, 384).cuda()
y = (torch.rand(b, 1, 384, 384) > 0.7).float().cuda()
for _ in range(2):
    train.train_step(m, opt, x, y)
rows.clear()
train.train_step(m, opt, x, y)
torch.cuda.synchronize()
print(sum(rows.values()), "immediate reductions in one step")
for (kind, shape, s), n in sorted(rows.items(), key=lambda kv: -kv[1]):
    print(f"n={n:4d} {kind:9s} {str(shape):34s} {s}")
