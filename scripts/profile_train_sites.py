#!/usr/bin/env python3
"""Which tramba_amd source lines issue the training step's small torch ops?  CPU-side op events with Python stacks."""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tramba_amd as ta
from tramba_amd import parallel, train
from torch.profiler import profile, ProfilerActivity
torch.manual_seed(1026)
m = ta.bulid_model(use_pretrain=False, img_size=384).cuda().train()
m.compute_dtype = torch.bfloat16
opt = train.get_opt(1e-4, m)
red = parallel.GradBucketReducer(m)
x = torch.randn(8, 3, 384, 384).cuda()
y = (torch.rand(8, 1, 384, 384) > 0.7).float().cuda()
for _ in range(2):
    train.train_step(m, opt, x, y, reducer=red)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    train.train_step(m, opt, x, y, reducer=red)
    torch.cuda.synchronize()
want = set(sys.argv[1:]) or {"aten::copy_", "aten::fill_", "aten::add_", "aten::add", "aten::sum", "aten::mul", "aten::index_add_", "aten::index"}
agg = collections.Counter()
for e in prof.events():
    if e.name in want:
        site = next((s for s in (e.stack or []) if "tramba_amd" in s), "(autograd / torch internals)")
        agg[(e.name, site.split("/")[-1][:110])] += 1
for (name, site), n in agg.most_common(60):
    print(f"n={n:4d} {name:18s} {site}")
