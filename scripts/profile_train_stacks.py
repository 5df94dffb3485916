#!/usr/bin/env python3
"""Where do the training step's elementwise copies / adds come from?  torch.profiler with Python stacks."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tramba_amd as ta
from tramba_amd import parallel, train
from torch.profiler import profile, ProfilerActivity
torch.manual_seed(1026)
m = ta.bulid_model(use_pretrain=False, img_size=384).cuda().train()
m.compute_dtype = torch.bfloat16
opt = train.get_opt(1e-4, m)
x = torch.randn(8, 3, 384, 384).cuda()
y = (torch.rand(8, 1, 384, 384) > 0.7).float().cuda()
for _ in range(2):
    train.train_step(m, opt, x, y)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    train.train_step(m, opt, x, y)
    torch.cuda.synchronize()
evs = prof.key_averages(group_by_stack_n=12)
rows = []
for e in evs:
    t = getattr(e, "self_device_time_total", None) or getattr(e, "self_cuda_time_total", 0)
    if t <= 0:
        continue
    st = [s for s in (e.stack or []) if "tramba_amd" in s or "oracle" in s]
    rows.append((t, e.key, e.count, " <- ".join(x.split("/")[-1] for x in st[:4])))
rows.sort(reverse=True)
for t, k, n, st in rows[:45]:
    print(f"{t/1e3:8.2f} ms n={n:4d} {k[:48]:48s} {st[:200]}")
