#!/usr/bin/env python3
"""Which host lines launch the non-tramba kernels / copies of the inference forward?  torch.profiler + Python stacks."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tramba_amd as ta
from tramba_amd import models
from torch.profiler import profile, ProfilerActivity
models.OVERLAP_BRANCHES = False
torch.manual_seed(1026)
m = ta.bulid_model(deep_supervision=True, use_pretrain=False, img_size=384).cuda().eval()
m = ta.prepare_inference(m, torch.bfloat16)
x = torch.randn(4, 3, 384, 384).cuda()
with torch.no_grad():
    for _ in range(3):
        m(x)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        m(x)
        torch.cuda.synchronize()
evs = prof.key_averages(group_by_stack_n=14)
rows = []
for e in evs:
    t = getattr(e, "self_device_time_total", None) or getattr(e, "self_cuda_time_total", 0)
    if t <= 0:
        continue
    st = [s for s in (e.stack or []) if "tramba_amd" in s]
    rows.append((t, e.key, e.count, " <- ".join(x.split("/")[-1] for x in st[:4])))
rows.sort(reverse=True)
for t, k, n, st in rows[:60]:
    print(f"{t/1e3:8.3f} ms n={n:4d} {k[:44]:44s} {st[:220]}")
