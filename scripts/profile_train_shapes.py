#!/usr/bin/env python3
"""Which tensors do the training step's small torch ops (fill_, add_, copy_, sum, index_add_) touch?  torch.profiler, shapes."""
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    train.train_step(m, opt, x, y, reducer=red)
    torch.cuda.synchronize()
want = ("aten::fill_", "aten::zero_", "aten::add_", "aten::add", "aten::copy_", "aten::sum", "aten::index_add_", "aten::mul", "aten::_to_copy")
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if e.name in want:
        key = (e.name, str(e.input_shapes)[:90])
        agg[key][0] += 1
        agg[key][1] += getattr(e, "device_time_total", 0) or getattr(e, "cuda_time_total", 0)
rows = sorted(agg.items(), key=lambda kv: -kv[1][0])
only = sys.argv[1:] or None
for (name, shp), (n, t) in (rows[:70] if not only else [r for r in rows if r[0][0] in only][:90]):
    print(f"n={n:4d} {t/1e3:8.2f} ms {name:18s} {shp}")
