#!/usr/bin/env python3
"""Which source lines of tramba_amd launch the torch glue kernels of a training step, and what do they cost on the device?
torch.profiler with stacks: aten ops grouped by (op, innermost tramba_amd frame), sorted by device time."""
import collections, os, sys, torch
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    train.train_step(m, opt, x, y, reducer=red)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    dev = getattr(e, "self_device_time_total", 0) or 0
    if dev <= 0 or not e.name.startswith("aten::"):
        continue
    frame = "?"
    for f in (e.stack or []):
        if "tramba_amd/" in f:
            frame = f.split("tramba_amd/")[-1]
            break
    if frame == "?":      # no Python frame (autograd thread): the operand shapes identify the call site
        frame = str(e.input_shapes)[:120]
    k = (e.name, frame)
    agg[k][0] += 1
    agg[k][1] += dev
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for v in agg.values())
print(f"aten device time in one step: {tot / 1e3:.2f} ms over {sum(v[0] for v in agg.values())} ops")
for (name, frame), (n, t) in rows[:110]:
    print(f"{t / 1e3:7.3f} ms {n:5d}  {name:34s} {frame}")
