#!/usr/bin/env python3
"""aten ops (= torch glue kernels) of one steady-state inference forward, by op and operand shapes, with device time."""
import collections, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tramba_amd as ta
from torch.profiler import profile, ProfilerActivity
torch.manual_seed(0)
m = ta.bulid_model(deep_supervision=True, use_pretrain=False, img_size=384).cuda().eval()
m = ta.prepare_inference(m, torch.bfloat16)
x = torch.randn(4, 3, 384, 384, device="cuda")
with torch.no_grad():
    for _ in range(4):
        m(x)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        m(x)
        torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    dev = getattr(e, "self_device_time_total", 0) or 0
    if dev <= 0:
        continue
    frame = ""
    for f in (e.stack or []):
        if "tramba_amd/" in f:
            frame = f.split("tramba_amd/")[-1]
            break
    k = (e.name, str(e.input_shapes)[:90], frame)
    agg[k][0] += 1
    agg[k][1] += dev
tot = sum(v[1] for v in agg.values())
print(f"profiled device time {tot / 1e3:.2f} ms")
for (name, shp, frame), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print(f"{t:8.1f} us {n:4d}  {name:28s} {shp:90s} {frame}")
