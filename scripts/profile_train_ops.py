#!/usr/bin/env python3
"""Which torch glue ops does one training step launch?  Counts of aten ops grouped by input shapes (torch.profiler)."""
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
with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
    train.train_step(m, opt, x, y, reducer=red)
    torch.cuda.synchronize()
want = sys.argv[1:] or ["aten::fill_", "aten::zero_", "aten::copy_", "aten::add", "aten::add_", "aten::mul", "aten::sum", "aten::index",
                        "aten::_to_copy", "aten::contiguous", "aten::clone", "aten::zeros", "aten::zeros_like", "aten::new_zeros",
                        "aten::empty", "aten::empty_like"]
tot = collections.Counter()
by = collections.defaultdict(collections.Counter)
for e in prof.events():
    tot[e.name] += 1
    if e.name in want:
        by[e.name][str(e.input_shapes)[:110]] += 1
print("top ops by count:", tot.most_common(40))
for name in want:
    print(f"== {name}: {sum(by[name].values())}")
    for shp, c in by[name].most_common(14):
        print(f"   {c:5d}  {shp}")
