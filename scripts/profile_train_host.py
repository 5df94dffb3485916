#!/usr/bin/env python3
"""Host-side cost of one eager training step: cProfile over 3 steps (the GPU runs behind; the step is host-bound when the
launch thread cannot keep ahead of ~46 ms of kernels)."""
import cProfile, os, pstats, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tramba_amd as ta
from tramba_amd import parallel, train
torch.manual_seed(1026)
m = ta.bulid_model(use_pretrain=False, img_size=384).cuda().train()
m.compute_dtype = torch.bfloat16
opt = train.get_opt(1e-4, m)
red = parallel.GradBucketReducer(m)
x = torch.randn(8, 3, 384, 384).cuda()
y = (torch.rand(8, 1, 384, 384) > 0.7).float().cuda()
for _ in range(3):
    train.train_step(m, opt, x, y, reducer=red)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    train.train_step(m, opt, x, y, reducer=red)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host issue time per step {(t1 - t0) / 5 * 1e3:.1f} ms; with drain {(t2 - t0) / 5 * 1e3:.1f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    train.train_step(m, opt, x, y, reducer=red)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
