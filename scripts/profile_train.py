import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tramba_amd as ta
from tramba_amd import parallel, train
torch.manual_seed(1026)
m = ta.bulid_model(use_pretrain=False, img_size=384).cuda().train()
m.compute_dtype = torch.bfloat16
red = parallel.GradBucketReducer(m)
opt = train.get_opt(1e-4, m)
b = int(sys.argv[1]) if len(sys.argv) > 1 else 8
x = torch.randn(b, 3, 384, 384).cuda()
y = (torch.rand(b, 1, 384, 384) > 0.7).float().cuda()
for _ in range(4):       # the summary keeps the last one (scripts/train_trace_steps.py)
    train.train_step(m, opt, x, y, reducer=red)
torch.cuda.synchronize()
