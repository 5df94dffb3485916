#!/usr/bin/env python3
"""Which tramba_amd source lines issue the torch (non-library) ops of one training step?  A TorchDispatchMode logs every
aten op that runs a device kernel with the innermost tramba_amd frame of the Python stack (autograd-engine internals such as
gradient accumulation show up as "(engine)").  usage: python scripts/trace_glue.py [batch] [infer]   (infer: the bf16 inference
forward of bench.py instead of the training step)"""
import collections, os, sys, traceback, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tramba_amd as ta
from tramba_amd import parallel, train
from torch.utils._python_dispatch import TorchDispatchMode

SKIP = {"aten::view", "aten::_unsafe_view", "aten::reshape", "aten::t", "aten::transpose", "aten::permute", "aten::detach",
        "aten::expand", "aten::slice", "aten::select", "aten::unsqueeze", "aten::squeeze", "aten::as_strided", "aten::alias",
        "aten::empty", "aten::empty_like", "aten::empty_strided", "aten::new_empty", "aten::_local_scalar_dense", "aten::unbind",
        "aten::split", "aten::split_with_sizes", "aten::view_as", "aten::is_same_size", "aten::stride", "aten::size",
        "aten::lift_fresh", "aten::is_pinned", "aten::_has_compatible_shallow_copy_type", "aten::narrow", "aten::unfold"}


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.rows = collections.Counter()
        self.elems = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func._schema.name
        if name not in SKIP:
            site = "(engine)"
            for fr in reversed(traceback.extract_stack()[:-1]):
                if "tramba_amd" in fr.filename and "trace_glue" not in fr.filename:
                    site = f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}"
                    break
            n = 0
            for a in list(args) + [out]:
                if torch.is_tensor(a):
                    n = max(n, a.numel())
            self.rows[(name, site)] += 1
            self.elems[(name, site)] += n
        return out


if len(sys.argv) > 2 and sys.argv[2] == "infer":
    torch.manual_seed(1026)
    m = ta.prepare_inference(ta.bulid_model(deep_supervision=True, use_pretrain=False, img_size=384).cuda(), torch.bfloat16)
    x = torch.randn(int(sys.argv[1]), 3, 384, 384).cuda()
    with torch.no_grad():
        for _ in range(2):
            m(x)
        torch.cuda.synchronize()
        log = Log()
        with log:
            m(x)
    torch.cuda.synchronize()
    print(f"{sum(log.rows.values())} aten ops in one inference forward (views / allocations not counted)")
    for (name, site), n in sorted(log.rows.items(), key=lambda kv: -log.elems[kv[0]]):
        print(f"n={n:4d}  Melem={log.elems[(name, site)] / 1e6:9.1f}  {name:28s} {site}")
    sys.exit(0)

torch.manual_seed(1026)
m = ta.bulid_model(use_pretrain=False, img_size=384).cuda().train()
m.compute_dtype = torch.bfloat16
opt = train.get_opt(1e-4, m)
red = parallel.GradBucketReducer(m)
b = int(sys.argv[1]) if len(sys.argv) > 1 else 8
x = torch.randn(b, 3, 384, 384).cuda()
y = (torch.rand(b, 1, 384, 384) > 0.7).float().cuda()
for _ in range(2):
    train.train_step(m, opt, x, y, reducer=red)
torch.cuda.synchronize()
log = Log()
with log:
    train.train_step(m, opt, x, y, reducer=red)
torch.cuda.synchronize()
tot = sum(log.rows.values())
print(f"{tot} aten ops in one step (views / allocations not counted)")
for (name, site), n in sorted(log.rows.items(), key=lambda kv: -log.elems[kv[0]]):
    print(f"n={n:4d}  Melem={log.elems[(name, site)] / 1e6:9.1f}  {name:28s} {site}")
