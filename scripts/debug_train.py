import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tramba_amd as ta
from tramba_amd import train
torch.manual_seed(0)
DEV = "cuda"
m = ta.bulid_model(use_pretrain=False, img_size=384).to(DEV).train()
for mod in m.modules():
    if isinstance(mod, ta.DropPath):
        mod.drop_prob = 0.0
x = torch.randn(2, 3, 384, 384, generator=torch.Generator().manual_seed(0)).to(DEV)
y = (torch.rand(2, 1, 384, 384, generator=torch.Generator().manual_seed(1)) > 0.7).float().to(DEV)
loss0 = train.tramba_loss(m(x), y)
loss0.backward()
g = [p.grad.clone() for p in m.parameters()]
gn = float(torch.sqrt(sum((t.float() ** 2).sum() for t in g)))
print("loss0", float(loss0), "gnorm", gn)
# eval-mode (fused HIP) loss for the same weights, for reference
with torch.no_grad():
    print("no_grad loss", float(train.tramba_loss(m(x), y)))
base = [p.detach().clone() for p in m.parameters()]
for step in (1e-4, 1e-3, 1e-2):
    with torch.no_grad():
        for p, b, gg in zip(m.parameters(), base, g):
            p.copy_(b - (step / gn) * gg)
    l1 = float(train.tramba_loss(m(x), y).detach())
    print(f"step {step:g}: loss {l1:.6f}  actual decrease {float(loss0) - l1:.3e}  predicted {step * gn:.3e}")
# per-group norms
names = [n for n, _ in m.named_parameters()]
big = sorted(zip(names, g), key=lambda t: -float(t[1].float().norm()))[:8]
for n, t in big:
    print(n, float(t.float().norm()))
