#!/usr/bin/env python3
"""bf16 parameter gradients of Tramba-V against the oracle's autograd (the comparison of tests/test_gpu_grad.py), worst
tensors by norm ratio / cosine, for both forms of the DCT backward's intermediate and for two seeds of the input."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import tramba_amd as ta
from tramba_amd import train, modules
from oracle import model as om
from test_gpu_grad import _oracle_grads

torch.manual_seed(0)
m = ta.bulid_model(use_pretrain=False, img_size=384)
for mod in m.modules():
    if isinstance(mod, ta.DropPath):
        mod.drop_prob = 0.0
x = torch.randn(1, 3, 384, 384, generator=torch.Generator().manual_seed(0))
label = (torch.rand(1, 1, 384, 384, generator=torch.Generator().manual_seed(1)) > 0.7).float()
sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
_, gp_ref, _ = _oracle_grads(lambda s, xx: om.tramba_v(s, xx), sd, x, label=label)
m = m.cuda().train()
m.compute_dtype = torch.bfloat16
for flag in (True, False):
    modules.DCT_BWD_LOWP_INTERMEDIATE = flag
    m.zero_grad(set_to_none=True)
    loss = train.tramba_loss(m(x.cuda()), label.cuda())
    loss.backward()
    rows = []
    for n, p in m.named_parameters():
        g, ref = p.grad.double().cpu(), gp_ref[n]
        rn = float(ref.norm())
        if rn == 0.0:
            continue
        cos = float((g * ref).sum() / (g.norm() * ref.norm()).clamp_min(1e-300))
        rows.append((abs(float(g.norm()) / rn - 1.0), 1.0 - cos, n))
    print(f"bf16 intermediate = {flag}: worst by norm ratio:", [(round(a, 4), n) for a, _, n in sorted(rows, reverse=True)[:5]])
    print("   worst by cosine:", [(round(1 - c, 5), n) for _, c, n in sorted(rows, key=lambda r: -r[1])[:5]], flush=True)
