#!/usr/bin/env python3
"""debug aid: which variation removes the NaN of the overlapped training step (batch 8, immediate sums)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tramba_amd as ta
from tramba_amd import hip, models, train

mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
models.OVERLAP_TRAINING = True   # (off by default: this script reproduces why)
train.DEFER_SUMS = False
torch.manual_seed(7)
m = ta.bulid_model(use_pretrain=False, img_size=384).cuda().train()
for mod in m.modules():
    if isinstance(mod, ta.DropPath):
        mod.drop_prob = 0.0
m.compute_dtype = torch.bfloat16
x = torch.randn(8, 3, 384, 384, generator=torch.Generator().manual_seed(0)).cuda()
y = (torch.rand(8, 1, 384, 384, generator=torch.Generator().manual_seed(1)) > 0.7).float().cuda()
opt = train.get_opt(1e-4, m)
models._DEBUG_STREAMS = []
if mode.startswith("tn"):          # no LDS-DMA GEMM kernels: register-staged weight-gradient and projection kernels (tune 8)
    hip.tune_set(hip.TUNE_GEMM_TILE, 8)
if mode.startswith("t18"):         # the r03 LDS-DMA projection kernels (no producer / consumer, no weight-stationary form)
    hip.tune_set(hip.TUNE_GEMM_TILE, 18)
if mode.startswith("wg"):          # only the weight-gradient GEMM on its register-staged kernel
    hip.tune_set(hip.TUNE_WGRAD_FORM, 1)
if mode.startswith("sc"):          # the fused scans on the register ring (no LDS-DMA scan kernel)
    hip.tune_set(hip.TUNE_SCAN_FORM, 1)
if mode.startswith("st"):          # st0 / st1 / st2 / st01 ...: only these guide branches run on the side stream
    models._OVERLAP_TRAINING_STAGES = {int(c) for c in mode[2:].split("_")[0]}
print("main stream", torch.cuda.current_stream().cuda_stream, "side", models._side_stream(x.device).cuda_stream)
if mode == "anomaly":
    torch.autograd.set_detect_anomaly(True, check_nan=True)
import contextlib
ctxm = torch.cuda.stream(torch.cuda.Stream()) if mode.startswith("ns") else contextlib.nullcontext()
if mode.startswith("ns"):
    torch.cuda.synchronize()
for i in range(4):
    try:
        with ctxm:
            loss = float(train.train_step(m, opt, x, y))
    except Exception as e:
        print("step", i, "raised:", str(e)[:1500])
        break
    torch.cuda.synchronize()
    bad = [n for n, p in m.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    if i == 0:
        print("stream edges seen in backward (node stream, consumer):", models._DEBUG_STREAMS)
    print(f"[{mode}] step {i}: loss {loss:.6f}, non-finite grads {len(bad)} {bad[:4]}", flush=True)
