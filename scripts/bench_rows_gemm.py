import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tramba_amd import hip
dev = torch.device("cuda")
for (z, m, d, r) in ((32, 576, 1024, 32), (32, 2304, 512, 16), (64, 2304, 512, 16), (64, 576, 1024, 32), (64, 9216, 256, 8), (16, 144, 2048, 64)):
    x = torch.randn(z, m, d, device=dev).bfloat16()
    w = torch.randn(4, r, d, device=dev).bfloat16()
    y = torch.zeros(z, m, r + 4, device=dev)
    fn = lambda: hip.rows_gemm_cl(x, w, y, r)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
    print(f"z={z} m={m} D={d} R={r}: {e0.elapsed_time(e1)/40*1e3:.1f} us  ({z*m*d*2/1e6:.1f} MB)", flush=True)
