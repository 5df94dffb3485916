"""The five heaviest GEMM shapes of a batch-4 Tramba-V forward (time x calls per forward, scripts/bench_gemm.py), six launches
each with their model epilogue, for `rocprofv3 --pmc` passes (scripts/pmc_gemm.sh -> profiles/<tag>_gemm_counters.json:
MFMA busy, SQ busy, waits, LDS bank conflicts -- the MFMA utilisation of the projections)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tramba_amd import hip
SHAPES = [(2304, 512, 2048), (2304, 2048, 512), (2304, 1024, 512), (2304, 512, 1024), (36864, 512, 128)]
dev = torch.device("cuda")
for (m, n, k) in SHAPES:
    x = torch.randn(m, k, device=dev).bfloat16()
    w = (torch.randn(n, k, device=dev) * k ** -0.5).bfloat16()
    b = torch.randn(n, device=dev)
    for _ in range(6):
        hip.linear_cl(x, w, b, None, 2)
torch.cuda.synchronize()
