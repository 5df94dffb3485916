#!/usr/bin/env python3
"""Tramba hot-path benchmark on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one forward of Tramba-V (VMamba-B encoder + Dual-Frequency/Helix decoder) over one
synthetic batch already resident in HBM (BASELINE.json configs[1]: 384x384, bf16, batch 4 per GPU).
Inference shards by image: every rank runs an independent replica on its own batch, no data-path
collective ("weak" scaling); the barrier only brackets the timed region.

Rank 0 prints ONE JSON line.  Extra objects on that line:
  roofline      the dominant kernel launch of the step -- the fused channels-last selective scan on the Helix order at
                96x96 (K = 8, D = 256, B = 4; 2 launches per forward, the largest single item): algorithmic bytes per
                launch / average launch duration (one HIP-event pair around 20 back-to-back launches on the launch
                stream), against 8 TB/s HBM peak; `traffic` = PMC bytes of the same launch (profiles/*_traffic.json)
  roofline_fused_scan_all   every fused-scan launch of a forward (33, all shapes) timed inside an eager pass of the model
  roofline_boundary   the reference-layout selective scan (the L0 drop-in op, 8 B/element at bf16-in/fp32-out, SURVEY
                8d) on the largest call shape of this model
  roofline_gemm       the 1x1-conv projections (MFMA): 2*M*N*K of every GEMM launch of a forward / their time
  cpu_baseline  the CPU oracle (a port of the reference forward, oracle/) timed on this host
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
MFMA_PEAK_TFS = 2500.0  # dense bf16 MFMA peak (no 2:1 sparsity)


def pmc_traffic(key):
    """HBM bytes per launch from the newest committed PMC summary (profiles/*_traffic.json, produced by
    scripts/collect_profiles.sh + scripts/summarize_profiles.py with separate FETCH_SIZE / WRITE_SIZE passes and
    the gfx950 x2 FETCH_SIZE correction).  None when no summary covers this kernel shape."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json"))):
        try:
            d = json.load(open(path)).get(key)
        except Exception:
            d = None
        if d and "traffic_bytes" in d:
            best = {"bytes": round(d["traffic_bytes"]), "algorithmic_bytes": d.get("algorithmic_bytes"),
                    "source": os.path.relpath(path, ROOT), "what": d.get("what")}
    return best


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=4, help="images per GPU per step (config: 4)")
    ap.add_argument("--img", type=int, default=384)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--train", action="store_true",
                    help="also time the training step (fwd+bwd+Adam, DP all-reduce when N>1) and report it as 'train'")
    ap.add_argument("--train-batch", type=int, default=8, help="images per GPU per training step (config: 8)")
    return ap.parse_args()


def build_model(img, dtype):
    import tramba_amd as ta
    torch.manual_seed(1026)  # train.py:284
    m = ta.bulid_model(deep_supervision=True, use_pretrain=False, img_size=img, dims=128, depths=[2, 2, 2, 2])
    return ta.prepare_inference(m.cuda(), dtype)


def cpu_baseline(img):
    """Reference forward as restated by the oracle (kind "port"), bounded sample on the host."""
    import synth
    from oracle import model as om
    from oracle import ops as oo
    import tramba_amd as ta
    torch.manual_seed(1026)
    m = ta.bulid_model(use_pretrain=False, img_size=img)
    sd = {k: v.detach() for k, v in m.state_dict().items()}
    x = torch.randn(1, 3, img, img, generator=torch.Generator().manual_seed(0))
    cores = torch.get_num_threads()
    with torch.no_grad():
        om.tramba_v(sd, x)  # warm-up (builds tables, loads the C scan)
        n, t0 = 0, time.perf_counter()
        while n < 2 or (time.perf_counter() - t0 < 10.0 and n < 8):
            om.tramba_v(sd, x)
            n += 1
        dt = time.perf_counter() - t0
    return {"value": round(n / dt, 4), "unit": "img/s", "cores": cores, "kind": "port",
            "sample": f"{n} forward passes of Tramba-V {img}x{img}, batch 1, fp32, oracle/model.py "
                      f"(torch-CPU ops + OpenMP C scan)"}


def boundary_scan_roofline(dtype):
    """Op-level run of the L0 selective scan on this model's largest call shape (4,1024,9216)."""
    from tramba_amd import hip
    dev = torch.device("cuda")
    nb, kd, k, l = 4, 1024, 4, 9216
    g = torch.Generator(device="cpu").manual_seed(0)
    u = torch.randn(nb, kd, l, generator=g).to(dev, dtype)
    delta = (0.5 * torch.randn(nb, kd, l, generator=g)).to(dev, dtype)
    A = -torch.ones(kd, 1, device=dev)
    B = torch.randn(nb, k, 1, l, generator=g).to(dev, dtype)
    C = torch.randn(nb, k, 1, l, generator=g).to(dev, dtype)
    D = torch.ones(kd, device=dev)
    bias = torch.full((kd,), -3.0, device=dev)
    for _ in range(5):
        hip.selective_scan_fwd(u, delta, A, B, C, D, bias, True, True, want_ckpt=False)
    torch.cuda.synchronize()
    # algorithmic bytes of one launch from the library's own accounting ...
    hip.profile_enable(hip.PROF_SCAN_BOUNDARY, True)
    hip.selective_scan_fwd(u, delta, A, B, C, D, bias, True, True, want_ckpt=False)
    _, _, bytes_per_launch = hip.profile_read(hip.PROF_SCAN_BOUNDARY)
    hip.profile_enable(hip.PROF_SCAN_BOUNDARY, False)
    # ... and the average launch duration from ONE pair of HIP events around n back-to-back launches on the launch
    # stream (a pair per launch adds the event markers' own ~8 us of serialisation to a 60 us kernel and no longer
    # agrees with the rocprofv3 kernel trace)
    n = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        hip.selective_scan_fwd(u, delta, A, B, C, D, bias, True, True, want_ckpt=False)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    nbytes = bytes_per_launch * n
    gbs = nbytes / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": "selective_scan_fwd_kernel", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
            "traffic": (pmc_traffic("selective_scan_fwd_kernel@grid262144") or {}).get("bytes"),
            "traffic_detail": pmc_traffic("selective_scan_fwd_kernel@grid262144"),
            "shape": [nb, kd, l], "launches": n, "avg_us": round(ms / n * 1e3, 2)}


def helix_scan_roofline(step, nrep):
    """The Helix-SS2D launch of the fused scan kernel at the decoder's 96x96 stage (K = 8, D = 256, B = 4): the largest
    single (kernel, shape) item of the forward (2 launches, ~0.24 ms of 5.3), timed INSIDE eager forwards of the model:
    the library's launch profiler (one HIP-event pair per launch, on the launch stream) is restricted to launches
    accounting for >= 300 MB, which only this shape does (335 MB; the next largest fused scan is 177 MB)."""
    from tramba_amd import hip
    hip.profile_min_units(hip.PROF_SCAN_FUSED, 300e6)
    hip.profile_enable(hip.PROF_SCAN_FUSED, True)
    for _ in range(nrep):
        step()
    n, ms, nbytes = hip.profile_read(hip.PROF_SCAN_FUSED)
    hip.profile_enable(hip.PROF_SCAN_FUSED, False)
    hip.profile_min_units(hip.PROF_SCAN_FUSED, 0.0)
    if n == 0:
        return None
    gbs = nbytes / (ms * 1e-3) / 1e9
    pmc = pmc_traffic("ss2d_scan_cl_kernel@grid131072")
    return {"bound": "hbm", "kernel": "ss2d_scan_cl_kernel (Helix-SS2D launch: 96x96, K=8, D=256, B=4, ys f32)",
            "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
            "traffic": pmc["bytes"] if pmc else None, "traffic_detail": pmc, "algorithmic_bytes": int(nbytes / n),
            "launches": n, "avg_us": round(ms / n * 1e3, 2),
            "note": "per launch: algorithmic bytes (x once + x_proj rows + ys) / average HIP-event duration of this "
                    "shape's launches inside eager single-stream forwards of the model; traffic = PMC bytes of the same "
                    "launch (separate FETCH_SIZE / WRITE_SIZE passes); all 33 fused-scan launches of a forward together: "
                    "roofline_fused_scan_all"}


def bench_train(args, world, rank, dtype):
    """BASELINE configs[2]/[3]: fwd + bwd (BCE+IoU on 4 outputs) + two-group Adam, batch 8 per GPU,
    gradients all-reduced over RCCL in 32 MB buckets overlapped with backward when world > 1."""
    import tramba_amd as ta
    from tramba_amd import parallel, train
    torch.manual_seed(1026)
    model = ta.bulid_model(use_pretrain=False, img_size=args.img).cuda().train()
    model.compute_dtype = None if dtype == torch.float32 else dtype
    if world > 1:
        parallel.broadcast_parameters(model, src=0)
    red = parallel.GradBucketReducer(model)
    opt = train.get_opt(1e-4, model)
    b = args.train_batch
    x = torch.randn(b, 3, args.img, args.img, generator=torch.Generator().manual_seed(100 + rank)).cuda()
    y = (torch.rand(b, 1, args.img, args.img, generator=torch.Generator().manual_seed(200 + rank)) > 0.7).float().cuda()
    steps, warm = max(5, args.steps // 2), 3
    for _ in range(warm):
        train.train_step(model, opt, x, y, reducer=red)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        train.train_step(model, opt, x, y, reducer=red)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], device="cuda")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    graphed = None
    if world == 1:      # the same step replayed as ONE hipGraph (single process: no collective inside the capture)
        try:
            import tramba_amd as ta
            gstep = ta.GraphedTrainStep(model, train.get_opt(1e-4, model, capturable=True))
            for _ in range(2):
                gstep(x, y)
            torch.cuda.synchronize()
            g0 = time.perf_counter()
            for _ in range(steps):
                gstep(x, y)
            torch.cuda.synchronize()
            gdt = time.perf_counter() - g0
            graphed = {"value": round(b * steps / gdt, 2), "unit": "img/s", "ms_per_step": round(gdt / steps * 1e3, 2),
                       "what": "tramba_amd.GraphedTrainStep: forward + loss + backward + Adam + weight-shadow refresh "
                               "as one hipGraph replay"}
        except Exception as e:  # an optimisation, never a requirement
            graphed = {"error": f"{type(e).__name__}: {e}"[:300]}
    return {"metric": "images/sec fwd+bwd+Adam Tramba-V 384x384", "value": round(world * b * steps / dt, 2),
            "unit": "img/s", "steps": steps, "ms_per_step": round(dt / steps * 1e3, 2), "batch_per_gpu": b,
            "grad_bytes_per_step": red.bytes_per_step(), "stochastic_depth": "on (0.6 enc / 0.2 dec)",
            "dtype": args.dtype + " activations, fp32 master weights", "graphed": graphed}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    if args.gpus != world and rank == 0 and world == 1 and args.gpus > 1:
        print(f"bench.py: --gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks", file=sys.stderr)
        sys.exit(2)
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    from tramba_amd import hip
    dtype = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[args.dtype]
    model = build_model(args.img, dtype)
    x = torch.randn(args.batch, 3, args.img, args.img, generator=torch.Generator().manual_seed(rank)).cuda()

    def step():
        with torch.no_grad():
            return model(x)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # ---- optional hipGraph capture of the whole forward (launch-bound otherwise)
    graph = None
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    if not args.no_graph:
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out_static = step()
        except Exception as e:  # graph capture is an optimisation, never a requirement
            if rank == 0:
                print(f"bench.py: hipGraph capture unavailable ({type(e).__name__}: {e}); running eager", file=sys.stderr)
            graph = None
            torch.cuda.synchronize()
    run = (lambda: graph.replay()) if graph is not None else step

    for _ in range(args.warmup):
        run()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    sync_all()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], device="cuda")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    value = world * args.batch * args.steps / dt

    # ---- roofline of the dominant kernel: eager pass of the same steps with HIP events around
    #      every launch of the fused scan kernel (events cannot live inside a captured graph)
    roof = roof_all = roof_b = roof_g = cpu = None
    if rank == 0:
        from tramba_amd import models as _models
        overlap_was = _models.OVERLAP_BRANCHES
        _models.OVERLAP_BRANCHES = False      # one stream: a launch is timed alone, like the rocprofv3 trace
        hip.profile_enable(hip.PROF_SCAN_FUSED, True)
        hip.profile_enable(hip.PROF_GEMM, True)
        nrep = min(args.steps, 10)
        for _ in range(nrep):
            step()
        n, ms, nbytes = hip.profile_read(hip.PROF_SCAN_FUSED)
        ng, msg, flops = hip.profile_read(hip.PROF_GEMM)
        hip.profile_enable(hip.PROF_SCAN_FUSED, False)
        hip.profile_enable(hip.PROF_GEMM, False)
        _models.OVERLAP_BRANCHES = overlap_was
        gbs = nbytes / (ms * 1e-3) / 1e9
        roof_all = {"bound": "hbm", "kernel": "ss2d_scan_cl_kernel + ss2d_seg_kernel, all shapes", "achieved": round(gbs, 1),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None,
                    "launches": n, "avg_us": round(ms / n * 1e3, 2), "ms_per_step": round(ms / nrep, 3),
                    "note": "sum of algorithmic bytes / sum of HIP-event time over every fused-scan launch of a step "
                            "(single-stream eager pass, one event pair per launch)"}
        _models.OVERLAP_BRANCHES = False
        roof = helix_scan_roofline(step, nrep)
        _models.OVERLAP_BRANCHES = overlap_was
        tfs = flops / (msg * 1e-3) / 1e12
        roof_g = {"bound": "mfma", "kernel": "linear_lean_kernel / linear_tiled_kernel (1x1-conv projections)",
                  "achieved": round(tfs, 1), "peak": MFMA_PEAK_TFS, "unit": "TFLOP/s", "frac": round(tfs / MFMA_PEAK_TFS, 4),
                  "traffic": None, "launches": ng, "avg_us": round(msg / ng * 1e3, 2), "ms_per_step": round(msg / nrep, 3),
                  "note": "2*M*N*K of every tramba_linear_cl launch of a step / their HIP-event time; these GEMMs are "
                          "small (M = 576..36864, K <= 4096): LDS- and latency-bound, far from the dense MFMA peak"}
        roof_b = boundary_scan_roofline(dtype)
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(args.img)
    train_obj = None
    if args.train:
        train_obj = bench_train(args, world, rank, dtype)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        line = {
            "metric": "images/sec fwd Tramba-V 384x384", "value": round(value, 2), "unit": "img/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"Tramba-V (VMamba-B encoder) {args.img}x{args.img} {args.dtype} inference, "
                                   f"batch {args.batch} per GPU, random-init weights (seed 1026), randn images",
                       "global_batch": args.batch * world, "parallelism": f"dp{world} replicas, no collective",
                       "launch": ("hipGraph replay" if graph is not None else "eager") +
                                 (", decoder guide branches on a side stream"
                                  if os.environ.get("TRAMBA_OVERLAP", "1") != "0" else "")},
            "roofline": roof, "roofline_fused_scan_all": roof_all, "roofline_boundary": roof_b, "roofline_gemm": roof_g,
            "cpu_baseline": cpu,
        }
        if train_obj is not None:
            line["train"] = train_obj
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
