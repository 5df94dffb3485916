#!/usr/bin/env python3
"""Tramba hot-path benchmark on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself: the parent
process (which never touches the GPU) runs `python -m torch.distributed.run ... bench.py <same flags>` as a child and
exits with its code.

One "step" = one forward of Tramba-V (VMamba-B encoder + Dual-Frequency/Helix decoder) over one synthetic batch already
resident in HBM (BASELINE.json configs[1]: 384x384, bf16, batch 4 per GPU).  Inference shards by image: every rank runs
an independent replica on its own batch, no data-path collective ("weak" scaling); the barrier only brackets the timed
region.  `value` = images/s of that forward, whole job.

Rank 0 prints ONE JSON line.  Objects on that line besides the contract's fields:
  std_ms        standard deviation of the per-step device time (HIP events between the steps of the timed region)
  latency_b1    the reference's own latency protocol (Trambav6.py:219-255): batch 1, 50 warm-up + 500 timed forwards
                (`wall_clock`: its other protocol, test_TSOD.py:71-108 -- 5 + 195 synchronised forwards on the host's clock),
                one HIP-event pair per forward, mean / std / FPS
  roofline      the Helix-SS2D core at the decoder's top stage (img/4 squared positions: 96x96 at 384, 192x192 at 768; K = 8,
                D = 256; 2 per forward) as a PAIR of launches, fused scan + merge/out_norm: `achieved` / `peak` / `frac` =
                SURVEY 8(d)'s fused-Helix bytes (x once + low-rank x_proj rows + merged y) / the summed average duration of
                the two launches, timed by HIP events on the launch stream inside eager single-stream forwards of the
                model, against the HBM peak; `bound` = "valu": the pair is vector-issue bound, not HBM bound, and
                `valu` = {issue_floor_us, frac = floor / measured, ...} prices it against THAT ceiling from the committed SQ
                counters; `traffic` = PMC bytes of the pair (`traffic_source`: the committed summary it is read from)
  roofline_kernel_boundary   the same two launches at their own kernel boundaries (the K-fold `ys` intermediate counted
                where it is written and where it is read)
  roofline_fused_scan_all    every fused-scan launch of a forward (33, all shapes)
  roofline_boundary   the reference-layout selective scan (the L0 drop-in op, 8 B/element) on the largest call
                shape of this model
  roofline_gemm       the 1x1-conv projections (MFMA): 2*M*N*K of every GEMM launch of a forward / their time
  cpu_baseline  the CPU oracle (a port of the reference forward, oracle/) timed on this host (N = 1 only)
  train         BASELINE configs[2]/[3]: fwd + bwd + two-group Adam at batch 8 per GPU, gradients averaged over RCCL
                when N > 1 (bucketed, overlapped with backward), eager launches and -- when the capture succeeds on every
                rank -- the same step replayed as one hipGraph (`train.value` = the faster of the two, both reported as
                `train.eager` / `train.graphed`); `train.scaling_value` = `train.value` (THE figure a data-parallel scaling
                curve is about: the forward leg at N > 1 is independent replicas); `train.roofline` = the forward +
                input-gradient GEMMs of the step against the dense MFMA peak, `train.roofline_wgrad` = the weight-gradient
                GEMMs, `train.roofline_scan_bwd` = the fused scan backward (12 B per element), `train.roofline_layernorm` / `train.roofline_dw`
                = the two streaming families (LayerNorm forms and depth-wise stencils, HIP-event timed per call), `train.roofline_adam`
                = the optimizer step against the HBM peak (28 B per parameter), all launches
                and the Helix top-stage launch alone; each of the three carries `traffic` = PMC bytes per step of its kernel
                family at batch 8 (`traffic_source`: profiles/<tag>_train_traffic.json, scripts/pmc_train.sh);
                `train.step_ms_by_rank` = every rank's own time for the timed steps.  The leg runs under a watchdog
                (--train-timeout): if a collective never completes, rank 0 still prints the line, with train = {"error": ...}
"""
import argparse
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
MFMA_PEAK_TFS = 2500.0  # dense bf16 MFMA peak (no 2:1 sparsity)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=4, help="images per GPU per step (config: 4)")
    ap.add_argument("--img", type=int, default=384)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-full", action="store_true",
                    help="SURVEY 8d protocol for the CPU port: 3 warm-up + 10 timed passes on all cores, plus one "
                         "single-thread pass (minutes)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--no-overlap", action="store_true", help="one stream (no side stream for the guide branches)")
    ap.add_argument("--no-train", action="store_true", help="skip the training leg")
    ap.add_argument("--train", action="store_true", help="(default) kept for older command lines")
    ap.add_argument("--no-latency", action="store_true", help="skip the batch-1 latency protocol")
    ap.add_argument("--train-batch", type=int, default=8, help="images per GPU per training step (config: 8)")
    ap.add_argument("--bucket-dtype", default="fp32", choices=["fp32", "bf16"], help="gradient all-reduce buckets")
    ap.add_argument("--graph-dp", action="store_true", help="(default since round 3) kept for older command lines")
    ap.add_argument("--no-graph-dp", action="store_true",
                    help="with N > 1 do NOT attempt the data-parallel step as one hipGraph (RCCL all-reduces captured with it) "
                         "after the eager leg.  The attempt runs under the watchdog with the eager result already in the "
                         "line, so a rank stuck in a capture costs only the graphed figure")
    ap.add_argument("--train-timeout", type=float, default=420.0,
                    help="seconds the training leg (and the final barrier) may take before every rank gives up on it: rank 0 "
                         "then prints the line with the forward result and train = {error}, so that a stuck collective "
                         "cannot cost the run its forward measurement")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend (gloo only to rehearse the N > 1 plumbing on a one-GPU box)")
    return ap.parse_args()


def spawn_ranks(args):
    """--gpus N without a launcher: start the N ranks as a child job before this process touches the GPU."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # The child's stdout passes through this process: if rank 0 printed its line and a rank then ABORTED (a runtime abort inside a
    # captured collective takes the whole job down with a non-zero code), the measurement that was printed still stands -- exit 0
    # and say on stderr what happened (VERDICT r3 "next" 3c)
    return relay_child(cmd, env)


def relay_child(cmd, env=None):
    """run `cmd`, pass its stdout through; exit code 0 if it printed a result line, whatever it died of afterwards"""
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    have_line = False
    for out in proc.stdout:
        sys.stdout.write(out)
        sys.stdout.flush()
        if out.lstrip().startswith("{") and '"metric"' in out:
            have_line = True
    rc = proc.wait()
    if rc != 0 and have_line:
        print(f"bench.py: the rank job ended with code {rc} AFTER rank 0 had printed its line; exiting 0", file=sys.stderr)
        return 0
    return rc


def pmc_traffic(key):
    """HBM bytes per launch from the newest committed PMC summary (profiles/*_traffic.json, produced by
    scripts/collect_profiles.sh + scripts/summarize_profiles.py with separate FETCH_SIZE / WRITE_SIZE passes and
    the gfx950 x2 FETCH_SIZE correction).  None when no summary covers this kernel shape."""
    import glob
    best = None
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")))
    try:   # profiles/LATEST names the tag of the newest set (file names do not sort by time: r02i was collected after r02z)
        tag = open(os.path.join(ROOT, "profiles", "LATEST")).read().strip()
        paths.sort(key=lambda q: os.path.basename(q).startswith(tag + "_"))
    except OSError:
        pass
    for path in paths:
        try:
            d = json.load(open(path)).get(key)
        except Exception:
            d = None
        if d and "traffic_bytes" in d:
            best = {"bytes": round(d["traffic_bytes"]), "algorithmic_bytes": d.get("algorithmic_bytes"),
                    "source": os.path.relpath(path, ROOT), "what": d.get("what")}
    return best


def build_model(img, dtype):
    import torch
    import tramba_amd as ta
    torch.manual_seed(1026)  # train.py:284
    m = ta.bulid_model(deep_supervision=True, use_pretrain=False, img_size=img, dims=128, depths=[2, 2, 2, 2])
    return ta.prepare_inference(m.cuda(), dtype)


def host_cpu():
    """(model name, physical cores, logical cpus) of this host from /proc/cpuinfo"""
    model, phys, logical = None, set(), 0
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("processor"):
                logical += 1
            elif line.startswith("model name") and model is None:
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                pid = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                cid = line.split(":", 1)[1].strip()
                phys.add((pid, cid))
    except OSError:
        pass
    return model, (len(phys) or None), logical


def cpu_baseline(img, full):
    """Reference forward as restated by the oracle (kind "port") on the host: timed on all physical cores AND on one
    thread, `value` = the faster of the two with `cores` = the threads it used (on a 128-core host the single thread wins:
    the forward is thousands of small CPU ops whose fork-join cost grows with the team).  Default: a bounded sample (1
    warm-up + the passes that fit ~15 s, at least 3, then 2 single-thread passes); `full`: SURVEY 8d -- 3 warm-up + 10
    timed on all cores, 3 single-thread passes."""
    import torch
    from oracle import model as om
    import tramba_amd as ta
    torch.manual_seed(1026)
    m = ta.bulid_model(use_pretrain=False, img_size=img)
    sd = {k: v.detach() for k, v in m.state_dict().items()}
    x = torch.randn(1, 3, img, img, generator=torch.Generator().manual_seed(0))
    name, phys, logical = host_cpu()
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = logical
    threads = max(1, min(phys or usable, usable))

    def run(nthreads, warm, nmin, nmax, budget_s):
        torch.set_num_threads(nthreads)
        os.environ["OMP_NUM_THREADS"] = str(nthreads)
        with torch.no_grad():
            for _ in range(warm):
                om.tramba_v(sd, x)
            n, t0 = 0, time.perf_counter()
            while n < nmin or (n < nmax and time.perf_counter() - t0 < budget_s):
                om.tramba_v(sd, x)
                n += 1
            return n, time.perf_counter() - t0

    n_all, dt_all = run(threads, 3 if full else 1, 10 if full else 3, 10, 15.0)   # warm-up builds tables, loads the C scan
    n_one, dt_one = run(1, 0, 3 if full else 2, 3 if full else 2, 0.0)
    torch.set_num_threads(threads)
    all_cores = {"value": round(n_all / dt_all, 4), "unit": "img/s", "cores": threads, "passes": n_all}
    single = {"value": round(n_one / dt_one, 4), "unit": "img/s", "cores": 1, "passes": n_one}
    best = all_cores if all_cores["value"] >= single["value"] else single
    return {"value": best["value"], "unit": "img/s", "cores": best["cores"], "kind": "port",
            "cpu_model": name, "physical_cores": phys, "logical_cpus": logical, "usable_cpus": usable,
            "all_cores": all_cores, "single_thread": single,
            "sample": f"Tramba-V {img}x{img}, batch 1, fp32, oracle/model.py (torch-CPU ops + OpenMP C scan): {n_all} timed "
                      f"forward passes on {threads} threads after {3 if full else 1} warm-up, then {n_one} on one thread; "
                      f"value = the faster"
                      + ("" if full else " (bounded sample; SURVEY 8d's 3 + 10 protocol: --cpu-baseline-full, recorded in "
                                         "profiles/)")}


def boundary_scan_roofline(dtype, batch=4, img=384):
    """Op-level run of the L0 selective scan on this model's largest call shape -- (B, 1024, (img/4)^2): (4,1024,9216) at the
    BASELINE configuration, (2,1024,36864) at 768x768 batch 2: algorithmic bytes from the library's own accounting / the
    average launch duration from ONE pair of HIP events around 20 back-to-back launches replayed as a hipGraph (a pair per
    launch adds the markers' own ~8 us of serialisation to a 60 us kernel; eager launches add the host's)."""
    import torch
    from tramba_amd import hip
    dev = torch.device("cuda")
    nb, kd, k, l = batch, 1024, 4, (img // 4) ** 2
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
    hip.profile_enable(hip.PROF_SCAN_BOUNDARY, True)
    hip.selective_scan_fwd(u, delta, A, B, C, D, bias, True, True, want_ckpt=False)
    _, _, bytes_per_launch = hip.profile_read(hip.PROF_SCAN_BOUNDARY)
    hip.profile_enable(hip.PROF_SCAN_BOUNDARY, False)
    # 20 launches captured as ONE hipGraph and replayed: launched eagerly from Python the host's ~25 us per call sits between the
    # 60 us kernels on a slow box, and the event pair then times the host (r03: 0.48 by events against 0.56 in the rocprofv3
    # trace of the same kernel).  Falls back to eager launches if the capture fails.
    n = 20
    out = hip.selective_scan_fwd(u, delta, A, B, C, D, bias, True, True, want_ckpt=False)    # (output allocated outside the capture)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    how = "one HIP-event pair around a hipGraph of 20 launches, 3 replays"
    try:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for _ in range(n):
                keep = hip.selective_scan_fwd(u, delta, A, B, C, D, bias, True, True, want_ckpt=False)
        graph.replay()
        torch.cuda.synchronize()
        reps = 3
        e0.record()
        for _ in range(reps):
            graph.replay()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
    except Exception:
        how = "one HIP-event pair around 20 eager launches (graph capture failed)"
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n):
            hip.selective_scan_fwd(u, delta, A, B, C, D, bias, True, True, want_ckpt=False)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
    gbs = bytes_per_launch * n / (ms * 1e-3) / 1e9
    pmc = pmc_traffic("selective_scan_fwd_kernel@grid262144") if (nb, l, dtype != torch.float32) == (4, 9216, True) else None
    return {"bound": "hbm", "kernel": "selective_scan_fwd_kernel", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": (pmc or {}).get("bytes"),
            "traffic_source": (pmc or {}).get("source"), "traffic_detail": pmc, "shape": [nb, kd, l], "launches": n,
            "avg_us": round(ms / n * 1e3, 2), "timing": how}


def sq_valu_floor(key):
    """Vector-issue floor of one launch from the newest committed SQ-counter summary (profiles/*_sq_counters.json, scripts/
    pmc_sq.sh): SQ_ACTIVE_INST_VALU (quad-cycles summed over waves) is the time the SIMDs' vector pipes were occupied;
    spread over the chip's 1024 SIMDs at the 2.4 GHz peak clock it is the time the launch would take if nothing but vector
    issue limited it."""
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_sq_counters.json")))
    try:
        tag = open(os.path.join(ROOT, "profiles", "LATEST")).read().strip()
        paths.sort(key=lambda q: os.path.basename(q).startswith(tag + "_"))
    except OSError:
        pass
    for path in reversed(paths):
        try:
            ks = json.load(open(path))["kernels"]
        except Exception:
            continue
        for name, cs in ks.items():
            if key in name and "SQ_ACTIVE_INST_VALU" in cs:
                quad = cs["SQ_ACTIVE_INST_VALU"]["value"]
                out = {"issue_floor_us": round(quad * 4 / 1024 / 2.4e9 * 1e6, 2), "source": os.path.relpath(path, ROOT)}
                if "SQ_INSTS_VALU" in cs:
                    out["valu_wave_instructions"] = cs["SQ_INSTS_VALU"]["value"]
                if "SQ_INSTS_VALU_TRANS_F32" in cs:
                    out["transcendental_wave_instructions"] = cs["SQ_INSTS_VALU_TRANS_F32"]["value"]
                return out
    return None


def train_traffic(keys):
    """HBM bytes per training step of a kernel family from the newest committed PMC summary of the training step
    (profiles/*_train_traffic.json: scripts/pmc_train.sh + scripts/summarize_train_traffic.py, batch 8).  keys: substrings of
    the summary's family names, summed.  (bytes, source) or (None, None)."""
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_train_traffic.json")))
    try:
        tag = open(os.path.join(ROOT, "profiles", "LATEST")).read().strip()
        paths.sort(key=lambda q: os.path.basename(q).startswith(tag + "_"))
    except OSError:
        pass
    for path in reversed(paths):
        try:
            ks = json.load(open(path))["kernels"]
        except Exception:
            continue
        tot = sum(v["traffic_bytes"] for name, v in ks.items() if any(k in name for k in keys))
        if tot > 0:
            return int(tot), os.path.relpath(path, ROOT)
    return None, None


def helix_pair_roofline(step, nrep, batch, act_bytes, img):
    """The Helix-SS2D core at the decoder's top stage (H = img / 4: 96x96 at 384, 192x192 at 768; K = 8, D = 256): fused scan
    launch + merge/out_norm launch, both timed INSIDE eager single-stream forwards of the model by the library's launch
    profiler (one HIP-event pair per launch on the launch stream), restricted by `tramba_profile_min_units` to launches of
    at least 0.9x this shape's kernel-boundary bytes, which only this shape reaches (it is the largest fused-scan and the
    largest merge launch of a forward)."""
    from tramba_amd import hip
    h = img // 4
    b, l, d, k, r = batch, h * h, 256, 8, 8
    rg = hip.ss2d_group_stride(r)
    scan_bytes = b * l * d * act_bytes + b * l * k * rg * 4 + b * k * l * d * act_bytes
    merge_bytes = b * k * l * d * act_bytes + b * l * d * act_bytes
    for which, thr in ((hip.PROF_SCAN_FUSED, 0.9 * scan_bytes), (hip.PROF_MERGE, 0.9 * merge_bytes)):
        hip.profile_min_units(which, thr)
        hip.profile_enable(which, True)
    for _ in range(nrep):
        step()
    ns, ms_s, by_s = hip.profile_read(hip.PROF_SCAN_FUSED)
    nm, ms_m, by_m = hip.profile_read(hip.PROF_MERGE)
    for which in (hip.PROF_SCAN_FUSED, hip.PROF_MERGE):
        hip.profile_enable(which, False)
        hip.profile_min_units(which, 0.0)
    if ns == 0 or nm == 0:
        return None, None
    us_s, us_m = ms_s / ns * 1e3, ms_m / nm * 1e3
    # SURVEY 8(d), fused Helix-SS2D: read x once + low-rank (R + 2N) x_proj rows per direction (fp32 as stored) + write
    # the merged y
    alg = b * l * d * act_bytes + (r + 2) * k * b * l * 4 + b * l * d * act_bytes
    gbs = alg / ((us_s + us_m) * 1e-6) / 1e9
    # the committed PMC / SQ summaries were collected on the BASELINE shape (96x96, batch 4, 16-bit): quoted only there
    base_shape = h == 96 and b == 4 and act_bytes == 2
    pmc_s = pmc_traffic("ss2d_scan_dma_kernel@helix96") if base_shape else None
    pmc_m = pmc_traffic("ss2d_merge_norm_deep_kernel@helix96") if base_shape else None
    traffic = pmc_s["bytes"] + pmc_m["bytes"] if pmc_s and pmc_m else None
    valu = None
    if base_shape:
        fs, fm = sq_valu_floor("ss2d_scan_dma_kernel"), sq_valu_floor("ss2d_merge_norm_deep_kernel")
        if fs and fm:
            floor = fs["issue_floor_us"] + fm["issue_floor_us"]
            valu = {"issue_floor_us": round(floor, 2), "frac": round(floor / (us_s + us_m), 4), "scan": fs, "merge": fm,
                    "what": "time the SIMDs' vector pipes are occupied (SQ_ACTIVE_INST_VALU, 4 cycles per quad-cycle, 1024 "
                            "SIMDs, 2.4 GHz): the ceiling of a launch limited by vector issue alone; frac = floor / measured"}
    shape = f"{h}x{h} (K={k}, D={d}, B={b})"
    pair = {"bound": "valu",
            "kernel": f"Helix-SS2D core at {shape}: ss2d_scan_dma_kernel + ss2d_merge_norm_deep_kernel",
            "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
            "traffic": traffic, "traffic_source": (pmc_s or {}).get("source"), "algorithmic_bytes": alg, "launches": ns,
            "avg_us": round(us_s + us_m, 2), "scan_us": round(us_s, 2), "merge_us": round(us_m, 2), "valu": valu,
            "formula": "achieved / peak / frac: SURVEY 8(d) fused Helix-SS2D bytes -- B*L*D*s (x once) + (R+2N)*K*B*L*4 (x_proj "
                       "rows) + B*L*D*s (merged y), s = activation bytes -- over the summed duration of the two launches, "
                       "against the HBM peak; the K-fold per-direction `ys` between the two launches is NOT counted (it is "
                       "this implementation's intermediate: roofline_kernel_boundary counts it)",
            "note": "bound = valu: both launches are vector / transcendental-issue and latency bound, not HBM bound (DESIGN.md "
                    "section 4: 3 transcendentals + ~17 other vector instructions per element); `valu` prices the pair "
                    "against that ceiling.  Average HIP-event durations inside eager single-stream forwards"}
    kb = {"scan": {"bound": "valu", "kernel": f"ss2d_scan_dma_kernel (Helix {shape})",
                   "achieved": round(by_s / (ms_s * 1e-3) / 1e9, 1),
                   "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(by_s / (ms_s * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                   "algorithmic_bytes": int(by_s / ns), "avg_us": round(us_s, 2), "traffic": (pmc_s or {}).get("bytes"),
                   "traffic_source": (pmc_s or {}).get("source"),
                   "formula": "x once + x_proj rows (padded groups) + ys (B,K,L,D) written"},
          "merge": {"bound": "hbm", "kernel": f"ss2d_merge_norm_deep_kernel (Helix {shape})",
                    "achieved": round(by_m / (ms_m * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(by_m / (ms_m * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_bytes": int(by_m / nm),
                    "avg_us": round(us_m, 2), "traffic": (pmc_m or {}).get("bytes"),
                    "traffic_source": (pmc_m or {}).get("source"),
                    "formula": "ys (B,K,L,D) read + y (B,L,D) written"}}
    return pair, kb


def latency_b1(model, img, graph_ok):
    """Trambav6.py:219-255: batch 1, 50 warm-up + 500 timed forwards, one event pair per forward, mean / std / FPS."""
    import torch
    import tramba_amd as ta
    x = torch.randn(1, 3, img, img, generator=torch.Generator().manual_seed(0)).cuda()
    gf = ta.GraphedForward(model) if graph_ok else None
    run = (lambda: gf(x)) if gf is not None else (lambda: model(x))
    with torch.no_grad():
        for _ in range(50):
            run()
        torch.cuda.synchronize()
        reps = 500
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in evs:
            a.record()
            run()
            b.record()
        torch.cuda.synchronize()
        # the reference's other protocol (test_TSOD.py:71-108, measure_inference_speed): 200 forwards, each between two
        # synchronisations on the host's clock, the first 5 not counted
        pure = 0.0
        for i in range(200):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run()
            torch.cuda.synchronize()
            if i >= 5:
                pure += time.perf_counter() - t0
    ts = torch.tensor([a.elapsed_time(b) for a, b in evs], dtype=torch.float64)
    mean, std = float(ts.mean()), float(ts.std(unbiased=False))
    return {"mean_ms": round(mean, 4), "std_ms": round(std, 4), "fps": round(1000.0 / mean, 2), "batch": 1,
            "warmup": 50, "reps": reps, "launch": "hipGraph replay" if gf is not None else "eager",
            "protocol": "Trambav6.py:219-255 (one event pair per forward)",
            "wall_clock": {"fps": round(195.0 / pure, 2), "ms_per_image": round(pure / 195.0 * 1e3, 4), "warmup": 5, "iters": 200,
                           "protocol": "test_TSOD.py:71-108 (host clock around each synchronised forward)"}}


def graphed_train_leg(model, red, x, y, world, b, steps, timed):
    """tramba_amd.GraphedTrainStep on the same model / batch: an optimisation, never a requirement -- any failure is
    reported in the object instead of raised, and with several ranks the replay only runs if EVERY rank holds a graph."""
    import torch
    import torch.distributed as dist
    import tramba_amd as ta
    from tramba_amd import train
    try:
        gstep = ta.GraphedTrainStep(model, train.get_opt(1e-4, model, capturable=True), reducer=red)
        ok, err = torch.ones(1, device="cuda"), "capture failed on another rank"
        try:
            for _ in range(2):
                gstep(x, y)
            torch.cuda.synchronize()
        except Exception as e:
            ok.zero_()
            err = f"{type(e).__name__}: {e}"[:300]
        if world > 1:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok.item()) <= 0:
            return {"error": err}
        gdt = timed(lambda: gstep(x, y))
        return {"value": round(world * b * steps / gdt, 2), "unit": "img/s", "ms_per_step": round(gdt / steps * 1e3, 2),
                "what": "tramba_amd.GraphedTrainStep: forward + loss + backward (+ bucketed all-reduce) + Adam + "
                        "weight-shadow refresh as one hipGraph replay per step"}
    except Exception as e:
        return {"error": f"{type(e).__name__}: {e}"[:300]}


def bench_train(args, world, rank, dtype, sync_all, publish=None):
    """BASELINE configs[2]/[3]: fwd + bwd (BCE+IoU on 4 outputs) + two-group Adam, batch 8 per GPU, stochastic depth
    on; gradients averaged over RCCL in 32 MB buckets launched from autograd hooks (overlapped with backward) when
    world > 1.  Timed like the forward leg: barrier + synchronize on both sides, max over ranks."""
    import torch
    import torch.distributed as dist
    import tramba_amd as ta
    from tramba_amd import parallel, train
    torch.manual_seed(1026)
    model = ta.bulid_model(use_pretrain=False, img_size=args.img).cuda().train()
    model.compute_dtype = None if dtype == torch.float32 else dtype
    if world > 1:
        parallel.broadcast_parameters(model, src=0)
    bdt = torch.bfloat16 if args.bucket_dtype == "bf16" else None
    red = parallel.GradBucketReducer(model, bucket_dtype=bdt)
    opt = train.get_opt(1e-4, model)
    b = args.train_batch
    x = torch.randn(b, 3, args.img, args.img, generator=torch.Generator().manual_seed(100 + rank)).cuda()
    y = (torch.rand(b, 1, args.img, args.img, generator=torch.Generator().manual_seed(200 + rank)) > 0.7).float().cuda()
    steps, warm = max(20, args.steps), 3      # (at least 20 timed steps whatever --steps is)

    by_rank = {}

    def timed(fn, tag=None):
        sync_all()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        mine = time.perf_counter() - t0                 # this rank's own time, before it waits for the others
        sync_all()
        t = torch.tensor([time.perf_counter() - t0], device="cuda")
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            every = [torch.zeros(1, device="cuda") for _ in range(world)]
            dist.all_gather(every, torch.tensor([mine], device="cuda"))
            mine_all = [float(e.item()) for e in every]
        else:
            mine_all = [mine]
        if tag is not None:                              # host contention between the ranks of one node shows up here
            by_rank[tag] = [round(v / steps * 1e3, 2) for v in mine_all]
        return float(t.item())

    for _ in range(warm):
        train.train_step(model, opt, x, y, reducer=red)
    dt = timed(lambda: train.train_step(model, opt, x, y, reducer=red), "eager")
    # share of the step the collective is not hidden behind backward: the same step with the reducer left out of the
    # exchange (world 1 semantics) cannot be run without desynchronising the replicas, so time the all-reduce of the
    # buckets alone instead and report it beside the step
    comm = None
    if world > 1:
        torch.cuda.synchronize()
        dist.barrier()
        c0 = time.perf_counter()
        for _ in range(3):
            hs = [dist.all_reduce(f, op=dist.ReduceOp.SUM, async_op=True) for f in red.flat]
            for h in hs:
                h.wait()
        torch.cuda.synchronize()
        cdt = (time.perf_counter() - c0) / 3
        nbytes = red.bytes_per_step()
        comm = {"allreduce_alone_ms": round(cdt * 1e3, 3), "share_of_step_if_exposed": round(cdt / (dt / steps), 4),
                "bus_GBps_per_rank": round(2.0 * (world - 1) / world * nbytes / cdt / 1e9, 1),
                "note": "all buckets reduced back to back with nothing else running; inside the step the buckets are "
                        "launched from autograd hooks and overlap the rest of backward"}
        # (the buckets now hold scribble; the next step's prepare() / fill overwrites them)
    # the same step replayed as ONE hipGraph (collectives captured with it when world > 1: opt-in, --graph-dp)
    # the dominant kernel family of the step: forward + input-gradient GEMMs (tramba_linear_cl), HIP-event pairs around
    # every launch of two extra eager steps (not part of the timed region)
    from tramba_amd import hip
    fams = (hip.PROF_GEMM, hip.PROF_WGRAD, hip.PROF_SCAN_BWD, hip.PROF_LAYERNORM, hip.PROF_DW)
    for which in fams:
        hip.profile_enable(which, True)
    for _ in range(2):
        train.train_step(model, opt, x, y, reducer=red)
    ng, msg, flops = hip.profile_read(hip.PROF_GEMM)
    nw, msw, flopw = hip.profile_read(hip.PROF_WGRAD)
    nsb, mssb, bytes_sb = hip.profile_read(hip.PROF_SCAN_BWD)
    nln, msln, bytes_ln = hip.profile_read(hip.PROF_LAYERNORM)
    ndw, msdw, bytes_dw = hip.profile_read(hip.PROF_DW)
    for which in fams:
        hip.profile_enable(which, False)
    # the Helix top-stage launch of the scan backward alone (the largest one: 12 B x B*K*L*D elements)
    hh = args.img // 4
    helix_sb = 12.0 * b * 8 * hh * hh * 256
    hip.profile_min_units(hip.PROF_SCAN_BWD, 0.9 * helix_sb)
    hip.profile_enable(hip.PROF_SCAN_BWD, True)
    train.train_step(model, opt, x, y, reducer=red)
    nsh, mssh, bytes_sh = hip.profile_read(hip.PROF_SCAN_BWD)
    hip.profile_enable(hip.PROF_SCAN_BWD, False)
    hip.profile_min_units(hip.PROF_SCAN_BWD, 0.0)
    tfw = flopw / (msw * 1e-3) / 1e12 if msw > 0 else 0.0
    # the optimizer alone (tramba_adam_step: 28 B per parameter -- g read, p / exp_avg / exp_avg_sq read and written), HIP
    # events on the launch stream around 10 replays of one captured step on the gradients the last training step left
    roof_adam = None
    if isinstance(opt, train.Adam) and world == 1:        # (N = 1 only: a roofline figure, not part of the scaling protocol)
        try:
            nupd = sum(p.numel() for g in opt.param_groups for p in g["params"] if p.grad is not None)
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            opt.step()
            torch.cuda.synchronize()
            adam_graph = torch.cuda.CUDAGraph()      # (launched from Python the 12 launches of a step take longer to issue than to run)
            with torch.cuda.graph(adam_graph):
                opt.step()
            adam_graph.replay()
            torch.cuda.synchronize()
            ev0.record()
            for _ in range(10):
                adam_graph.replay()
            ev1.record()
            torch.cuda.synchronize()
            us_adam = ev0.elapsed_time(ev1) / 10 * 1e3
            del adam_graph
            tr_a, src_a = train_traffic(("adam (",)) if b == 8 else (None, None)
            gadam = 28.0 * nupd / (us_adam * 1e-6) / 1e9
            roof_adam = {"bound": "hbm", "kernel": "adam_kernel (tramba_adam_step), every launch of one optimizer step",
                         "achieved": round(gadam, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gadam / HBM_PEAK_GBS, 4),
                         "traffic": tr_a, "traffic_source": src_a, "traffic_unit": "bytes per step, all launches of the family",
                         "algorithmic_bytes_per_step": int(28 * nupd), "parameters": int(nupd),
                         "us_per_step": round(us_adam, 1),
                         "formula": "28 B per parameter (gradient read; p, exp_avg, exp_avg_sq read and written, fp32) / HIP-event "
                                    "time of 10 replays of one captured optimizer step"}
        except Exception as exc:   # (an auxiliary figure must not cost the leg its result)
            roof_adam = {"error": f"{type(exc).__name__}: {exc}"}
    # PMC traffic of the families (bytes per STEP at batch 8, from the committed summary: only meaningful for this batch)
    tr_w, src_w = train_traffic(("wgrad",)) if b == 8 else (None, None)
    tr_sb, src_sb = train_traffic(("ss2d_scan_bwd",)) if b == 8 else (None, None)
    tr_g, src_g = train_traffic(("linear (",)) if b == 8 else (None, None)
    roof_w = {"bound": "mfma", "kernel": "wgrad_dma_kernel: weight / bias gradients of every 1x1 conv, the per-direction "
                                         "dt_projs_weight contractions, the DCT backward (slab sums recorded for the step's "
                                         "batched reduction)",
              "achieved": round(tfw, 1), "peak": MFMA_PEAK_TFS, "unit": "TFLOP/s", "frac": round(tfw / MFMA_PEAK_TFS, 4),
              "traffic": tr_w, "traffic_source": src_w, "traffic_unit": "bytes per step, all launches of the family",
              "launches": nw // 2, "avg_us": round(msw / max(nw, 1) * 1e3, 2), "ms_per_step": round(msw / 2, 3),
              "note": "2*M*N*K of every tramba_wgrad_cl call / HIP-event time around the call (TN GEMM + its fixed-order slab "
                      "sum), two eager steps"}
    # the two streaming families of the step: every launch of the LayerNorm family (forward, residual-add forms,
    # pixel-shuffle forms, backward) and of the depth-wise stencils (forward, input gradient = the same kernel on flipped taps,
    # weight gradient); bytes = each activation-sized tensor of a call once
    tr_ln, src_ln = train_traffic(("layernorm forward",)) if b == 8 else (None, None)
    tr_dw, src_dw = train_traffic(("depth-wise",)) if b == 8 else (None, None)
    gln = bytes_ln / (msln * 1e-3) / 1e9 if msln > 0 else 0.0
    gdw = bytes_dw / (msdw * 1e-3) / 1e9 if msdw > 0 else 0.0
    roof_ln = {"bound": "hbm", "kernel": "layernorm_* / add_ln_* / layernorm_bwd_* kernels, every launch of a step",
               "achieved": round(gln, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gln / HBM_PEAK_GBS, 4),
               "traffic": tr_ln, "traffic_source": src_ln, "traffic_unit": "bytes per step, all launches of the family",
               "algorithmic_bytes_per_step": int(bytes_ln / 2), "launches": nln // 2,
               "avg_us": round(msln / max(nln, 1) * 1e3, 2), "ms_per_step": round(msln / 2, 3),
               "formula": "bytes of every activation-sized tensor a call reads or writes once (forward: x, y; residual form: + the "
                          "branch and the sum; backward: x, dy, dx (+ the skip gradient and the masked copy)) / HIP-event time "
                          "around the call, two eager steps"}
    roof_dw = {"bound": "hbm", "kernel": "dwconv4_cl_kernel (3x3) / dwconv7_march_kernel (7x7): forward and input gradient; dwconv_wgrad_cl_kernel / dwconv7_wgrad_march_kernel: weight gradient; every launch of a step",
               "achieved": round(gdw, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gdw / HBM_PEAK_GBS, 4),
               "traffic": tr_dw, "traffic_source": src_dw, "traffic_unit": "bytes per step, all launches of the family",
               "algorithmic_bytes_per_step": int(bytes_dw / 2), "launches": ndw // 2,
               "avg_us": round(msdw / max(ndw, 1) * 1e3, 2), "ms_per_step": round(msdw / 2, 3),
               "formula": "x read + y written (+ the pre-activation copy) per stencil launch, x + gy read per weight-gradient launch / "
                          "HIP-event time around the call, two eager steps",
               "note": "the 7x7 launches are bound by the vector ALUs (49 FMAs per output plus the exact GELU of the forward: counters in "
                       "profiles/r04_dw7_counters.txt), not by HBM"}
    gsb = bytes_sb / (mssb * 1e-3) / 1e9 if mssb > 0 else 0.0
    roof_sb = {"bound": "valu", "kernel": "ss2d_scan_bwd_cl_kernel, all 33 launches of a step", "achieved": round(gsb, 1),
               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gsb / HBM_PEAK_GBS, 4), "traffic": tr_sb,
               "traffic_source": src_sb, "traffic_unit": "bytes per step, all launches of the family",
               "algorithmic_bytes_per_step": int(bytes_sb / 2),
               "launches": nsb // 2, "avg_us": round(mssb / max(nsb, 1) * 1e3, 2), "ms_per_step": round(mssb / 2, 3),
               "formula": "SURVEY 8(d) backward of the op it replaces: 12 B per (b,k,d,l) element at 16-bit activations (u, "
                          "delta, dout read; du, ddelta written) / HIP-event time",
               "note": "like the forward scan the kernel is vector-issue bound (recomputes dt_proj, softplus, exp and the "
                       "forward states per tile, then runs the adjoint recurrence), not HBM bound"}
    if nsh > 0 and mssh > 0:
        gsh = bytes_sh / (mssh * 1e-3) / 1e9
        roof_sb["helix_top_stage"] = {"kernel": f"ss2d_scan_bwd_cl_kernel (Helix {hh}x{hh}, K=8, D=256, B={b})",
                                      "achieved": round(gsh, 1), "frac": round(gsh / HBM_PEAK_GBS, 4), "launches": nsh,
                                      "avg_us": round(mssh / nsh * 1e3, 2), "algorithmic_bytes": int(bytes_sh / nsh)}
    tfs = flops / (msg * 1e-3) / 1e12
    roof = {"bound": "mfma", "kernel": "linear_dma_kernel / linear_lean_kernel: forward + input-gradient GEMMs of the step",
            "achieved": round(tfs, 1), "peak": MFMA_PEAK_TFS, "unit": "TFLOP/s", "frac": round(tfs / MFMA_PEAK_TFS, 4),
            "traffic": tr_g, "traffic_source": src_g, "traffic_unit": "bytes per step, all launches of the family",
            "launches": ng // 2, "avg_us": round(msg / ng * 1e3, 2), "ms_per_step": round(msg / 2, 3),
            "note": "2*M*N*K of every tramba_linear_cl launch / HIP-event time, two eager steps; the weight-gradient GEMMs "
                    "(wgrad_tn_kernel) and the scan backward are the next two families (profiles/*_train_kernel_stats.csv)"}
    eager = {"value": round(world * b * steps / dt, 2), "unit": "img/s", "ms_per_step": round(dt / steps * 1e3, 2)}

    def assemble(graphed):
        # `value` = the faster of the product's two ways to run the step (train.train_step launch by launch, whose rate
        # follows the host CPU of the box; tramba_amd.GraphedTrainStep = fit(graph=True)); both are reported
        best, launch = eager, "eager"
        if isinstance(graphed, dict) and graphed.get("value", 0.0) > eager["value"]:
            best, launch = graphed, "hipGraph replay (tramba_amd.GraphedTrainStep)"
        return {"metric": f"images/sec fwd+bwd+Adam Tramba-V {args.img}x{args.img}", "value": best["value"],
                "scaling_value": best["value"], "unit": "img/s", "steps": steps, "ms_per_step": best["ms_per_step"],
                "batch_per_gpu": b, "global_batch": b * world, "launch": launch,
                "grad_bytes_per_step": red.bytes_per_step() if world > 1 else 0,
                "grad_bucket_bytes": red.bytes_per_step(), "grad_buckets": len(red.buckets),
                "grad_bucket_dtype": args.bucket_dtype, "allreduce": comm,
                "parallelism": f"dp{world}" + ((", RCCL all-reduce (ncclAvg)" if args.backend == "nccl" else ", gloo all-reduce")
                                                + " of gradient buckets from autograd hooks" if world > 1 else ", no collective"),
                "stochastic_depth": "on (0.6 enc / 0.2 dec)", "dtype": args.dtype + " activations, fp32 master weights",
                "roofline": roof, "roofline_wgrad": roof_w, "roofline_scan_bwd": roof_sb, "roofline_adam": roof_adam,
                "roofline_layernorm": roof_ln, "roofline_dw": roof_dw, "eager": eager,
                "graphed": graphed,
                "step_ms_by_rank": dict(by_rank)}

    if world > 1 and (args.no_graph_dp or args.backend != "nccl"):
        graphed = {"skipped": "--no-graph-dp" if args.no_graph_dp else
                   "gloo collectives stage through the host and cannot be captured into a hipGraph (RCCL's can)"}
    else:
        if publish is not None:   # the eager result is in the line BEFORE the capture of collectives is attempted
            publish(assemble({"pending": "graphed data-parallel leg had not finished when the line was printed"}))
        graphed = graphed_train_leg(model, red, x, y, world, b, steps, lambda fn: timed(fn, "graphed"))
    return assemble(graphed)

def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; running with {world} ranks", file=sys.stderr)
    ndev = torch.cuda.device_count()
    assert ndev > 0, "bench.py needs an MI355X"
    dev_index = local % ndev
    if world > 1:
        torch.cuda.set_device(dev_index)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group("gloo")
    from tramba_amd import hip
    from tramba_amd import models as _models
    if args.no_overlap:
        _models.OVERLAP_BRANCHES = False
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
                out_static = step()  # noqa: F841
        except Exception as e:  # graph capture is an optimisation, never a requirement
            if rank == 0:
                print(f"bench.py: hipGraph capture unavailable ({type(e).__name__}: {e}); running eager", file=sys.stderr)
            graph = None
            torch.cuda.synchronize()
    run = (lambda: graph.replay()) if graph is not None else step
    graph_used = graph is not None

    for _ in range(args.warmup):
        run()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    sync_all()
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        run()
        marks[i + 1].record()
    sync_all()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], device="cuda")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    value = world * args.batch * args.steps / dt
    per = torch.tensor([marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)], dtype=torch.float64)
    std_ms = float(per.std(unbiased=False)) if args.steps > 1 else 0.0

    # ---- rank 0: latency protocol, rooflines (eager single-stream passes with HIP events around the library's
    #      launches; events cannot live inside a captured graph), CPU baseline
    lat = roof = roof_kb = roof_all = roof_b = roof_g = cpu = None
    extras_error = None
    if rank == 0:
        try:   # the headline number above stands on its own: a failure in here is reported in the line, not raised
            if not args.no_latency:
                lat = latency_b1(model, args.img, graph is not None)
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
            gbs = nbytes / (ms * 1e-3) / 1e9
            roof_all = {"bound": "hbm", "kernel": "ss2d_scan_dma_kernel + ss2d_scan_cl_kernel + ss2d_seg_kernel, all shapes", "achieved": round(gbs, 1),
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None,
                        "launches": n, "avg_us": round(ms / n * 1e3, 2), "ms_per_step": round(ms / nrep, 3),
                        "note": "kernel-boundary bytes (x once + x_proj rows + ys) / HIP-event time, summed over every "
                                "fused-scan launch of a step (single-stream eager pass, one event pair per launch)"}
            roof, roof_kb = helix_pair_roofline(step, nrep, args.batch, 4 if dtype == torch.float32 else 2, args.img)
            _models.OVERLAP_BRANCHES = overlap_was
            tfs = flops / (msg * 1e-3) / 1e12
            roof_g = {"bound": "mfma", "kernel": "linear_pc_kernel / linear_ws_kernel / linear_dma_kernel / linear_lean_kernel (1x1-conv projections)",
                      "achieved": round(tfs, 1), "peak": MFMA_PEAK_TFS, "unit": "TFLOP/s", "frac": round(tfs / MFMA_PEAK_TFS, 4),
                      "traffic": None, "launches": ng, "avg_us": round(msg / ng * 1e3, 2), "ms_per_step": round(msg / nrep, 3),
                      "note": "2*M*N*K of every tramba_linear_cl launch of a step / their HIP-event time; these GEMMs are "
                              "small (M = 576..36864, K <= 4096): LDS- and latency-bound, far from the dense MFMA peak"}
            roof_b = boundary_scan_roofline(dtype, args.batch, args.img)
        except Exception as e:
            extras_error = f"{type(e).__name__}: {e}"[:400]
            print(f"bench.py: latency / roofline section failed: {extras_error}", file=sys.stderr)
            _models.OVERLAP_BRANCHES = True if not args.no_overlap else False
    line = None
    if rank == 0:
        line = {
            "metric": f"images/sec fwd Tramba-V {args.img}x{args.img}", "value": round(value, 2), "unit": "img/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "std_ms": round(std_ms, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"Tramba-V (VMamba-B encoder) {args.img}x{args.img} {args.dtype} inference, "
                                   f"batch {args.batch} per GPU, random-init weights (seed 1026), randn images",
                       "global_batch": args.batch * world,
                       "parallelism": f"dp{world}: the forward leg (`value`) is {world} independent replica(s), no collective; "
                                      f"the data-parallel figure is train.scaling_value (gradient all-reduce over RCCL)",
                       "launch": ("hipGraph replay" if graph_used else "eager") +
                                 ("" if args.no_overlap else ", decoder guide branches on a side stream")},
            # (filled in last -- the CPU passes run after the GPU legs -- but kept HERE, in front of the long objects, with its
            #  all_cores / single_thread members: `cores` = 1 beside `physical_cores` = 128 reads correctly only with both)
            "cpu_baseline": None,
            "roofline": roof, "latency_b1": lat, "roofline_kernel_boundary": roof_kb, "roofline_fused_scan_all": roof_all,
            "roofline_boundary": roof_b, "roofline_gemm": roof_g, "train": None,
        }
        if extras_error is not None:
            line["extras_error"] = extras_error
    # From here on the ranks exchange gradients.  A watchdog thread bounds the leg: if it (or the final barrier) has not
    # finished in --train-timeout seconds, rank 0 prints the line it already holds and every rank leaves -- a collective that
    # never completes must not cost the run its forward measurement.
    done = threading.Event()

    def watchdog():
        if done.wait(args.train_timeout):
            return
        if rank == 0:
            msg = (f"training leg not finished after {args.train_timeout:.0f} s (rank 0 gave up; N = {world}, backend "
                   f"{args.backend})")
            if isinstance(line.get("train"), dict) and "eager" in line["train"]:
                line["train"]["graphed"] = {"error": msg}      # the eager data-parallel result was already published
            else:
                line["train"] = {"error": msg}
            print(json.dumps(line), flush=True)
        os._exit(0 if rank == 0 else 3)

    threading.Thread(target=watchdog, daemon=True).start()

    # A rank that ABORTS (rather than hangs) makes the launcher send the others SIGTERM: rank 0 then prints the line it holds -- the
    # forward result, and the eager data-parallel result if the leg had got that far -- before it leaves (VERDICT r3 "next" 3c).
    def on_term(signum, _frame):
        if rank == 0 and line is not None and not done.is_set():
            msg = f"terminated by signal {signum} during the training leg (another rank failed; N = {world})"
            if isinstance(line.get("train"), dict) and "eager" in line["train"]:
                line["train"]["graphed"] = {"error": msg}
            else:
                line["train"] = {"error": msg}
            try:
                sys.stdout.write(json.dumps(line) + "\n")
                sys.stdout.flush()
            except Exception:
                pass
        os._exit(0 if rank == 0 else 3)

    if world > 1:
        import signal
        signal.signal(signal.SIGTERM, on_term)
    train_obj = None
    if not args.no_train:
        ok = torch.ones(1, device="cuda")
        try:
            train_obj = bench_train(args, world, rank, dtype, sync_all,
                                    publish=(lambda obj: line.__setitem__("train", obj)) if rank == 0 else None)
        except Exception as e:   # reported, not raised: the forward result stands on its own
            train_obj = {"error": f"{type(e).__name__}: {e}"[:400]}
            ok.zero_()
            print(f"bench.py: rank {rank}: training leg failed: {train_obj['error']}", file=sys.stderr)
        if world > 1 and float(ok.item()) > 0:
            dist.barrier()
    if world > 1 and (train_obj is None or "error" not in train_obj):
        dist.destroy_process_group()
    done.set()
    # the CPU baseline runs LAST: its OpenMP team keeps spinning on the host cores for a while and slows the eager launch
    # thread of whatever follows (r02: 60.5 instead of 55 ms per eager training step right after it)
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(args.img, args.cpu_baseline_full)
            except Exception as e:   # reported, not raised
                line["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"[:400]}
        line["train"] = train_obj
        print(json.dumps(line), flush=True)
    if world > 1 and train_obj is not None and "error" in train_obj:
        os._exit(0 if rank == 0 else 3)   # peers may be parked in a collective this rank never joins: leave without teardown

if __name__ == "__main__":
    main()
