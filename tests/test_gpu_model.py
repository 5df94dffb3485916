"""GPU parity tests, module/model level: the product nn.Modules (HIP kernels underneath) against
the golden vectors captured from the reference and against the CPU oracle."""
import numpy as np
import pytest
import torch

import synth
from oracle import model as om
from oracle import ops as oo

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _load_synth(module, dtype=torch.float32):
    sd = module.state_dict()
    new = synth.synth_state_dict(((k, v.shape) for k, v in sd.items()), keep=synth.CONST_KEYS)
    for k in sd:
        if k not in new:
            new[k] = sd[k]
    module.load_state_dict(new, strict=True)
    return module.to(DEV).eval()


def _blocks():
    import tramba_amd as ta
    return {
        "ss2d_raster": (lambda: ta.SS2D(d_model=16, d_state=1, ssm_ratio=2.0, dt_rank="auto", d_conv=3, conv_bias=False,
                                        channel_first=True), (2, 16, 12, 12)),
        "vssblock": (lambda: ta.VSSBlock(hidden_dim=16, drop_path=0.0, channel_first=True), (2, 16, 12, 12)),
        "freqblock": (lambda: ta.FreqBlockv6(dim=16, input_resolution=(12, 12)), (2, 16, 12, 12)),
        "helixblock": (lambda: ta.MultiScaleDecoderBlock(hidden_dim=16, drop_path=0.0, channel_first=True), (2, 16, 12, 12)),
        "freqblock24": (lambda: ta.FreqBlockv6(dim=32, input_resolution=(24, 24)), (1, 32, 24, 24)),
        "helixblock24": (lambda: ta.MultiScaleDecoderBlock(hidden_dim=32, drop_path=0.0, channel_first=True), (1, 32, 24, 24)),
        "patchexpand": (lambda: ta.PatchExpand(dim=32, dim_scale=2), (2, 32, 6, 6)),
        "finalexpand": (lambda: ta.FinalPatchExpand_X4(dim=8, dim_scale=4), (2, 8, 6, 6)),
        "freqexpand": (lambda: ta.FreqExpand2D(dim=8), (2, 8, 6, 6)),
    }


TAGS = ["ss2d_raster", "vssblock", "freqblock", "helixblock", "freqblock24", "helixblock24", "patchexpand",
        "finalexpand", "freqexpand"]


@pytest.mark.parametrize("tag", TAGS)
def test_block_inference_matches_reference_golden(golden, golden_meta, tag):
    ctor, shape = _blocks()[tag]
    m = _load_synth(ctor())
    assert [(k, list(v.shape)) for k, v in m.state_dict().items()] == [(e[0], e[1]) for e in golden_meta["G4_manifest"][tag]]
    x = synth.synth_input("g4_" + tag, shape).to(DEV)
    with torch.no_grad():
        y = m(x)
    assert tuple(y.shape) == tuple(golden[f"g4_{tag}_y"].shape)
    np.testing.assert_allclose(y.float().cpu().numpy(), golden[f"g4_{tag}_y"], rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("tag", TAGS)
def test_block_training_path_matches_reference_golden(golden, golden_meta, tag):
    """grad-enabled path: scan plugins + SelectiveScanOflex fwd/bwd HIP kernels under autograd."""
    ctor, shape = _blocks()[tag]
    m = _load_synth(ctor())
    x = synth.synth_input("g4_" + tag, shape).to(DEV).requires_grad_()
    y = m(x)
    np.testing.assert_allclose(y.detach().float().cpu().numpy(), golden[f"g4_{tag}_y"], rtol=1e-3, atol=1e-4)
    gy = synth.synth_input("g4_gy_" + tag, tuple(y.shape)).to(DEV)
    params = list(m.named_parameters())
    grads = torch.autograd.grad(y, [x] + [p for _, p in params], gy)
    np.testing.assert_allclose(grads[0].cpu().numpy(), golden[f"g4_{tag}_dx"], rtol=5e-3, atol=2e-4)
    ref = golden_meta["G4_param_grads"][tag]
    for (n, _), g in zip(params, grads[1:]):
        s, a = ref[n]
        assert abs(float(g.double().abs().sum()) - a) <= 5e-3 * a + 2e-4, n
        assert abs(float(g.double().sum()) - s) <= 5e-3 * a + 2e-4, n


def test_custom_scan_plugin_goes_through_generic_path():
    """A user-supplied scan class (the reference's plugin API) must work without the fused path."""
    import tramba_amd as ta

    class MyScan(torch.autograd.Function):  # plain raster, written by a "user"
        @staticmethod
        def forward(ctx, x):
            return ta.CrossScan.apply(x)

    class MyMerge(torch.autograd.Function):
        @staticmethod
        def forward(ctx, ys):
            return ta.CrossMerge.apply(ys)

    torch.manual_seed(0)
    a = ta.SS2D(d_model=16, d_state=1, channel_first=True).to(DEV).eval()
    b = ta.SS2D(d_model=16, d_state=1, channel_first=True, scan=MyScan, merge=MyMerge).to(DEV).eval()
    b.load_state_dict(a.state_dict())
    x = torch.randn(1, 16, 12, 12, device=DEV)
    with torch.no_grad():
        assert not b._fused_ok(x) and a._fused_ok(x)
        np.testing.assert_allclose(a(x).cpu().numpy(), b(x).cpu().numpy(), rtol=2e-4, atol=2e-5)


def test_cpu_tensor_is_rejected():
    import tramba_amd as ta
    m = ta.LayerNorm2d(8)
    with pytest.raises(RuntimeError):
        m(torch.randn(1, 8, 4, 4))


def _full_state(manifest):
    sd = synth.synth_state_dict(manifest, keep=synth.DCT_KEYS)
    for name, shape in manifest:
        if name not in sd:
            sd[name] = oo.dct_matrix(shape[0])
    return sd


@pytest.fixture(scope="module")
def tramba_v():
    import tramba_amd as ta
    m = ta.bulid_model(deep_supervision=True, use_pretrain=False, img_size=384, dims=128, depths=[2, 2, 2, 2])
    return _load_synth(m)


def test_tramba_v_fp32_matches_reference_golden(golden, golden_meta, tramba_v):
    x = synth.synth_input("g5_v", (1, 3, 384, 384)).to(DEV)
    with torch.no_grad():
        feats = tramba_v.vssm_encoder(x)
        outs = tramba_v(x)
    for i, f in enumerate(feats[1:]):
        pooled = torch.nn.functional.avg_pool2d(f.float(), f.shape[-1] // 6).cpu().numpy()
        np.testing.assert_allclose(pooled, golden[f"g5_v_enc{i}_pool"], rtol=2e-3, atol=5e-4)
    assert [tuple(o.shape) for o in outs] == [(1, 1, 24, 24), (1, 1, 48, 48), (1, 1, 96, 96), (1, 1, 384, 384)]
    for i in range(3):
        np.testing.assert_allclose(outs[i].cpu().numpy(), golden[f"g5_v_out{i}"], rtol=2e-3, atol=1e-3)
    np.testing.assert_allclose(outs[3][:, :, 160:224, 160:224].cpu().numpy(), golden["g5_v_out3_crop"], rtol=2e-3, atol=1e-3)
    pred = torch.sigmoid(outs[3])[0, 0].cpu().numpy()
    gt = (synth.synth_input("g5_gt", (384, 384)) > 0.5).numpy()
    assert round(oo.mae_metric(pred, gt), 4) == round(golden_meta["G5_tramba_v_mae"], 4)


# north_star: "MAE metric unchanged to 4 d.p." -- i.e. |dMAE| < 5e-5 -- holds in the 16-bit modes too.  Measured on MI355X (printed by
# the tests below, r04; profiles/r04_parity_measurements.txt): 384x384 bf16 1.0e-5, fp16 3.0e-6; 768x768 fp16 1.1e-5, fp32 0.
MAE_TOL = {torch.bfloat16: 5e-5, torch.float16: 5e-5}


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_tramba_v_low_precision_keeps_mae(golden_meta, dtype):
    """bf16/fp16 inference keeps the MAE of the fp32 reference to 4 decimal places (north_star's contract; |dMAE| < 5e-5).  (A weak
    check by itself -- the golden map is at chance level, MAE 0.516 -- the element-wise test below is the one that sees wrong
    logits.)"""
    import tramba_amd as ta
    m = ta.bulid_model(deep_supervision=True, use_pretrain=False, img_size=384)
    m = ta.prepare_inference(_load_synth(m), dtype)
    x = synth.synth_input("g5_v", (1, 3, 384, 384)).to(DEV)
    with torch.no_grad():
        outs = m(x)
    pred = torch.sigmoid(outs[3])[0, 0].cpu().numpy()
    gt = (synth.synth_input("g5_gt", (384, 384)) > 0.5).numpy()
    mae = oo.mae_metric(pred, gt)
    print(f"measured |dMAE| at 384x384 {dtype}: {abs(mae - golden_meta['G5_tramba_v_mae']):.2e} (MAE {mae:.6f})")
    assert abs(mae - golden_meta["G5_tramba_v_mae"]) < MAE_TOL[dtype], (mae, golden_meta["G5_tramba_v_mae"])


def _err_stats(got, want):
    """max-abs and RMS error relative to the RMS of the reference map, and the fraction of sigmoid > 0.5 decisions that flip"""
    d = got.double() - want.double()
    ref = float(want.double().square().mean().sqrt())
    return float(d.abs().max()) / ref, float(d.square().mean().sqrt()) / ref, float(((got > 0) != (want > 0)).double().mean())


@pytest.fixture(scope="module")
def batch4_oracle():
    """Tramba-V 384x384 at the BENCHMARKED batch (4): closed-form weights, image 0 = the G5 golden input, fp32 CPU oracle
    outputs of all four maps (the oracle itself is pinned to the reference's forward by tests/test_oracle.py)."""
    import tramba_amd as ta
    m = _load_synth(ta.bulid_model(deep_supervision=True, use_pretrain=False, img_size=384))
    x = torch.cat([synth.synth_input("g5_v", (1, 3, 384, 384))] +
                  [synth.synth_input(f"g5_v_b{i}", (1, 3, 384, 384)) for i in range(1, 4)], 0)
    with torch.no_grad():
        want = om.tramba_v({k: v.detach().cpu() for k, v in m.state_dict().items()}, x)
    del m
    return x, want


# measured on MI355X (scripts/measure_lowp_parity.py, profiles/r02_lowp_parity.json), relative to the RMS of each
# reference map: (max-abs, RMS, decision flips) -- asserted with ~1.6x headroom
LOWP_TOL = {torch.bfloat16: (0.16, 0.04, 0.02), torch.float16: (0.025, 0.006, 0.008), torch.float32: (2e-4, 2e-5, 1e-4)}


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_tramba_v_batch4_elementwise_against_fp32_oracle(golden, batch4_oracle, dtype):
    """BASELINE configs[1] is bf16 at batch 4: every logit of all four output maps against the fp32 reference forward on
    the same inputs -- max-abs and RMS error and the fraction of pixels whose saliency decision flips (the
    MAE of a chance-level map cannot see wrong logits)."""
    import tramba_amd as ta
    x, want = batch4_oracle
    m = ta.prepare_inference(_load_synth(ta.bulid_model(deep_supervision=True, use_pretrain=False, img_size=384)), dtype)
    with torch.no_grad():
        got = [o.float().cpu() for o in m(x.to(DEV))]
    tmax, trms, tflip = LOWP_TOL[dtype]
    for i, (g, w) in enumerate(zip(got, want)):
        emax, erms, flips = _err_stats(g, w)
        assert emax <= tmax and erms <= trms and flips <= tflip, (str(dtype), i, emax, erms, flips)
    # image 0 is the golden input of the reference run itself
    for i in range(3):
        emax, erms, _ = _err_stats(got[i][:1], torch.from_numpy(golden[f"g5_v_out{i}"]))
        assert emax <= tmax and erms <= trms, (str(dtype), i, emax, erms)
    emax, erms, _ = _err_stats(got[3][:1, :, 160:224, 160:224], torch.from_numpy(golden["g5_v_out3_crop"]))
    assert emax <= 1.5 * tmax and erms <= 1.5 * trms, (str(dtype), emax, erms)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("tag", TAGS)
def test_block_inference_low_precision_against_reference_golden(golden, tag, dtype):
    """the nine G4 blocks with 16-bit activations / GEMM weights against the reference's fp32 outputs"""
    import tramba_amd as ta
    ctor, shape = _blocks()[tag]
    m = _load_synth(ctor())
    for mod in m.modules():            # the prepare_inference policy on a bare block
        if isinstance(mod, ta.Linear2d):
            mod.weight.data = mod.weight.data.to(dtype)
        elif isinstance(mod, ta.SS2D):
            mod.x_proj_weight.data = mod.x_proj_weight.data.to(dtype)
    x = synth.synth_input("g4_" + tag, shape).to(DEV, dtype)
    with torch.no_grad():
        y = m(x).float().cpu()
    emax, erms, _ = _err_stats(y, torch.from_numpy(golden[f"g4_{tag}_y"]))
    tmax, trms = (0.08, 0.012) if dtype == torch.bfloat16 else (0.01, 0.0015)
    assert emax <= tmax and erms <= trms, (tag, str(dtype), emax, erms)


def test_tramba_r_384_matches_reference_golden(golden):
    """Tramba-Res at the reference's own resolution against the reference forward (G5: Trambav6_enc.py:131-159)"""
    import tramba_amd as ta
    m = _load_synth(ta.bulid_model_enc("Tramba-R-TSOD", img_size=384))
    x = synth.synth_input("g5_r", (1, 3, 384, 384)).to(DEV)
    with torch.no_grad():
        outs = m(x)
    assert [tuple(o.shape) for o in outs] == [(1, 1, 48, 48), (1, 1, 96, 96), (1, 1, 384, 384)]
    for i in range(2):
        np.testing.assert_allclose(outs[i].cpu().numpy(), golden[f"g5_r_out{i}"], rtol=2e-3, atol=1e-3)
    np.testing.assert_allclose(outs[2][:, :, 160:224, 160:224].cpu().numpy(), golden["g5_r_out2_crop"], rtol=2e-3, atol=1e-3)
    np.testing.assert_allclose(torch.nn.functional.avg_pool2d(outs[2], 8).cpu().numpy(), golden["g5_r_out2_pool8"],
                               rtol=2e-3, atol=1e-3)


def test_loss_on_device_matches_oracle_and_reference_constants(golden_meta):
    """train.py:76-85 / utils/loss.py:6-11 evaluated on the device: the G7 known answers of the reference's own
    functions and the oracle's restatement on the same deep-supervision pyramid"""
    from tramba_amd import train
    pred = synth.synth_input("g7_pred", (2, 1, 24, 24), scale=2.0)
    mask = (synth.synth_input("g7_mask", (2, 1, 24, 24)) > 0.3).float()
    pd, md = pred.to(DEV), mask.to(DEV)
    assert abs(float(train.iou_loss(pd, md)) - golden_meta["G7"]["iou_loss"]) < 1e-6
    assert abs(float(torch.nn.functional.binary_cross_entropy_with_logits(pd, md)) - golden_meta["G7"]["bce"]) < 1e-6
    outs = [torch.nn.functional.avg_pool2d(pred, 4), torch.nn.functional.avg_pool2d(pred, 2), pred * 0.5, pred]
    want = float(oo.tramba_loss(outs, mask))
    got = float(train.tramba_loss([o.to(DEV) for o in outs], md))
    assert abs(got - want) < 1e-5 * max(1.0, abs(want)), (got, want)
    got16 = float(train.tramba_loss([o.to(DEV, torch.bfloat16) for o in outs], md))     # model outputs arrive as bf16
    assert abs(got16 - want) < 2e-2 * abs(want)


def _rccl_one_rank_worker(rank, port, captures):
    """Body of test_reducer_on_a_one_rank_rccl_group, run in a CHILD process (torch.multiprocessing spawn): r03 recorded one
    `Fatal Python error: Aborted` inside hipGraph capture_end of this path (1 full-suite run in 6, never standalone); an abort
    here now fails one test instead of taking the pytest process -- and every later test -- with it.  `captures` = how many
    times the captured step is rebuilt and replayed (scripts/loop_capture_rccl.py runs it 30 times for profiles/)."""
    import faulthandler
    import os
    import sys
    import torch.distributed as dist
    from tramba_amd import parallel, train
    import tramba_amd as ta
    faulthandler.enable(file=sys.stderr, all_threads=True)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        torch.manual_seed(0)
        m = ta.MultiScaleDecoderBlock(hidden_dim=32, drop_path=0.0, channel_first=True).to(DEV).train()
        x = torch.randn(2, 32, 24, 24, device=DEV)
        m(x).square().mean().backward()
        want = {n: p.grad.clone() for n, p in m.named_parameters()}
        # (two backward passes differ in the last bits: index_add_ / atomic sums of the scan backward)
        for bdt, tol in ((None, 1e-4), (torch.bfloat16, 1e-2)):
            red = parallel.GradBucketReducer(m, bucket_mb=0.01, bucket_dtype=bdt)
            assert red.world == 1 and red._native_avg and len(red.buckets) > 1
            red.world = 2                      # force the collective path (a one-rank group averages over one rank)
            red.prepare()
            m(x).square().mean().backward()
            red.finish()
            torch.cuda.synchronize()
            for n, p in m.named_parameters():
                assert p.grad is not None and p.grad.dtype == p.dtype, n
                assert float((p.grad - want[n]).abs().max()) <= tol * float(want[n].abs().max()) + 1e-12, n
            red.remove_hooks()
        # the same collective path CAPTURED into the training step's hipGraph (GraphedTrainStep with a reducer): the
        # all-reduces issued from the autograd hooks become graph nodes and replay
        class Tiny(torch.nn.Module):
            def __init__(self):
                super().__init__()
                torch.manual_seed(1)
                self.encoder = ta.VSSBlock(hidden_dim=64, drop_path=0.0, channel_first=True).to(DEV).train()
                self.compute_dtype = None

            def forward(self, z):
                return [self.encoder(z).mean(dim=1, keepdim=True)]

        xs = torch.randn(2, 64, 24, 24, device=DEV)
        ys = (torch.rand(2, 1, 24, 24, device=DEV) > 0.5).float()
        ref = Tiny()
        opt = train.get_opt(1e-3, ref)
        want = [float(train.train_step(ref, opt, xs, ys)) for _ in range(3)]
        assert want[2] < want[0]
        for it in range(captures):
            model = Tiny()
            red = parallel.GradBucketReducer(model, bucket_mb=0.05)
            red.world = 2
            step = ta.GraphedTrainStep(model, train.get_opt(1e-3, model, capturable=True), reducer=red)
            got = [float(step(xs, ys)) for _ in range(3)]
            assert np.allclose(got, want, rtol=2e-3), (it, got, want)
            red.remove_hooks()
            del step, red, model
            if captures > 1:
                print(f"capture {it + 1}/{captures} ok", flush=True)
    finally:
        dist.destroy_process_group()


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_reducer_on_a_one_rank_rccl_group():
    """GradBucketReducer through torch.distributed's nccl backend (= RCCL) with a single rank: the collective code path
    (ncclAvg all-reduce of the flat buckets launched from autograd hooks, wait, .grad views) runs on the device and leaves
    exactly the gradients of a plain backward; the same with bf16 buckets up to their rounding; then the same path CAPTURED
    into a training step's hipGraph.  In a fresh child process (see _rccl_one_rank_worker): an abort inside the runtime fails
    this test only."""
    import torch.multiprocessing as mp
    mp.spawn(_rccl_one_rank_worker, args=(_free_port(), 1), nprocs=1, join=True)


class _TinyDP(torch.nn.Module):
    """a VSSBlock behind the interface train.train_step expects (a list of logit maps; the name holds "encoder")"""

    def __init__(self):
        super().__init__()
        import tramba_amd as ta
        torch.manual_seed(11)
        self.encoder = ta.VSSBlock(hidden_dim=64, drop_path=0.0, channel_first=True)
        self.compute_dtype = None

    def forward(self, z):
        return [self.encoder(z).mean(dim=1, keepdim=True)]


def _dp_data(n):
    g = torch.Generator().manual_seed(5)
    return torch.randn(n, 64, 24, 24, generator=g), (torch.rand(n, 1, 24, 24, generator=g) > 0.5).float()


def _dp_gpu_worker(rank, world, port, out_dir, steps):
    import os
    import torch.distributed as dist
    from tramba_amd import parallel, train
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)                       # both ranks share the one card of the test box
        model = _TinyDP().to("cuda").train()
        parallel.broadcast_parameters(model, src=0)
        red = parallel.GradBucketReducer(model, bucket_mb=0.05)
        opt = train.get_opt(1e-2, model)
        x, y = _dp_data(4 * world)
        xs, ys = x[rank::world].to("cuda"), y[rank::world].to("cuda")
        for _ in range(steps):
            train.train_step(model, opt, xs, ys, reducer=red)
        torch.cuda.synchronize()
        if rank == 0:
            torch.save({"sd": {k: v.cpu() for k, v in model.state_dict().items()}, "nbuckets": len(red.buckets),
                        "bytes": red.bytes_per_step()}, os.path.join(out_dir, "r0.pt"))
    finally:
        dist.destroy_process_group()


def test_dp2_on_the_device_matches_a_single_process(tmp_path):
    """Two data-parallel ranks with the model ON THE GPU (both on the one card, gloo carrying the buckets): hooks, bucket
    fill, all-reduce, .grad views and the fused Adam leave the weights a single process gets from the mean of the two
    shard losses -- the N > 1 code path of bench.py's training leg, with the device kernels in it."""
    import socket
    import torch.multiprocessing as mp
    from tramba_amd import train
    steps, world = 3, 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_dp_gpu_worker, args=(world, port, str(tmp_path), steps), nprocs=world, join=True)
    got = torch.load(tmp_path / "r0.pt")
    assert got["nbuckets"] > 1
    model = _TinyDP().to(DEV).train()
    opt = train.get_opt(1e-2, model)
    x, y = _dp_data(4 * world)
    x, y = x.to(DEV), y.to(DEV)
    for _ in range(steps):
        opt.zero_grad(set_to_none=True)
        loss = sum(train.tramba_loss(model(x[r::world]), y[r::world]) for r in range(world)) / world
        loss.backward()
        opt.step()
    assert got["bytes"] == sum(p.numel() * 4 for p in model.parameters())
    moved = 0.0
    ref0 = _TinyDP().state_dict()
    for k, v in model.state_dict().items():
        np.testing.assert_allclose(got["sd"][k].numpy(), v.cpu().numpy(), rtol=2e-3, atol=2e-4, err_msg=k)
        moved = max(moved, float((v.cpu() - ref0[k]).abs().max()))
    assert moved > 1e-3                                   # the steps did move the weights


def test_tramba_v_batch4_consistent(tramba_v):
    """images are independent units: a batch of 4 equals four batches of 1."""
    x = torch.randn(4, 3, 384, 384, generator=torch.Generator().manual_seed(0)).to(DEV)
    with torch.no_grad():
        full = tramba_v(x)[-1]
        one = tramba_v(x[2:3])[-1]
    np.testing.assert_allclose(full[2:3].cpu().numpy(), one.cpu().numpy(), rtol=1e-3, atol=1e-3)


def test_tramba_r_256_against_oracle():
    """BASELINE config 1: Tramba-Res 256x256 batch 1 (feature sizes 64/32/16 are OFF the reference's
    table -- the oracle defines the behaviour, tests/test_oracle.py pins its generators)."""
    import tramba_amd as ta
    m = _load_synth(ta.bulid_model_enc("Tramba-R-TSOD", img_size=256))
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    x = synth.synth_input("c1", (1, 3, 256, 256))
    with torch.no_grad():
        want = om.tramba_r(sd, x)
        got = m(x.to(DEV))
    assert [tuple(o.shape) for o in got] == [(1, 1, 32, 32), (1, 1, 64, 64), (1, 1, 256, 256)]
    for g, w in zip(got, want):
        np.testing.assert_allclose(g.cpu().numpy(), w.numpy(), rtol=2e-3, atol=1e-3)


@pytest.mark.parametrize("tag,name", [("s", "Tramba-S-TSOD"), ("p", "Tramba-P-TSOD")])
def test_tramba_s_p_match_reference_golden(golden_enc, tag, name):
    """Tramba-S (Swin-B) / Tramba-P (PVTv2-b4), SURVEY 8f-4: fp32 against the reference's forward; the encoder's
    reference-style NCHW outputs (deepest first) and the token-major features the decoder consumes agree; bf16
    inference keeps the MAE."""
    import tramba_amd as ta
    m = _load_synth(ta.bulid_model_enc(name))
    x = synth.synth_input(f"g5_{tag}", (1, 3, 384, 384)).to(DEV)
    with torch.no_grad():
        feats = m.encoder(x)
        outs = m(x)
    for i, f in enumerate(feats):
        pooled = torch.nn.functional.avg_pool2d(f.float(), f.shape[-1] // 6).cpu().numpy()
        np.testing.assert_allclose(pooled, golden_enc[f"g5_{tag}_enc{i}_pool"], rtol=2e-3, atol=5e-4)
    assert [tuple(o.shape) for o in outs] == [(1, 1, 24, 24), (1, 1, 48, 48), (1, 1, 96, 96), (1, 1, 384, 384)]
    for i in range(3):
        np.testing.assert_allclose(outs[i].cpu().numpy(), golden_enc[f"g5_{tag}_out{i}"], rtol=2e-3, atol=1e-3)
    np.testing.assert_allclose(outs[3][:, :, 160:224, 160:224].cpu().numpy(), golden_enc[f"g5_{tag}_out3_crop"], rtol=2e-3, atol=1e-3)
    np.testing.assert_allclose(torch.nn.functional.avg_pool2d(outs[3], 8).cpu().numpy(), golden_enc[f"g5_{tag}_out3_pool8"],
                               rtol=2e-3, atol=1e-3)
    gt = (synth.synth_input("g5_gt", (384, 384)) > 0.5).numpy()
    mae32 = oo.mae_metric(torch.sigmoid(outs[3])[0, 0].cpu().numpy(), gt)
    m16 = ta.prepare_inference(m, torch.bfloat16)
    with torch.no_grad():
        o16 = m16(x)
    assert abs(oo.mae_metric(torch.sigmoid(o16[3])[0, 0].cpu().numpy(), gt) - mae32) < 1e-3
    # training-mode graph (stock torch ops) gives the same numbers as the HIP inference path, DropPath off
    m32 = _load_synth(ta.bulid_model_enc(name))
    for mod in m32.modules():
        if isinstance(mod, ta.DropPath):
            mod.drop_prob = 0.0
    m32.train()
    xg = x.clone().requires_grad_(True)
    og = m32(xg)
    np.testing.assert_allclose(og[2].detach().cpu().numpy(), golden_enc[f"g5_{tag}_out2"], rtol=5e-3, atol=2e-3)
    og[3].mean().backward()
    assert xg.grad is not None and torch.isfinite(xg.grad).all() and float(xg.grad.abs().sum()) > 0


def test_level0_extension_signature_matches_reference_call_sites():
    """selective_scan_cuda_oflex.fwd/.bwd exactly as called at csms6s.py:910 and :920-922."""
    from oracle import selective_scan as oss
    from tramba_amd.ops import selective_scan_cuda_oflex as ext
    g = torch.Generator().manual_seed(0)
    nb, k, dper, l = 2, 4, 8, 576
    kd = k * dper
    u = torch.randn(nb, kd, l, generator=g)
    delta = 0.5 * torch.randn(nb, kd, l, generator=g) - 1
    A = -(torch.rand(kd, 1, generator=g) + 0.2)
    B, C = torch.randn(nb, k, 1, l, generator=g), torch.randn(nb, k, 1, l, generator=g)
    D, bias = torch.ones(kd), 0.1 * torch.randn(kd, generator=g)
    dout = torch.randn(nb, kd, l, generator=g)
    dev = lambda *t: [x.to(DEV) for x in t]
    out, x, *rest = ext.fwd(*dev(u, delta, A, B, C, D, bias), True, 1, True)
    want = oss.selective_scan_fwd(u, delta, A, B, C, D, bias, True)
    np.testing.assert_allclose(out.cpu().double().numpy(), want.numpy(), rtol=2e-4, atol=2e-4)
    du, ddelta, dA, dB, dC, dD, dbias, *rest = ext.bwd(*dev(u, delta, A, B, C, D, bias, dout), x, True, 1)
    wg = oss.selective_scan_bwd(u, delta, A, B, C, D, bias, dout, True)
    for got, w in zip((du, ddelta, dA, dB, dC, dD, dbias), wg):
        w = w.numpy()
        assert np.abs(got.cpu().double().numpy() - w).max() <= 3e-4 * max(1.0, np.abs(w).max())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_tramba_v_train_step(dtype):
    """BASELINE config 3 path: fwd + bwd (HIP scan backward) + two-group Adam on Tramba-V, 2 steps."""
    import tramba_amd as ta
    from tramba_amd import parallel, train
    torch.manual_seed(0)
    m = ta.bulid_model(use_pretrain=False, img_size=384).to(DEV).train()
    for mod in m.modules():  # deterministic check: no stochastic depth (never compare RNG streams)
        if isinstance(mod, ta.DropPath):
            mod.drop_prob = 0.0
    m.compute_dtype = None if dtype == torch.float32 else dtype
    opt = train.get_opt(1e-4, m)
    red = parallel.GradBucketReducer(m)  # world 1: exercises the bucket plumbing on the GPU
    x = torch.randn(2, 3, 384, 384, generator=torch.Generator().manual_seed(0)).to(DEV)
    y = (torch.rand(2, 1, 384, 384, generator=torch.Generator().manual_seed(1)) > 0.7).float().to(DEV)
    # (1) full-model gradient check: the loss change along -g matches the first-order prediction
    red.prepare()
    loss0 = train.tramba_loss(m(x), y)
    loss0.backward()
    red.finish()
    gnorm = float(torch.sqrt(sum((p.grad.float() ** 2).sum() for p in m.parameters())))
    assert np.isfinite(gnorm) and gnorm > 0
    step = 1e-3
    with torch.no_grad():
        fused = float(train.tramba_loss(m(x), y))  # autograd off -> fused all-HIP inference path
        for p in m.parameters():
            p.add_(p.grad, alpha=-step / gnorm)
        loss1 = float(train.tramba_loss(m(x), y))
    rel = 2e-3 if dtype == torch.float32 else 3e-2
    assert abs(fused - float(loss0)) <= rel * abs(float(loss0)), (fused, float(loss0))  # both SS2D paths agree
    drop, pred = float(loss0) - loss1, step * gnorm
    assert (0.7 if dtype == torch.float32 else 0.3) * pred < drop < 1.3 * pred, (float(loss0), loss1, pred)
    # (2) the full step (zero-grad via the reducer, backward with bucket hooks, two-group Adam) runs
    l2 = float(train.train_step(m, opt, x, y, reducer=red))
    l3 = float(train.train_step(m, opt, x, y, reducer=red))
    assert np.isfinite([l2, l3]).all()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())


@pytest.fixture(scope="module")
def c5_oracle():
    """BASELINE configs[4] at ITS batch (2): the two 768x768 inputs and the fp32 CPU oracle's four maps for both"""
    import tramba_amd as ta
    m = _load_synth(ta.bulid_model(deep_supervision=True, use_pretrain=False, img_size=768))
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    del m
    x = torch.cat([synth.synth_input("c5", (1, 3, 768, 768)), synth.synth_input("c5b", (1, 3, 768, 768))])
    with torch.no_grad():
        want = om.tramba_v(sd, x)
    return x, want


# fp16 at 768x768 (L up to 36 864), relative to the RMS of each reference map: (max-abs, RMS, decision flips); measured on
# MI355X (printed by the test, r04), asserted with ~1.6x headroom like LOWP_TOL at 384x384
# measured (r04): max-abs 0.006-0.013, RMS 0.0014-0.0033, flips 0.0002-0.0005
LOWP_TOL_768 = {torch.float16: (0.022, 0.0055, 0.001), torch.float32: (5e-4, 5e-5, 2e-4)}


def test_tramba_v_768_fp16_long_sequence_against_oracle(c5_oracle):
    """BASELINE config 5: 768x768 fp16, feature sizes 192/96/48/24 (L up to 36 864; 192 is OFF the
    reference's tables: window 16, dilation 4 by the documented rule).  Compared with the fp32 oracle."""
    import tramba_amd as ta
    m = _load_synth(ta.bulid_model(deep_supervision=True, use_pretrain=False, img_size=768))
    x, want2 = c5_oracle
    x, want = x[:1], [w[:1] for w in want2]
    with torch.no_grad():
        got32 = m(x.to(DEV))
        m16 = ta.prepare_inference(m, torch.float16)
        got16 = m16(x.to(DEV))
    assert [tuple(o.shape) for o in got32] == [(1, 1, 48, 48), (1, 1, 96, 96), (1, 1, 192, 192), (1, 1, 768, 768)]
    for g, w in zip(got32, want):
        np.testing.assert_allclose(g.cpu().numpy(), w.numpy(), rtol=5e-3, atol=2e-3)
    gt = (synth.synth_input("c5_gt", (768, 768)) > 0.5).numpy()
    mae = lambda o: oo.mae_metric(torch.sigmoid(o[-1])[0, 0].float().cpu().numpy(), gt)
    print(f"measured |dMAE| at 768x768 fp16: {abs(mae(got16) - mae(want)):.2e}; fp32: {abs(mae(got32) - mae(want)):.2e}")
    assert abs(mae(got16) - mae(want)) < MAE_TOL[torch.float16]
    assert abs(mae(got32) - mae(want)) < 5e-5             # fp32: unchanged to 4 d.p.


def test_tramba_v_train_step_at_the_baseline_batch():
    """BASELINE config 3 at ITS batch size (8 per GPU, bf16): the loss and the gradients of the batch equal the mean of its
    two halves run on their own (images are independent; the scan kernels pick other schedules at batch 8 than at 4), and
    the full optimisation step runs."""
    import tramba_amd as ta
    from tramba_amd import train
    torch.manual_seed(0)
    m = ta.bulid_model(use_pretrain=False, img_size=384).to(DEV).train()
    for mod in m.modules():
        if isinstance(mod, ta.DropPath):
            mod.drop_prob = 0.0
    m.compute_dtype = torch.bfloat16
    x = torch.randn(8, 3, 384, 384, generator=torch.Generator().manual_seed(0)).to(DEV)
    y = (torch.rand(8, 1, 384, 384, generator=torch.Generator().manual_seed(1)) > 0.7).float().to(DEV)
    probes = [p for n, p in m.named_parameters() if p.numel() >= 65536][::12] + [p for n, p in m.named_parameters() if "A_logs" in n][:3]

    def run(xs, ys):
        for p in m.parameters():
            p.grad = None
        loss = train.tramba_loss(m(xs), ys)
        loss.backward()
        return float(loss.detach()), [p.grad.detach().double().clone() for p in probes]

    l8, g8 = run(x, y)
    la, ga = run(x[:4], y[:4])
    lb, gb = run(x[4:], y[4:])
    assert abs(l8 - 0.5 * (la + lb)) <= 2e-3 * abs(l8), (l8, la, lb)
    for g, a, b in zip(g8, ga, gb):
        want = 0.5 * (a + b)
        cos = float((g * want).sum() / (g.norm() * want.norm() + 1e-30))
        assert cos > 0.995 and 0.97 < float(g.norm() / want.norm()) < 1.03, (cos, float(g.norm() / want.norm()))
    opt = train.get_opt(1e-4, m)
    losses = [float(train.train_step(m, opt, x, y)) for _ in range(2)]
    assert np.isfinite(losses).all()       # (Adam's first steps on random-init weights need not lower the loss)


def test_tramba_v_768_fp16_at_the_baseline_batch(c5_oracle):
    """BASELINE config 5 at ITS batch size (2) in ITS dtype (fp16): every logit of all four maps of both images against the
    fp32 CPU oracle on the same inputs -- max-abs and RMS error relative to the RMS of each reference map and the fraction of
    flipped saliency decisions, as test_tramba_v_batch4_elementwise_against_fp32_oracle does at 384x384 (VERDICT r3 'missing'
    #4) -- and every image of the batch equals the same image run alone to fp16 rounding (the kernels may pick another
    schedule for the larger launch)."""
    import tramba_amd as ta
    m = ta.prepare_inference(_load_synth(ta.bulid_model(deep_supervision=True, use_pretrain=False, img_size=768)), torch.float16)
    x, want = c5_oracle
    x = x.to(DEV)
    with torch.no_grad():
        both = m(x)
        alone = [m(x[i:i + 1]) for i in range(2)]
    tmax, trms, tflip = LOWP_TOL_768[torch.float16]
    for j, (o, w) in enumerate(zip(both, want)):
        emax, erms, flips = _err_stats(o.float().cpu(), w)
        print(f"768x768 fp16 batch 2, map {j}: max-abs {emax:.4f} rms {erms:.5f} flips {flips:.5f} (relative to the map's RMS)")
        assert emax <= tmax and erms <= trms and flips <= tflip, (j, emax, erms, flips)
    for j, o in enumerate(both):
        for i in range(2):
            ref = alone[i][j][0].float()
            scale = float(ref.abs().max())
            assert float((o[i].float() - ref).abs().max()) <= 2e-2 * scale + 1e-3, (j, i)
            assert float(((o[i] > 0) != (alone[i][j][0] > 0)).float().mean()) < 2e-3   # saliency decisions


def test_training_guide_branch_overlap_is_bitwise_neutral():
    """r04: under autograd the decoder's guide branches run on a side stream too (models.OVERLAP_TRAINING), forward and -- through
    the autograd engine's stream rule -- backward, their deferred partial sums flushed on that stream: losses, every parameter
    gradient and the weights after three optimisation steps at the BASELINE batch (8) are BIT-identical to the single-stream
    step, eagerly and as one hipGraph.  (The first version of this test failed one run in four: wgrad_dma_kernel consumed
    transposed LDS reads behind counted lgkmcnt waits, which only hold while no other kernel shares the CU.)"""
    import tramba_amd as ta
    from tramba_amd import models, train

    def run(overlap, graphed):
        models.OVERLAP_TRAINING = overlap
        try:
            torch.manual_seed(7)
            m = ta.bulid_model(use_pretrain=False, img_size=384).to(DEV).train()
            for mod in m.modules():
                if isinstance(mod, ta.DropPath):
                    mod.drop_prob = 0.0
            m.compute_dtype = torch.bfloat16
            x = torch.randn(8, 3, 384, 384, generator=torch.Generator().manual_seed(0)).to(DEV)
            y = (torch.rand(8, 1, 384, 384, generator=torch.Generator().manual_seed(1)) > 0.7).float().to(DEV)
            opt = train.get_opt(1e-4, m, capturable=graphed)
            step = ta.GraphedTrainStep(m, opt) if graphed else (lambda a, b: train.train_step(m, opt, a, b))
            losses = [float(step(x, y)) for _ in range(3)]
            torch.cuda.synchronize()
            grads = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
            return losses, grads, {n: p.detach().clone() for n, p in m.named_parameters()}
        finally:
            models.OVERLAP_TRAINING = True

    for graphed in (False, True):
        l0, g0, w0 = run(False, graphed)
        for _ in range(2):                       # twice: the failure this test was written for came and went
            l1, g1, w1 = run(True, graphed)
            assert l0 == l1, (graphed, l0, l1)
            assert g0.keys() == g1.keys()
            bad = [n for n in g0 if not torch.equal(g0[n], g1[n])] + [n for n in w0 if not torch.equal(w0[n], w1[n])]
            assert not bad, (graphed, len(bad), bad[:6])


def test_training_weight_shadows_follow_the_optimizer():
    """train_step refreshes the bf16 shadows of the fp32 Linear2d weights with one fused cast; a shadow is used only while
    its parameter is unchanged (version counter), so an out-of-band update falls back to a fresh cast."""
    import tramba_amd as ta
    from tramba_amd import modules as M, train
    torch.manual_seed(0)
    m = ta.bulid_model(use_pretrain=False, img_size=384).to(DEV).train()
    m.compute_dtype = torch.bfloat16
    opt = train.get_opt(1e-4, m)
    x = torch.randn(1, 3, 384, 384, generator=torch.Generator().manual_seed(0)).to(DEV)
    y = (torch.rand(1, 1, 384, 384, generator=torch.Generator().manual_seed(1)) > 0.7).float().to(DEV)
    train.train_step(m, opt, x, y)
    lin = next(mod for mod in m.modules() if isinstance(mod, ta.Linear2d))
    sh = M._lowp(lin.weight, torch.bfloat16)
    assert sh.data_ptr() == M._lowp_shadow[id(lin.weight)][0].data_ptr()           # the shadow, not a new tensor
    assert torch.equal(sh, lin.weight.detach().to(torch.bfloat16))
    # every Linear2d weight: shadow == cast, transposed shadow == its transpose (one multi-tensor launch wrote them all)
    nlin = 0
    for mod in m.modules():
        if isinstance(mod, ta.Linear2d):
            w16 = mod.weight.detach().to(torch.bfloat16)
            assert torch.equal(M._lowp(mod.weight, torch.bfloat16), w16)
            wt = M._lowp_t(mod.weight, torch.bfloat16)
            assert wt is not None and wt.shape == (w16.shape[1], w16.shape[0]) and torch.equal(wt, w16.t())
            nlin += 1
    assert nlin > 100
    # every fused SS2D core: the padded (K*RG, D) x_proj weight the scan kernels read, and its transpose
    nss = 0
    for mod in m.modules():
        if isinstance(mod, ta.SS2D):
            ent = M._lowp_shadow[id(mod.x_proj_weight)]
            want = ta.hip.pad_x_proj_weight(mod.x_proj_weight.detach().to(torch.bfloat16)).contiguous()
            assert ent[1] == mod.x_proj_weight._version and torch.equal(ent[0], want) and torch.equal(ent[3], want.t())
            nss += 1
    assert nss > 20
    with torch.no_grad():
        lin.weight.mul_(0.5)                                                        # out-of-band update: version moves on
    fresh = M._lowp(lin.weight, torch.bfloat16)
    assert fresh.data_ptr() != sh.data_ptr() and torch.equal(fresh, lin.weight.detach().to(torch.bfloat16))
    assert M._lowp_t(lin.weight, torch.bfloat16) is None                           # stale: the caller transposes
    l2 = float(train.train_step(m, opt, x, y))                                      # and training goes on
    assert np.isfinite(l2)


def test_bucket_fill_waits_for_gradients_finished_on_another_stream():
    """With the guide branches on a side stream under autograd (models.OVERLAP_TRAINING) the gradients of ONE bucket are finished
    on different streams, and the hook that completes the bucket runs on only one of them: the fill (and the collective behind
    it) must wait for the others.  Emulated exactly: parameter a's gradient is written on a side stream BEHIND a device-side
    sleep, its hook fires there; parameter b's hook completes the bucket on the main stream right after."""
    from tramba_amd import parallel

    class Two(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = torch.nn.Parameter(torch.zeros(4096, device=DEV))
            self.b = torch.nn.Parameter(torch.zeros(4096, device=DEV))

    m = Two()
    red = parallel.GradBucketReducer(m, bucket_dtype=torch.bfloat16)      # (world 1, bf16 buckets: the fill path without a collective)
    assert len(red.buckets) == 1
    side = torch.cuda.Stream()
    for trial in range(3):
        red.prepare()
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            g = torch.empty(4096, device=DEV)
            torch.cuda._sleep(200_000_000)          # ~0.1 s on the device: the fill below is enqueued long before this ends
            g.fill_(3.0 + trial)
            m.a.grad = g
            red._on_grad(m.a)
        m.b.grad = torch.full((4096,), -1.0, device=DEV)
        red._on_grad(m.b)                           # completes the bucket on the main stream
        red.finish()
        torch.cuda.synchronize()
        assert float(m.a.grad.float().min()) == 3.0 + trial and float(m.a.grad.float().max()) == 3.0 + trial
        assert float(m.b.grad.float().min()) == -1.0
