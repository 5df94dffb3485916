"""CPU tests of the data-parallel step (world_size 2, gloo) and of the train-step host logic.
The reducer is device-agnostic, so it is exercised here with a small stock-torch model; the HIP
model itself needs a GPU (tests/test_gpu_*.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from oracle import ops as oo


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _toy():
    torch.manual_seed(7)
    return nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.GELU(), nn.Conv2d(8, 8, 3, padding=1), nn.GELU(),
                         nn.Conv2d(8, 1, 1))


class _Wrap(nn.Module):
    """gives the toy net the model contract of the path: a list of logits, 'encoder' in some names"""

    def __init__(self):
        super().__init__()
        self.encoder = _toy()
        self.decoder = nn.Conv2d(1, 1, 1)

    def forward(self, x):
        y = self.encoder(x)
        return [nn.functional.avg_pool2d(y, 2), self.decoder(y)]


def _data(n):
    g = torch.Generator().manual_seed(11)
    x = torch.randn(n, 3, 16, 16, generator=g)
    m = (torch.rand(n, 1, 16, 16, generator=g) > 0.7).float()
    return x, m


class _WrapSpare(_Wrap):
    """the same net plus a head that only rank 0 uses (`use_spare`), and one nobody uses"""

    def __init__(self):
        super().__init__()
        self.spare = nn.Conv2d(1, 1, 1)
        self.never = nn.Conv2d(1, 1, 1)
        self.use_spare = False

    def forward(self, x):
        outs = super().forward(x)
        if self.use_spare:
            outs[1] = outs[1] + 0.5 * self.spare(outs[1])
        return outs


def _worker(rank, world, port, out_dir, steps, bucket_mb, bucket_dtype=None, spare=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tramba_amd import parallel, train
    model = _WrapSpare() if spare else _Wrap()
    if spare:
        model.use_spare = rank == 0
    if rank != 0:  # replicas must not depend on identical local seeds
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)
    parallel.broadcast_parameters(model, src=0)
    red = parallel.GradBucketReducer(model, bucket_mb=bucket_mb, bucket_dtype=bucket_dtype, find_unused=spare)
    opt = train.get_opt(1e-2, model)
    x, m = _data(4 * world)
    xs, ms = x[rank::world], m[rank::world]
    losses = []
    for _ in range(steps):
        losses.append(float(train.train_step(model, opt, xs, ms, reducer=red)))
    extra = {}
    if spare:   # `spare` got a gradient on rank 0 only: every rank holds the mean; `never` stayed None everywhere
        extra = {"spare_grad": model.spare.weight.grad.clone(), "never_is_none": model.never.weight.grad is None}
    torch.save({"sd": model.state_dict(), "losses": losses, "nbuckets": len(red.buckets),
                "bytes": red.bytes_per_step(), **extra}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("bucket_mb", [32.0, 0.0005])
def test_dp2_matches_single_process(tmp_path, bucket_mb):
    """2 ranks x 4 images with bucketed all-reduce == 1 process with the mean of the two shard losses."""
    from tramba_amd import train
    steps, world = 3, 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), steps, bucket_mb), nprocs=world, join=True)
    got = torch.load(os.path.join(tmp_path, "r0.pt"))
    assert got["nbuckets"] == (1 if bucket_mb > 1 else got["nbuckets"]) and got["nbuckets"] >= 1
    if bucket_mb < 1:
        assert got["nbuckets"] > 1  # the multi-bucket / overlap path really ran
    model = _Wrap()
    opt = train.get_opt(1e-2, model)
    x, m = _data(4 * world)
    for _ in range(steps):
        opt.zero_grad(set_to_none=True)
        loss = sum(train.tramba_loss(model(x[r::world]), m[r::world]) for r in range(world)) / world
        loss.backward()
        opt.step()
    for k, v in model.state_dict().items():
        np.testing.assert_allclose(got["sd"][k].numpy(), v.numpy(), rtol=2e-5, atol=2e-6, err_msg=k)
    assert got["bytes"] == sum(p.numel() * 4 for p in model.parameters())


def test_dp2_bf16_buckets_follow_the_fp32_run(tmp_path):
    """bucket_dtype=bf16: half the bytes on the links, gradients rounded once -- the trajectory stays close"""
    steps, world = 3, 2
    os.makedirs(tmp_path / "a")
    os.makedirs(tmp_path / "b")
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path / "a"), steps, 32.0), nprocs=world, join=True)
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path / "b"), steps, 32.0, torch.bfloat16), nprocs=world, join=True)
    a, b = torch.load(tmp_path / "a" / "r0.pt"), torch.load(tmp_path / "b" / "r0.pt")
    assert b["bytes"] * 2 == a["bytes"]
    np.testing.assert_allclose(b["losses"], a["losses"], rtol=2e-2)
    for k, v in a["sd"].items():
        np.testing.assert_allclose(b["sd"][k].numpy(), v.numpy(), rtol=0.1, atol=2e-2, err_msg=k)


def test_dp2_parameter_unused_on_one_rank_or_on_all(tmp_path):
    """find_unused=True: a head used by rank 0 only receives the mean (half of rank 0's gradient) on both ranks and the
    replicas stay identical; a head no rank used keeps .grad = None, so Adam never touches it."""
    steps, world = 2, 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), steps, 32.0, None, True), nprocs=world, join=True)
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert r0["never_is_none"] and r1["never_is_none"]
    assert torch.equal(r0["spare_grad"], r1["spare_grad"]) and float(r0["spare_grad"].abs().sum()) > 0
    for k in r0["sd"]:
        assert torch.equal(r0["sd"][k], r1["sd"][k]), k
    fresh = _WrapSpare()
    assert torch.equal(r0["sd"]["never.weight"], fresh.state_dict()["never.weight"])
    assert not torch.equal(r0["sd"]["spare.weight"], fresh.state_dict()["spare.weight"])


def test_reducer_single_process_semantics():
    """world size 1: an unused parameter keeps .grad = None (Adam skips it, like the reference's plain loop); freezing
    the encoder rebuilds the buckets at the next step; remove_hooks() detaches the reducer from the model."""
    from tramba_amd import parallel, train
    model = _WrapSpare()
    red = parallel.GradBucketReducer(model, bucket_mb=0.0005)
    opt = train.get_opt(1e-2, model)
    x, m = _data(4)
    before = {k: v.clone() for k, v in model.state_dict().items()}
    train.train_step(model, opt, x, m, reducer=red)
    assert model.spare.weight.grad is None and model.never.bias.grad is None
    assert torch.equal(model.spare.weight, before["spare.weight"])
    assert not torch.equal(model.decoder.weight, before["decoder.weight"])
    nb = len(red.buckets)
    for p in model.encoder.parameters():
        p.requires_grad = False
    frozen = {k: v.clone() for k, v in model.encoder.state_dict().items()}
    train.train_step(model, opt, x, m, reducer=red)
    train.train_step(model, opt, x, m, reducer=red)
    assert len(red.buckets) < nb and all(p.grad is None for p in model.encoder.parameters())
    for k, v in model.encoder.state_dict().items():
        assert torch.equal(v, frozen[k]), k                         # no stale momentum keeps moving frozen weights
    red.remove_hooks()
    opt.zero_grad(set_to_none=True)
    train.tramba_loss(model(x), m).backward()                       # a plain step: the reducer no longer sees it
    assert model.decoder.weight.grad is not None and not red._armed


@pytest.mark.parametrize("bucket_dtype", [None, torch.bfloat16])
def test_bucket_views_start_on_16_byte_boundaries(bucket_dtype):
    """the `.grad` views of a bucket are what the optimizer's kernel reads (its 16-byte path needs aligned tensors):
    parameters of odd sizes are padded apart, the flags sit behind the padded payload, the reported bytes are the gradients'"""
    from tramba_amd import parallel
    model = nn.Sequential(nn.Linear(3, 5), nn.Linear(5, 7, bias=False), nn.Conv2d(1, 3, 3), nn.Linear(1, 1))
    red = parallel.GradBucketReducer(model, bucket_mb=0.0001, bucket_dtype=bucket_dtype)
    es = 4 if bucket_dtype is None else 2
    seen = 0
    for flat, bucket, views, flags in zip(red.flat, red.buckets, red._views, red._flags):
        assert flat.data_ptr() % 16 == 0 and flags.numel() == len(bucket)
        ends = []
        for p, v in zip(bucket, views):
            assert v.data_ptr() % 16 == 0 and v.shape == p.shape
            ends.append((v.data_ptr() - flat.data_ptr()) // es + p.numel())
            seen += p.numel()
        assert (flags.data_ptr() - flat.data_ptr()) // es >= max(ends)          # no view overlaps the flags
    assert seen == sum(p.numel() for p in model.parameters())
    assert red.bytes_per_step() == seen * es
    red.remove_hooks()


def test_loss_matches_oracle_and_reference_known_answer(golden_meta):
    import synth
    from tramba_amd import train
    pred = synth.synth_input("g7_pred", (2, 1, 24, 24), scale=2.0)
    mask = (synth.synth_input("g7_mask", (2, 1, 24, 24)) > 0.3).float()
    assert abs(float(train.iou_loss(pred, mask)) - golden_meta["G7"]["iou_loss"]) < 1e-6
    small = torch.nn.functional.avg_pool2d(pred, 2)
    a = train.tramba_loss([small, pred], mask)
    b = oo.tramba_loss([small, pred], mask)
    assert abs(float(a) - float(b)) < 1e-6


def test_optimizer_groups_and_lr_schedule():
    from tramba_amd import train
    model = _Wrap()
    opt = train.get_opt(1e-4, model)
    n_enc = sum(1 for n, _ in model.named_parameters() if "encoder" in n)
    assert len(opt.param_groups[0]["params"]) == n_enc and opt.param_groups[0]["lr"] == pytest.approx(1e-5)
    assert opt.param_groups[1]["lr"] == pytest.approx(1e-4)
    assert train.adjust_learning_rate(opt, 3, [60], 1e-4, [0.2]) == pytest.approx(1e-4)
    assert train.adjust_learning_rate(opt, 60, [60], 1e-4, [0.2]) == pytest.approx(2e-5)
    assert opt.param_groups[0]["lr"] == pytest.approx(2e-6)


def test_compat_shims_import():
    import sys
    compat = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tramba_amd", "compat")
    sys.path.insert(0, compat)
    try:
        for mod in ("get_model", "Trambav6", "Trambav6_enc", "Models", "Models.vmamba", "Models.SS2D.csms6s",
                    "Models.freq_mamba", "Models.DCT_2D", "Models.modules", "Models.mamba_init", "utils", "utils.loss",
                    "utils.lr"):
            sys.modules.pop(mod, None)
        import get_model
        import Trambav6
        import Trambav6_enc
        from Models.SS2D.csms6s import CrossScan_Line, SelectiveScanOflex, selective_scan_cuda_oflex  # noqa: F401
        from Models.vmamba import SS2D, VSSMEncoder  # noqa: F401
        from Models.freq_mamba import FreqBlockv6  # noqa: F401
        from utils.loss import iou_loss  # noqa: F401
        assert callable(get_model.build) and callable(Trambav6.bulid_model) and callable(Trambav6_enc.bulid_model)
    finally:
        sys.path.remove(compat)
        for mod in ("get_model", "Trambav6", "Trambav6_enc", "Models", "Models.vmamba", "Models.SS2D",
                    "Models.SS2D.csms6s", "Models.freq_mamba", "Models.DCT_2D", "Models.modules", "Models.mamba_init",
                    "utils", "utils.loss", "utils.lr"):
            sys.modules.pop(mod, None)


def test_bf16_buckets_with_a_bucket_nobody_filled():
    """world size 1, bf16 buckets small enough that the unused head owns a bucket: finish() must leave its gradients None
    and must not call a multi-tensor copy on empty lists (round-2 advisor finding)."""
    from tramba_amd import parallel, train
    model = _WrapSpare()
    red = parallel.GradBucketReducer(model, bucket_mb=1e-5, bucket_dtype=torch.bfloat16)
    assert len(red.buckets) > 4
    opt = train.get_opt(1e-2, model)
    x, m = _data(4)
    for _ in range(2):
        train.train_step(model, opt, x, m, reducer=red)
    assert model.never.weight.grad is None and model.spare.weight.grad is None
    g = model.decoder.weight.grad
    assert g is not None and g.dtype == torch.float32 and torch.isfinite(g).all()
