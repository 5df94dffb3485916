"""GPU parity tests of the two ends of the optimisation step that are not network layers (reference train.py:74-89):
the deep-supervision loss (train.py:76-85, utils/loss.py:6-11) against the fp64 oracle and autograd through it, and Adam
(train.py:266-280) against torch.optim.Adam's own arithmetic in fp64 on the CPU and its device forms."""
import copy

import numpy as np
import pytest
import torch

import synth
from oracle import ops as oo

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rel_l2(got, want):
    want = want.detach().double().cpu()
    return float((got.detach().double().cpu() - want).norm() / want.norm().clamp_min(1e-30))


LOSS_CASES = {
    # name: (batch, channels, label size, output sizes)
    "pyramid": (3, 1, (40, 40), [(5, 5), (10, 10), (20, 20), (40, 40)]),
    "ragged": (2, 1, (37, 45), [(7, 9), (12, 15), (37, 45)]),          # non-integer ratios, rectangular
    "one_pixel": (2, 1, (16, 16), [(1, 1), (16, 16)]),                 # a 1x1 map spread over the label
    "planes": (2, 3, (24, 20), [(6, 5), (24, 20)]),                    # several channels: mean over (batch, channel)
    "training": (8, 1, (384, 384), [(24, 24), (48, 48), (96, 96), (384, 384)]),   # the benchmarked step's shapes
}


@pytest.mark.parametrize("name", list(LOSS_CASES))
@pytest.mark.parametrize("soft", [False, True])
def test_loss_value_and_gradients_against_the_oracle(name, soft):
    from tramba_amd import train
    b, c, (hh, ww), sizes = LOSS_CASES[name]
    outs = [synth.synth_input(f"loss_{name}_{i}", (b, c, h, w), scale=3.0) for i, (h, w) in enumerate(sizes)]
    lab = synth.synth_input(f"loss_{name}_y", (b, c, hh, ww))
    lab = torch.sigmoid(3 * lab) if soft else (lab > 0.2).float()
    o64 = [o.double().requires_grad_() for o in outs]
    want = oo.tramba_loss(o64, lab.double())
    want.backward()
    od = [o.to(DEV).requires_grad_() for o in outs]
    got = train.tramba_loss(od, lab.to(DEV))
    assert got.dtype == torch.float32 and got.dim() == 0
    assert abs(float(got.detach()) - float(want.detach())) < 2e-6 * max(1.0, abs(float(want.detach()))), (got, want)
    (got * 1.0).backward()
    for i, (o, ref) in enumerate(zip(od, o64)):
        assert o.grad.shape == o.shape
        assert _rel_l2(o.grad, ref.grad) < 2e-5, (name, i, _rel_l2(o.grad, ref.grad))


def test_loss_weights_and_incoming_gradient():
    """loss_weights scale each output's term; a scaled loss scales every gradient (the incoming gradient is a device scalar)"""
    from tramba_amd import train
    outs = [synth.synth_input(f"lossw_{i}", (2, 1, s, s), scale=2.0) for i, s in enumerate((8, 16, 32))]
    lab = (synth.synth_input("lossw_y", (2, 1, 32, 32)) > 0).float()
    wts = (0.25, 2.0, 1.5)
    o64 = [o.double().requires_grad_() for o in outs]
    want = sum(w * oo.tramba_loss([o], lab.double()) for w, o in zip(wts, o64))
    (want * 0.37).backward()
    od = [o.to(DEV).requires_grad_() for o in outs]
    got = train.tramba_loss(od, lab.to(DEV), loss_weights=wts)
    assert abs(float(got.detach()) - float(want.detach())) < 2e-6 * abs(float(want.detach()))
    (got * 0.37).backward()
    for o, ref in zip(od, o64):
        assert _rel_l2(o.grad, ref.grad) < 2e-5


def test_loss_is_reproducible_and_ignores_outputs_that_need_no_gradient():
    from tramba_amd import train
    outs = [synth.synth_input(f"lossr_{i}", (4, 1, s, s), scale=2.0).to(DEV) for i, s in enumerate((12, 48, 96))]
    lab = (synth.synth_input("lossr_y", (4, 1, 96, 96)) > 0).float().to(DEV)
    runs = []
    for _ in range(3):
        od = [o.clone().requires_grad_(i != 1) for i, o in enumerate(outs)]
        loss = train.tramba_loss(od, lab)
        loss.backward()
        assert od[1].grad is None
        runs.append((loss.clone(), od[0].grad.clone(), od[2].grad.clone()))
    for r in runs[1:]:
        assert all(torch.equal(a, b) for a, b in zip(r, runs[0]))       # fixed summation order, no atomics
    with torch.no_grad():
        assert torch.equal(train.tramba_loss(outs, lab), runs[0][0])


def test_loss_rejects_what_the_kernels_cannot_take():
    from tramba_amd import hip
    lab = torch.zeros(1, 1, 8, 8, device=DEV)
    with pytest.raises(hip.TrambaHipError):
        hip.sod_loss([torch.zeros(1, 1, 16, 16, device=DEV)], lab)       # larger than the label
    with pytest.raises(hip.TrambaHipError):
        hip.sod_loss([torch.zeros(1, 1, 8, 8, device=DEV, dtype=torch.bfloat16)], lab)
    with pytest.raises(hip.TrambaHipError):
        hip.sod_loss([torch.zeros(1, 1, 8, 8)], lab)                     # a host tensor: no CPU fallback


# ------------------------------------------------------------------------------------------------------- Adam
def _adam_reference(ps, grads_per_step, lr, betas, eps, wd):
    """the oracle's restatement of the reference's optimizer (oracle/ops.py adam_steps, pinned against torch.optim.Adam on the
    CPU in tests/test_oracle.py), fp64"""
    return oo.adam_steps(ps, grads_per_step, lr, betas, eps, wd)


ADAM_SHAPES = [(1,), (3,), (4,), (5,), (8191,), (8192,), (8193,), (3, 7, 11), (100003,), (64, 1, 7, 7), (1024, 512), (2, 16389)]


@pytest.mark.parametrize("wd", [0.0, 0.01])
def test_adam_step_against_fp64(wd):
    from tramba_amd import train
    g = torch.Generator().manual_seed(7)
    ps = [torch.randn(s, generator=g) for s in ADAM_SHAPES]
    steps = [[torch.randn(s, generator=g) * (10.0 ** (i % 3 - 2)) for i, s in enumerate(ADAM_SHAPES)] for _ in range(4)]
    lr, betas, eps = 1e-2, (0.9, 0.999), 1e-8
    want_p, want_m, want_v = _adam_reference(ps, steps, lr, betas, eps, wd)
    params = [torch.nn.Parameter(p.to(DEV)) for p in ps]
    opt = train.Adam(params, lr, betas=betas, eps=eps, weight_decay=wd)
    for grads in steps:
        for p, gr in zip(params, grads):
            p.grad = gr.to(DEV)
        opt.step()
    for i, p in enumerate(params):
        st = opt.state[p]
        assert float(st["step"]) == len(steps) and st["step"].is_cuda
        assert _rel_l2(p.detach(), want_p[i]) < 1e-6, (ADAM_SHAPES[i], _rel_l2(p.detach(), want_p[i]))
        assert _rel_l2(st["exp_avg"], want_m[i]) < 1e-6
        assert _rel_l2(st["exp_avg_sq"], want_v[i]) < 1e-6


def test_adam_matches_torch_on_the_device_and_shares_its_state_dict():
    """same trajectory as torch.optim.Adam (foreach and fused forms) on the device; a state_dict of either loads into the
    other and the trajectories continue together (train.py:254-262 resume files)"""
    from tramba_amd import train
    g = torch.Generator().manual_seed(11)
    shapes = [(257,), (64, 33), (5, 5, 3, 3), (12289,)]
    init = [torch.randn(s, generator=g) for s in shapes]
    grads = [[torch.randn(s, generator=g) for s in shapes] for _ in range(6)]

    def run(make, n0, n1, state=None):
        params = [torch.nn.Parameter(p.to(DEV)) for p in (state[0] if state else init)]
        opt = make([{"params": params[:2], "lr": 1e-3}, {"params": params[2:], "lr": 1e-2}])
        if state:
            opt.load_state_dict(state[1])
        for k in range(n0, n1):
            for p, gr in zip(params, grads[k]):
                p.grad = gr.to(DEV)
            opt.step()
        return [p.detach().clone() for p in params], opt.state_dict()

    ours = lambda gs: train.Adam(gs, 1e-2)                                   # noqa: E731
    fused = lambda gs: torch.optim.Adam(gs, 1e-2, fused=True)                # noqa: E731
    foreach = lambda gs: torch.optim.Adam(gs, 1e-2, foreach=True)            # noqa: E731
    a, sa = run(ours, 0, 6)
    for other in (fused, foreach):
        b, _ = run(other, 0, 6)
        for x, y in zip(a, b):
            assert _rel_l2(x, y) < 1e-6
    # three steps in torch's optimizer, its state_dict loaded into ours for the remaining three -- and the other way round
    half_t = run(fused, 0, 3)
    half_o = run(ours, 0, 3)
    for x, y in zip(run(ours, 3, 6, half_t)[0], a):
        assert _rel_l2(x, y) < 1e-6
    for x, y in zip(run(fused, 3, 6, half_o)[0], a):
        assert _rel_l2(x, y) < 1e-6
    # a checkpoint written by the non-capturable form keeps its step counters on the host: they move to the device
    half_f = run(foreach, 0, 3)
    cpu_sd = copy.deepcopy(half_f[1])
    for st in cpu_sd["state"].values():
        st["step"] = st["step"].cpu()
    for x, y in zip(run(ours, 3, 6, (half_f[0], cpu_sd))[0], a):
        assert _rel_l2(x, y) < 1e-6
    assert set(sa["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
    assert sa["param_groups"][0]["lr"] == 1e-3 and sa["param_groups"][1]["lr"] == 1e-2


def test_adam_unaligned_gradients_skipped_parameters_and_graph_replay():
    """gradients that are views at odd offsets of a flat bucket (the data-parallel reducer's `.grad` views) take the
    unaligned path; a parameter without a gradient is not stepped; a captured step replays with the captured addresses"""
    from tramba_amd import train
    g = torch.Generator().manual_seed(3)
    shapes = [(1001,), (33, 5), (7,), (4096,)]
    init = [torch.randn(s, generator=g) for s in shapes]
    grads = [[torch.randn(s, generator=g) for s in shapes] for _ in range(3)]
    want_p, _, _ = _adam_reference([init[0], init[1], init[3]], [[gs[0], gs[1], gs[3]] for gs in grads], 1e-2, (0.9, 0.999), 1e-8, 0.0)
    params = [torch.nn.Parameter(p.to(DEV)) for p in init]
    opt = train.Adam(params, 1e-2, capturable=True)
    flat = torch.zeros(1 + sum(p.numel() for p in params), device=DEV)
    off = 1
    for p in params:
        p.grad = None
    views = []
    for p in params:
        views.append(flat[off:off + p.numel()].view_as(p))
        off += p.numel()
    assert any(v.data_ptr() % 16 for v in views)

    def load(k):
        for i, (p, v) in enumerate(zip(params, views)):
            v.copy_(grads[k][i].to(DEV))
            p.grad = None if i == 2 else v

    load(0)
    opt.step()                                     # eager: creates the state
    assert len(opt.state[params[2]]) == 0 and torch.equal(params[2].detach().cpu(), init[2])
    load(1)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):                   # records, executes nothing
        opt.step()
    graph.replay()                                  # step 2 on the gradients of step 1 ...
    load(2)
    graph.replay()                                  # ... step 3 on those now in the bucket
    torch.cuda.synchronize()
    for p, w in zip([params[0], params[1], params[3]], want_p):
        assert _rel_l2(p.detach(), w) < 1e-6
    assert float(opt.state[params[0]]["step"]) == 3


def test_adam_refuses_host_tensors_and_unsupported_modes():
    from tramba_amd import hip, train
    p = torch.nn.Parameter(torch.zeros(8))
    p.grad = torch.ones(8)
    with pytest.raises(hip.TrambaHipError):
        train.Adam([p], 1e-3).step()                 # no CPU fallback
    q = torch.nn.Parameter(torch.zeros(8, device=DEV))
    q.grad = torch.ones(8, device=DEV)
    opt = train.Adam([q], 1e-3)
    opt.param_groups[0]["amsgrad"] = True
    with pytest.raises(hip.TrambaHipError):
        opt.step()


def test_get_opt_builds_the_library_optimizer_with_the_reference_groups():
    """train.py:266-280: encoder parameters at a tenth of the rate; on the device the step is the library's kernel"""
    import tramba_amd as ta
    from tramba_amd import train
    m = ta.bulid_model(use_pretrain=False, img_size=64).to(DEV)
    opt = train.get_opt(1e-4, m, capturable=True)
    assert isinstance(opt, train.Adam) and isinstance(opt, torch.optim.Adam)
    assert [g["lr"] for g in opt.param_groups] == [1e-5, 1e-4]
    names = dict(m.named_parameters())
    enc = {id(p) for n, p in names.items() if "encoder" in n}
    assert {id(p) for p in opt.param_groups[0]["params"]} == enc
    assert np.isclose(sum(p.numel() for g in opt.param_groups for p in g["params"]), sum(p.numel() for p in m.parameters()))


def test_adam_follows_state_entries_replaced_behind_its_back():
    """the cached pointer arrays are only valid for the state tensors they were built from: after `opt.state.clear()` or a
    hand-made replacement of `exp_avg` the next step must update (and report in state_dict) the NEW tensors, not the orphans"""
    from tramba_amd import train
    g = torch.Generator().manual_seed(5)
    p = torch.nn.Parameter(torch.randn(1000, generator=g).to(DEV))
    opt = train.Adam([p], 1e-2)
    p.grad = torch.randn(1000, generator=g).to(DEV)
    opt.step()
    old_avg = opt.state[p]["exp_avg"]
    opt.state[p]["exp_avg"] = torch.zeros_like(p)          # replaced: the plan's pointer is stale now
    opt.step()
    new_avg = opt.state[p]["exp_avg"]
    assert new_avg is not old_avg and float(new_avg.abs().max()) > 0
    assert torch.allclose(new_avg, 0.1 * p.grad)           # (1 - beta1) * g on a zero average
    opt.state.clear()
    opt.step()
    assert float(opt.state[p]["step"]) == 1 and torch.allclose(opt.state[p]["exp_avg"], 0.1 * p.grad)


def test_loss_weight_count_and_wide_labels():
    """a weight list that does not match the outputs raises (the fallback would have, the kernel would have zero-filled); a label
    wider than the gradient kernel's LDS rows (4096) with a resized output takes the framework path instead of failing in backward"""
    from tramba_amd import train
    lab = (torch.rand(1, 1, 8, 8, device=DEV) > 0.5).float()
    outs = [torch.randn(1, 1, 4, 4, device=DEV), torch.randn(1, 1, 8, 8, device=DEV)]
    with pytest.raises(ValueError):
        train.tramba_loss(outs, lab, [1.0])
    wide = (torch.rand(1, 1, 2, 4100, device=DEV) > 0.5).float()
    small = torch.randn(1, 1, 1, 2050, device=DEV, requires_grad=True)
    assert not train._loss_on_device([small], wide)
    train.tramba_loss([small], wide).backward()
    assert small.grad is not None and bool(torch.isfinite(small.grad).all())
