"""End to end on the GPU, the way run.py drives the reference: a dataset on disk in the reference's folder layout ->
RGB_Dataset loader -> fit() (train_step on the HIP path, files written) -> test_one_epoch metrics -> PNG dump."""
import glob
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import synth  # noqa: E402

pytestmark = pytest.mark.gpu


def _write_split(root, split, names, size):
    for sub in ("image", "mask"):
        os.makedirs(os.path.join(root, split, sub))
    for n in names:
        img, gt = synth.image_pair(n, *size)
        img.save(os.path.join(root, split, "image", n + ".png"))
        gt.save(os.path.join(root, split, "mask", n + ".png"))


def test_disk_dataset_to_metrics(tmp_path):
    import tramba_amd as ta
    from tramba_amd import data, evaluate, train
    from PIL import Image
    root, out = str(tmp_path / "ds"), str(tmp_path / "out")
    _write_split(root, "Train", [f"tr{i}" for i in range(4)], (150, 120))
    _write_split(root, "Test", [f"te{i}" for i in range(3)], (150, 120))
    torch.manual_seed(1026)
    np.random.seed(1026)
    m = ta.bulid_model(use_pretrain=False, img_size=384, dims=128, depths=[2, 2, 2, 2]).cuda().train()
    m.compute_dtype = torch.bfloat16
    opt = train.get_opt(1e-4, m)
    dl = data.train_loader(root, 384, batch_size=2, num_workers=0)   # no forked workers next to a live HIP context
    test_dl = data.eval_loader(root, 384, num_workers=0)
    seen = []

    def evaluate_epoch(model, epoch):
        r = evaluate.test_one_epoch(model, test_dl, weighted=False)
        seen.append(r)
        return float(r["MAE_r"])

    hist = train.fit(m, opt, data.device_batches(dl), epochs=5, base_lr=1e-4, decay_epochs=[3], decay_factors=[0.1],
                     save_model=out, method="Tramba-V-TSOD", evaluate=evaluate_epoch, see=4)
    assert len(hist) == 5 and all(np.isfinite(h["loss"]) for h in hist)
    assert hist[-1]["loss"] < hist[0]["loss"]                         # 10 Adam steps on 4 images do fit them a little
    assert [h["mae"] is not None for h in hist] == [False, False, False, True, True]
    assert hist[3]["lr"] == pytest.approx(1e-5)
    assert all(0.0 <= r["MAE_r"] <= 1.0 and 0.0 <= r["Smeasure_r"] <= 1.0 for r in seen)
    files = [os.path.basename(p) for p in glob.glob(os.path.join(out, "**", "*.pth"), recursive=True)]
    assert any(f.startswith("Tramba-V-TSOD_MAE_") for f in files), files       # best-MAE checkpoint (train.py:243-250)
    assert "Tramba-V-TSOD_resume.pth" in files, files                          # every 5th epoch (train.py:254-262)
    # the same loop with every step replayed as a hipGraph (fresh capturable optimizer, one more epoch)
    m.train()
    hist2 = train.fit(m, train.get_opt(1e-5, m, capturable=True), data.device_batches(dl), epochs=1, base_lr=1e-5,
                      decay_epochs=[], decay_factors=[], save_model=out, method="Tramba-V-TSOD", graph=True)
    assert len(hist2) == 1 and np.isfinite(hist2[0]["loss"]) and hist2[0]["loss"] < hist[0]["loss"]
    written = evaluate.save_predictions(m, test_dl, os.path.join(out, "pred"))
    assert sorted(os.path.basename(p) for p in written) == ["te0.png", "te1.png", "te2.png"]
    with Image.open(written[0]) as im:
        assert im.size == (150, 120) and im.mode == "L"


def test_graphed_forward_replays_the_eager_result():
    import tramba_amd as ta
    torch.manual_seed(3)
    m = ta.bulid_model(use_pretrain=False, img_size=384, dims=128, depths=[2, 2, 2, 2]).cuda()
    m.compute_dtype = torch.bfloat16
    with pytest.raises(RuntimeError, match="eval"):
        ta.GraphedForward(m)                                         # still in training mode
    m.eval()
    gf = ta.GraphedForward(m, strict=True)
    for batch, seed in ((1, 0), (1, 1), (2, 2), (1, 3)):             # two shapes -> two graphs, replayed alternately
        x = torch.randn(batch, 3, 384, 384, generator=torch.Generator().manual_seed(seed)).cuda()
        with torch.no_grad():
            want = [o.clone() for o in m(x)]
        got = gf(x)
        assert len(got) == len(want) and all(torch.equal(g, w) for g, w in zip(got, want)), (batch, seed)
    assert len(gf._graphs) == 2 and all(v is not None for v in gf._graphs.values())
    with pytest.raises(RuntimeError, match="device tensor"):
        gf(torch.zeros(1, 3, 384, 384))


def test_graphed_evaluation_gives_the_eager_metrics(tmp_path):
    import tramba_amd as ta
    from tramba_amd import data, evaluate
    root = str(tmp_path / "ds")
    _write_split(root, "Test", [f"te{i}" for i in range(3)], (150, 120))
    torch.manual_seed(5)
    m = ta.bulid_model(use_pretrain=False, img_size=384, dims=128, depths=[2, 2, 2, 2]).cuda().eval()
    m.compute_dtype = torch.bfloat16
    dl = data.eval_loader(root, 384, num_workers=0)
    eager = evaluate.test_one_epoch(m, dl, weighted=False)
    graphed = evaluate.test_one_epoch(m, dl, weighted=False, graph=True)
    assert eager.keys() == graphed.keys()
    for k in eager:
        assert np.array_equal(np.asarray(eager[k]), np.asarray(graphed[k])), k
    a = evaluate.save_predictions(m, dl, str(tmp_path / "a"))
    b = evaluate.save_predictions(m, dl, str(tmp_path / "b"), graph=True)
    assert [open(p, "rb").read() for p in a] == [open(p, "rb").read() for p in b]


def test_graphed_train_step_follows_the_eager_step():
    """The whole optimisation step replayed as one hipGraph walks the same trajectory as eager launches (stochastic
    depth off so that both consume no random numbers; index_add atomics and the device-side Adam step counters leave
    rounding-level differences, which six Adam steps amplify a little)."""
    import tramba_amd as ta
    from tramba_amd import train
    x = torch.randn(2, 3, 384, 384, generator=torch.Generator().manual_seed(0)).cuda()
    y = (torch.rand(2, 1, 384, 384, generator=torch.Generator().manual_seed(1)) > 0.7).float().cuda()

    def fresh(capturable):
        torch.manual_seed(11)
        m = ta.bulid_model(use_pretrain=False, img_size=384).cuda().train()
        for mod in m.modules():
            if isinstance(mod, ta.DropPath):
                mod.drop_prob = 0.0
        m.compute_dtype = torch.bfloat16
        return m, train.get_opt(1e-4, m, capturable=capturable)

    m, opt = fresh(False)
    with pytest.raises(RuntimeError, match="capturable"):
        ta.GraphedTrainStep(m, opt)
    eager = [float(train.train_step(m, opt, x, y)) for _ in range(8)]
    del m, opt
    m, opt = fresh(True)
    step = ta.GraphedTrainStep(m, opt)
    probe = next(p for n, p in m.named_parameters() if n.endswith("weight") and p.ndim == 2)
    start = probe.detach().clone()
    got = [float(step(x, y))]                # eager warm-up steps (undone), capture, replay: exactly step 1
    after_one = probe.detach().clone()
    assert not torch.equal(start, after_one)
    got += [float(step(x, y)) for _ in range(5)]
    assert not torch.equal(after_one, probe)                         # the replay really updates the weights
    assert np.allclose(got, eager[:6], rtol=3e-2), (got, eager)
    assert got[0] == pytest.approx(eager[0], rel=1e-5)               # same initial weights, same batch: same first loss
    assert got[3] > got[4] > got[5]                                  # and it is still fitting the batch
    x1, y1 = x[:1].contiguous(), y[:1].contiguous()                  # a short last batch: its own graph, no re-capture of
    assert np.isfinite(float(step(x1, y1))) and len(step._graphs) == 2     # the full-batch one
    train.adjust_learning_rate(opt, 1, [1], 1e-4, [0.1])             # new learning rates -> a new capture
    later = float(step(x, y))
    assert len(step._graphs) == 1 and step._lr_key == pytest.approx((1e-6, 1e-5)) and np.isfinite(later)


def test_inference_after_graphed_steps_sees_the_current_weights():
    """A hipGraph replay rewrites the weights in place; the inference path keeps derived copies of them (packed stencils,
    -exp(A_logs), fp32 views, head biases, bf16 shadows) keyed on the parameters' version counters.  Two rounds of
    [graphed steps -> evaluation] must evaluate the CURRENT weights: compare with a fresh model loaded from the
    state_dict (the first evaluation used to freeze those caches)."""
    import tramba_amd as ta
    from tramba_amd import train
    x = torch.randn(2, 3, 384, 384, generator=torch.Generator().manual_seed(0)).cuda()
    y = (torch.rand(2, 1, 384, 384, generator=torch.Generator().manual_seed(1)) > 0.7).float().cuda()
    xe = torch.randn(1, 3, 384, 384, generator=torch.Generator().manual_seed(2)).cuda()
    torch.manual_seed(7)
    m = ta.bulid_model(use_pretrain=False, img_size=384).cuda().train()
    m.compute_dtype = torch.bfloat16
    step = ta.GraphedTrainStep(m, train.get_opt(1e-3, m, capturable=True))
    with pytest.raises(RuntimeError, match="eval mode"):
        m.eval()
        step(x, y)
    m.train()
    outs = []
    for _ in range(2):
        for _ in range(2):
            step(x, y)
        m.eval()
        with torch.no_grad():
            outs.append([o.clone() for o in m(xe)])
        m.train()
    assert not torch.equal(outs[0][-1], outs[1][-1])                 # two more steps at lr 1e-3 do move the prediction
    fresh = ta.bulid_model(use_pretrain=False, img_size=384).cuda()
    fresh.load_state_dict(m.state_dict())
    fresh.compute_dtype = torch.bfloat16
    fresh.eval()
    with torch.no_grad():
        want = fresh(xe)
    for g, w in zip(outs[1], want):
        assert torch.equal(g, w)


def test_bench_line_survives_a_stuck_training_leg():
    """bench.py's contract under the driver: ONE JSON line with the forward result.  A training leg that does not finish (a
    collective that never completes at N > 1) must not cost that line: with the watchdog's deadline set to almost nothing,
    rank 0 still prints the forward measurement, train = {"error": ...}, and exits 0."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    run = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--no-latency",
                          "--no-cpu-baseline", "--train-timeout", "0.05"], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stderr[-2000:]
    line = json.loads(run.stdout.strip().splitlines()[-1])
    assert line["metric"].startswith("images/sec fwd") and line["value"] > 30.0 and line["n_gpus"] == 1
    assert line["roofline"]["frac"] > 0 and line["cpu_baseline"] is None
    assert "error" in line["train"] and "not finished" in line["train"]["error"]


def test_bench_starts_its_own_ranks_and_reports_the_dp_step():
    """`python bench.py --gpus 2` (the driver's form at N > 1 when it does not bring its own launcher): the parent starts the
    ranks as a child torch.distributed.run job, every rank runs a replica of the forward and the data-parallel training
    step, rank 0 prints ONE line with n_gpus = 2 and the gradient bytes of a step.  Two ranks on the one card of the test
    box, gloo instead of RCCL (one device cannot host two RCCL ranks)."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    run = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "4",
                          "--warmup", "1", "--no-latency"], capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, run.stderr[-2000:]
    lines = [ln for ln in run.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] > 30.0 and line["config"]["global_batch"] == 8
    t = line["train"]
    assert t["global_batch"] == 16 and t["grad_bytes_per_step"] == t["grad_bucket_bytes"] > 4e8
    assert t["allreduce"]["allreduce_alone_ms"] > 0 and "dp2" in t["parallelism"] and t["value"] > 0
