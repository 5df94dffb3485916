"""CPU tests: the oracle against the golden vectors captured from the reference import
(tests/golden/make_golden.py) and against closed-form known answers."""
import numpy as np
import pytest
import torch

import synth
from oracle import model as om
from oracle import ops as oo
from oracle import scan_tables as st
from oracle import selective_scan as oss


# --------------------------------------------------------------------------- G1 tables
@pytest.mark.parametrize("h", [12, 24, 48, 96])
def test_tables_match_reference_hashes(golden_meta, h):
    g1 = golden_meta["G1"]
    for fam in ("line", "dilation", "window"):
        got = [st.table_hash(r) for r in st.table(fam, h)]
        assert got == g1[f"{fam}_{h}"], (fam, h)


@pytest.mark.parametrize("h", [7, 14, 28, 56, 16, 32, 64, 192])
def test_line_tables_other_sizes(golden_meta, h):
    assert [st.table_hash(r) for r in st.line_table(h, h)] == golden_meta["G1"][f"line_{h}"]


@pytest.mark.parametrize("h,ws", [(16, 4), (16, 8), (32, 8), (64, 16), (192, 16)])
def test_offtable_generators(golden_meta, h, ws):
    g1 = golden_meta["G1"]
    assert [st.table_hash(r) for r in st.dilation_table(h, h)] == g1[f"dilation_{h}"]
    assert [st.table_hash(r) for r in st.window_table(h, h, ws)] == g1[f"window_{h}_ws{ws}"]


def test_default_window_rule():
    # reference constants (csms6s.py:107-108) + the documented rule for off-table sizes
    want = {12: 4, 24: 8, 48: 12, 96: 16, 16: 8, 32: 8, 64: 16, 192: 16}
    assert {h: st.default_window_size(h) for h in want} == want


def test_line_spot_values_and_multiplicity(golden_meta):
    g1 = golden_meta["G1"]
    t = st.line_table(12, 12)
    assert t[0].tolist() == g1["line_12_dir0_flat"]
    mult = np.bincount(t.reshape(-1), minlength=144)
    assert mult.tolist() == g1["line_12_multiplicity"]
    # 75 % coverage per direction, max multiplicity H/2 at the centre
    assert len(np.unique(t[0])) == 108 and np.bincount(t[0]).max() == 6


def test_permutation_families_are_permutations():
    for fam in ("raster", "window", "dilation"):
        for h in (12, 24, 32):
            t = st.table(fam, h)
            assert all(sorted(r.tolist()) == list(range(h * h)) for r in t)


# --------------------------------------------------------------------------- G2 scan/merge
@pytest.mark.parametrize("fam,k", [("raster", 4), ("helix", 8), ("window", 4), ("dilation", 4)])
def test_scan_merge_match_reference(golden, fam, k):
    x = torch.from_numpy(golden["g2_x"])
    assert torch.equal(synth.synth_input("g2_x", (2, 3, 12, 12)), x)
    assert np.array_equal(oo.cross_scan(x, fam).numpy(), golden[f"g2_scan_{fam}"])
    ys = synth.synth_input("g2_y_" + fam, (2, k, 3, 12, 12))
    got = oo.cross_merge(ys.reshape(2, k, 3, 144), fam, 12, 12).numpy()
    np.testing.assert_allclose(got, golden[f"g2_merge_{fam}"], rtol=0, atol=2e-6)
    # reference: backward(scan) == merge
    np.testing.assert_allclose(got.reshape(2, 3, 12, 12), golden[f"g2_scan_bwd_{fam}"], rtol=0, atol=2e-6)


def test_scan_merge_adjoint_and_multiplicity():
    for fam in ("raster", "helix", "window", "dilation"):
        t = st.table(fam, 12)
        k = t.shape[0]
        x = torch.randn(1, 2, 12, 12, dtype=torch.float64)
        y = torch.randn(1, k, 2, 144, dtype=torch.float64)
        lhs = (oo.cross_scan(x, fam) * y).sum()
        rhs = (x.reshape(1, 2, 144) * oo.cross_merge(y, fam, 12, 12)).sum()
        assert abs(lhs - rhs) < 1e-10
        mult = torch.from_numpy(np.bincount(t.reshape(-1), minlength=144)).double()
        rt = oo.cross_merge(oo.cross_scan(x, fam), fam, 12, 12)
        assert torch.allclose(rt, x.reshape(1, 2, 144) * mult)


# --------------------------------------------------------------------------- G3 DCT
def test_dct_matches_reference(golden):
    w = oo.dct_matrix(12)
    assert np.array_equal(w.numpy(), golden["g3_weight_12"])
    xd = (torch.arange(144, dtype=torch.float32) / 144).view(1, 1, 12, 12)
    high, low = oo.dct2d_split(xd, w, w)
    np.testing.assert_allclose(low.numpy(), golden["g3_arange_low"], atol=3e-6)
    np.testing.assert_allclose(high.numpy(), golden["g3_arange_high"], atol=3e-6)
    # SURVEY 8c G3 probe values
    assert abs(low[0, 0, 0, 0].item() - 5.9583330) < 1e-5 and abs(low[0, 0, 1, 0].item() + 3.4290752) < 1e-5
    x = synth.synth_input("g3_x", (2, 3, 24, 24))
    w24 = oo.dct_matrix(24)
    high, low = oo.dct2d_split(x, w24, w24)
    np.testing.assert_allclose(low.numpy(), golden["g3_low_24"], atol=1e-5)
    np.testing.assert_allclose(high.numpy(), golden["g3_high_24"], atol=1e-5)


def test_dct_is_orthonormal():
    w = oo.dct_matrix(48, torch.float64)
    assert torch.allclose(w @ w.T, torch.eye(48, dtype=torch.float64), atol=1e-6)


# --------------------------------------------------------------------------- G4 blocks
BLOCKS = {
    "ss2d_raster": (lambda p, x: om.ss2d(p, x, "raster"), (2, 16, 12, 12)),
    "vssblock": (om.vss_block, (2, 16, 12, 12)),
    "freqblock": (om.freq_block, (2, 16, 12, 12)),
    "helixblock": (om.multiscale_decoder_block, (2, 16, 12, 12)),
    "freqblock24": (om.freq_block, (1, 32, 24, 24)),
    "helixblock24": (om.multiscale_decoder_block, (1, 32, 24, 24)),
    "patchexpand": (lambda p, x: om._expand_shuffle_norm(p, x, 2), (2, 32, 6, 6)),
    "finalexpand": (lambda p, x: om._expand_shuffle_norm(p, x, 4), (2, 8, 6, 6)),
    "freqexpand": (lambda p, x: om._expand_shuffle_norm(p, x, 2), (2, 8, 6, 6)),
}


def block_state(golden_meta, tag, dtype=torch.float32):
    man = golden_meta["G4_manifest"][tag]
    sd = synth.synth_state_dict(man, keep=synth.DCT_KEYS, dtype=dtype)
    for name, shape in man:
        if name not in sd:  # DCT buffers: constants of the architecture
            sd[name] = oo.dct_matrix(shape[0], dtype)
    return sd


@pytest.mark.parametrize("tag", list(BLOCKS))
def test_blocks_forward_and_grads(golden, golden_meta, tag):
    fn, shape = BLOCKS[tag]
    sd = block_state(golden_meta, tag)
    params = [n for n, _ in golden_meta["G4_manifest"][tag] if not any(k in n for k in synth.DCT_KEYS)]
    for n in params:
        sd[n].requires_grad_()
    x = synth.synth_input("g4_" + tag, shape).requires_grad_()
    y = fn(om.SD(sd), x)
    want = golden[f"g4_{tag}_y"]
    assert y.shape == want.shape
    np.testing.assert_allclose(y.detach().numpy(), want, rtol=2e-4, atol=2e-5)
    gy = synth.synth_input("g4_gy_" + tag, tuple(y.shape))
    grads = torch.autograd.grad(y, [x] + [sd[n] for n in params], gy)
    np.testing.assert_allclose(grads[0].numpy(), golden[f"g4_{tag}_dx"], rtol=2e-3, atol=5e-5)
    ref = golden_meta["G4_param_grads"][tag]
    for n, g in zip(params, grads[1:]):
        s, a = ref[n]
        assert abs(float(g.double().abs().sum()) - a) <= 2e-3 * a + 1e-4, n
        assert abs(float(g.double().sum()) - s) <= 2e-3 * a + 1e-4, n


# --------------------------------------------------------------------------- G5/G6 models
def _full_state(manifest):
    sd = synth.synth_state_dict(manifest, keep=synth.DCT_KEYS)
    for name, shape in manifest:
        if name not in sd:
            sd[name] = oo.dct_matrix(shape[0])
    return sd


def test_manifest_counts(golden_meta):
    assert len(golden_meta["G6_tramba_v"]) == 679
    assert golden_meta["G6_tramba_v_params"] == 111_440_000 or abs(golden_meta["G6_tramba_v_params"] - 111.44e6) < 5e4


def test_tramba_v_full_forward(golden, golden_meta):
    torch.set_grad_enabled(False)
    try:
        sd = _full_state(golden_meta["G6_tramba_v"])
        x = synth.synth_input("g5_v", (1, 3, 384, 384))
        outs = om.tramba_v(sd, x)
    finally:
        torch.set_grad_enabled(True)
    assert [tuple(o.shape) for o in outs] == [(1, 1, 24, 24), (1, 1, 48, 48), (1, 1, 96, 96), (1, 1, 384, 384)]
    for i in range(3):
        np.testing.assert_allclose(outs[i].numpy(), golden[f"g5_v_out{i}"], rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(outs[3][:, :, 160:224, 160:224].numpy(), golden["g5_v_out3_crop"], rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(torch.nn.functional.avg_pool2d(outs[3], 8).numpy(), golden["g5_v_out3_pool8"],
                               rtol=1e-3, atol=2e-4)
    pred = torch.sigmoid(outs[3])[0, 0].numpy()
    gt = (synth.synth_input("g5_gt", (384, 384)) > 0.5).numpy()
    assert round(oo.mae_metric(pred, gt), 4) == round(golden_meta["G5_tramba_v_mae"], 4)


def test_tramba_r_full_forward(golden, golden_meta):
    torch.set_grad_enabled(False)
    try:
        sd = _full_state(golden_meta["G6_tramba_r"])
        x = synth.synth_input("g5_r", (1, 3, 384, 384))
        outs = om.tramba_r(sd, x)
    finally:
        torch.set_grad_enabled(True)
    assert [tuple(o.shape) for o in outs] == [(1, 1, 48, 48), (1, 1, 96, 96), (1, 1, 384, 384)]
    for i in range(2):
        np.testing.assert_allclose(outs[i].numpy(), golden[f"g5_r_out{i}"], rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(outs[2][:, :, 160:224, 160:224].numpy(), golden["g5_r_out2_crop"], rtol=1e-3, atol=2e-4)


@pytest.mark.parametrize("tag,fn", [("s", "tramba_s"), ("p", "tramba_p")])
def test_tramba_s_p_full_forward(golden_enc, golden_enc_meta, tag, fn):
    """Swin-B / PVTv2-b4 encoders + decoder against the reference's own forward (tests/golden/make_golden_enc.py)."""
    import tramba_amd as ta
    torch.set_grad_enabled(False)
    try:
        manifest = golden_enc_meta[f"G6_tramba_{tag}"]
        sd = synth.synth_state_dict(manifest, keep=synth.CONST_KEYS)
        consts = ta.bulid_model_enc("Tramba-S-TSOD" if tag == "s" else "Tramba-P-TSOD").state_dict()   # masks, index tables, DCT
        for name, _ in manifest:
            if name not in sd:
                sd[name] = consts[name]
        x = synth.synth_input(f"g5_{tag}", (1, 3, 384, 384))
        outs = getattr(om, fn)(sd, x)
        enc = (om.swin_b_encoder if tag == "s" else lambda p, v: om.pvt_v2_b4_encoder(p, v)[::-1])(om.SD(sd, "encoder."), x)
    finally:
        torch.set_grad_enabled(True)
    for i, f in enumerate(enc):
        np.testing.assert_allclose(torch.nn.functional.avg_pool2d(f, f.shape[-1] // 6).numpy(), golden_enc[f"g5_{tag}_enc{i}_pool"],
                                   rtol=1e-3, atol=2e-4)
    assert [tuple(o.shape) for o in outs] == [(1, 1, 24, 24), (1, 1, 48, 48), (1, 1, 96, 96), (1, 1, 384, 384)]
    for i in range(3):
        np.testing.assert_allclose(outs[i].numpy(), golden_enc[f"g5_{tag}_out{i}"], rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(outs[3][:, :, 160:224, 160:224].numpy(), golden_enc[f"g5_{tag}_out3_crop"], rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(torch.nn.functional.avg_pool2d(outs[3], 8).numpy(), golden_enc[f"g5_{tag}_out3_pool8"],
                               rtol=1e-3, atol=2e-4)


# --------------------------------------------------------------------------- G7 loss / metric
def test_loss_and_metric_known_answers(golden_meta):
    pred = synth.synth_input("g7_pred", (2, 1, 24, 24), scale=2.0)
    mask = (synth.synth_input("g7_mask", (2, 1, 24, 24)) > 0.3).float()
    assert abs(float(oo.iou_loss(pred, mask)) - golden_meta["G7"]["iou_loss"]) < 1e-6
    maes = [oo.mae_metric(torch.sigmoid(pred[i, 0]).numpy(), mask[i, 0].numpy()) for i in range(2)]
    assert abs(np.mean(maes) - golden_meta["G7"]["mae"]) < 1e-7
    total = oo.tramba_loss([pred], mask)
    assert abs(float(total) - (golden_meta["G7"]["iou_loss"] + golden_meta["G7"]["bce"])) < 1e-5


# --------------------------------------------------------------------------- scan KATs (parity unpinned)
def _rand_scan(nb=2, k=4, dper=3, n=1, l=33, seed=0):
    g = torch.Generator().manual_seed(seed)
    kd = k * dper
    r = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)
    return dict(u=r(nb, kd, l), delta=0.5 * r(nb, kd, l), A=-(torch.rand(kd, n, generator=g, dtype=torch.float64) + 0.2),
                B=r(nb, k, n, l), C=r(nb, k, n, l), D=r(kd), delta_bias=0.3 * r(kd))


def test_scan_c_matches_numpy_form():
    for n in (1, 3):
        a = _rand_scan(n=n)
        y1 = oss.selective_scan_fwd(**a)
        y2 = oss.selective_scan_numpy(**a)
        assert (y1 - y2).abs().max() < 1e-12


def test_scan_closed_forms():
    # A = 0, no softplus: h is a running sum of delta*B*u
    nb, k, dper, l = 1, 2, 2, 9
    kd = k * dper
    u = torch.arange(1, l + 1, dtype=torch.float64).repeat(nb, kd, 1)
    delta = torch.full((nb, kd, l), 0.5, dtype=torch.float64)
    A = torch.zeros(kd, 1, dtype=torch.float64)
    B = torch.ones(nb, k, 1, l, dtype=torch.float64) * 2
    C = torch.ones(nb, k, 1, l, dtype=torch.float64) * 3
    y = oss.selective_scan_fwd(u, delta, A, B, C, None, None, False)
    assert torch.allclose(y, 3 * torch.cumsum(0.5 * 2 * u, -1))
    # constant inputs: geometric series  h_l = b (1 - a^l) / (1 - a)
    A = torch.full((kd, 1), -1.0, dtype=torch.float64)
    u1 = torch.ones(nb, kd, l, dtype=torch.float64)
    y = oss.selective_scan_fwd(u1, delta, A, B / 2, C / 3, torch.full((kd,), 0.25, dtype=torch.float64), None, False)
    a, b = np.exp(-0.5), 0.5
    want = torch.tensor([b * (1 - a ** (i + 1)) / (1 - a) + 0.25 for i in range(l)], dtype=torch.float64)
    assert torch.allclose(y[0, 0], want)
    # L = 1 with softplus + bias, incl. the threshold-20 branch
    for raw in (-3.0, 0.7, 25.0):
        d1 = torch.full((1, kd, 1), raw - 0.5, dtype=torch.float64)
        y = oss.selective_scan_fwd(u1[:, :, :1], d1, A, B[..., :1], C[..., :1], None,
                                   torch.full((kd,), 0.5, dtype=torch.float64), True)
        dt = raw if raw > 20 else np.log1p(np.exp(raw))
        assert abs(y[0, 0, 0].item() - 3 * dt * 2) < 1e-12
    # group mapping k(d) = d // (KD/K)
    Bg = torch.stack([torch.ones(1, l), 5 * torch.ones(1, l)]).unsqueeze(0).double()
    y = oss.selective_scan_fwd(u1, delta, torch.zeros(kd, 1, dtype=torch.float64), Bg, torch.ones_like(Bg), None, None, False)
    assert torch.allclose(y[0, 1, -1] * 5, y[0, 2, -1])


def test_scan_backward_matches_autograd_of_numpy_form():
    a = _rand_scan(n=2, l=21)
    ins = {k: v.clone().requires_grad_() for k, v in a.items()}
    dt = torch.nn.functional.softplus(ins["delta"] + ins["delta_bias"][None, :, None], threshold=20)
    rep = 3
    Bx, Cx = ins["B"].repeat_interleave(rep, 1), ins["C"].repeat_interleave(rep, 1)
    h = torch.zeros(2, 12, 2, dtype=torch.float64)
    ys = []
    for l in range(21):
        h = torch.exp(dt[:, :, l, None] * ins["A"][None]) * h + dt[:, :, l, None] * Bx[:, :, :, l] * ins["u"][:, :, l, None]
        ys.append((Cx[:, :, :, l] * h).sum(-1))
    y = torch.stack(ys, -1) + ins["D"][None, :, None] * ins["u"]
    g = torch.randn(y.shape, dtype=torch.float64, generator=torch.Generator().manual_seed(5))
    y.backward(g)
    got = oss.selective_scan_bwd(a["u"], a["delta"], a["A"], a["B"], a["C"], a["D"], a["delta_bias"], g, True)
    for name, t in zip(("u", "delta", "A", "B", "C", "D", "delta_bias"), got):
        assert (t - ins[name].grad).abs().max() < 1e-10, name


@pytest.mark.parametrize("wd", [0.0, 0.02])
def test_adam_restatement_equals_torch_optim_adam(wd):
    """oracle.ops.adam_steps (the update rule of train.py:266-280's optimizer written out) against torch.optim.Adam itself, fp64
    on the CPU, single-tensor and foreach forms, five steps of gradients on several scales: the pin of the reference the HIP
    optimizer kernel is tested against (tests/test_gpu_step_ends.py)."""
    g = torch.Generator().manual_seed(13)
    shapes = [(7,), (3, 5), (2, 3, 3, 3), (1,)]
    init = [torch.randn(s, generator=g, dtype=torch.float64) for s in shapes]
    steps = [[torch.randn(s, generator=g, dtype=torch.float64) * 10.0 ** (i - 2) for i, s in enumerate(shapes)] for _ in range(5)]
    want_p, want_m, want_v = oo.adam_steps(init, steps, 3e-3, (0.9, 0.999), 1e-8, wd)
    for foreach in (False, True):
        params = [torch.nn.Parameter(p.clone()) for p in init]
        opt = torch.optim.Adam(params, 3e-3, weight_decay=wd, foreach=foreach)
        for grads in steps:
            for p, gr in zip(params, grads):
                p.grad = gr.clone()
            opt.step()
        for i, p in enumerate(params):
            st = opt.state[p]
            assert torch.allclose(p.detach(), want_p[i], rtol=1e-12, atol=1e-14)
            assert torch.allclose(st["exp_avg"], want_m[i], rtol=1e-12, atol=1e-14)
            assert torch.allclose(st["exp_avg_sq"], want_v[i], rtol=1e-12, atol=1e-16)
