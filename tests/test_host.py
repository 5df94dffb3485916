"""CPU tests of the host side: the C-ABI library loads and exports every symbol the header
declares, the native scan-table generators agree with the oracle and the reference hashes, and the
module tree is state_dict-compatible with the reference.  No compute kernel is launched."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from oracle import scan_tables as st

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported_and_bound():
    from tramba_amd import hip
    hdr = open(os.path.join(ROOT, "include", "tramba_hip.h")).read()
    declared = set(re.findall(r"\b(tramba_[a-z0-9_]+)\s*\(", hdr))
    lib = hip.lib()
    assert lib.tramba_abi_version() == 7
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/tramba_hip.h but not exported"
    assert declared == set(hip.SIGNATURES), declared ^ set(hip.SIGNATURES)


def test_bad_arguments_are_rejected_not_fatal():
    from tramba_amd import hip
    lib = hip.lib()
    assert lib.tramba_scan_family_k(99) < 0 and b"family" in lib.tramba_last_error()
    out = np.empty(4 * 144, dtype=np.int32)
    assert lib.tramba_scan_table(hip.SCAN_WINDOW, 12, 12, 5, out.ctypes.data) < 0  # 5 does not divide 12
    assert lib.tramba_scan_table(hip.SCAN_LINE, 12, 10, 0, out.ctypes.data) < 0  # not square
    # null tensors / bad shapes never reach a launch
    assert lib.tramba_selective_scan_fwd(None, None, None, None, None, None, None, None, None, 1, 4, 4, 1, 8, 0, 0, 1, None) < 0
    with pytest.raises(hip.TrambaHipError):
        hip.layernorm_cl(torch.zeros(2, 8), torch.ones(8), torch.zeros(8))  # CPU tensor: no fallback


@pytest.mark.parametrize("fam", ["raster", "line", "helix", "window", "dilation"])
@pytest.mark.parametrize("h", [12, 24, 48, 96, 16, 32, 64, 192])
def test_native_tables_equal_oracle(fam, h):
    from tramba_amd import hip
    got = hip.scan_table_host(fam, h)
    assert np.array_equal(got, st.table(fam, h))
    ptr, idx = hip.scan_table_inverse_host(got)
    flat = got.reshape(-1)
    assert ptr[0] == 0 and ptr[-1] == flat.size and np.all(np.diff(ptr) >= 0)
    assert np.array_equal(np.sort(idx), np.arange(flat.size))
    assert np.array_equal(flat[idx], np.repeat(np.arange(h * h), np.diff(ptr)))


def test_native_tables_match_reference_hashes(golden_meta):
    from tramba_amd import hip
    for h in (12, 24, 48, 96):
        for fam in ("line", "dilation", "window"):
            got = [st.table_hash(r) for r in hip.scan_table_host(fam, h)]
            assert got == golden_meta["G1"][f"{fam}_{h}"]
    for h in (7, 14, 28, 56):
        assert [st.table_hash(r) for r in hip.scan_table_host("line", h)] == golden_meta["G1"][f"line_{h}"]


def test_state_dict_manifests_match_reference(golden_meta):
    import tramba_amd as ta
    mv = ta.bulid_model(use_pretrain=False)
    assert {k: list(v.shape) for k, v in mv.state_dict().items()} == {k: s for k, s in golden_meta["G6_tramba_v"]}
    assert sum(p.numel() for p in mv.parameters()) == golden_meta["G6_tramba_v_params"]
    enc = [n for n, _ in mv.named_parameters() if "encoder" in n]
    assert enc and all(n.startswith("vssm_encoder.") for n in enc)  # train.py:266-269 optimizer split
    mr = ta.bulid_model_enc("Tramba-R-TSOD")
    assert {k: list(v.shape) for k, v in mr.state_dict().items()} == {k: s for k, s in golden_meta["G6_tramba_r"]}
    assert sum(p.numel() for p in mr.parameters()) == golden_meta["G6_tramba_r_params"]


@pytest.mark.parametrize("tag,name", [("s", "Tramba-S-TSOD"), ("p", "Tramba-P-SOD")])
def test_encoder_variant_manifests_match_reference(golden_enc_meta, tag, name):
    """Swin-B / PVTv2-b4 in front of the decoder (Trambav6_enc.py:167-192): same names, shapes (buffers included)
    and parameter count, so the ImageNet encoder checkpoints and Tramba-S / Tramba-P checkpoints load by name."""
    import tramba_amd as ta
    m = ta.bulid_model_enc(name)
    assert {k: list(v.shape) for k, v in m.state_dict().items()} == {k: s for k, s in golden_enc_meta[f"G6_tramba_{tag}"]}
    assert sum(p.numel() for p in m.parameters()) == golden_enc_meta[f"G6_tramba_{tag}_params"]
    assert all(n.startswith("encoder.") for n, _ in m.named_parameters() if "encoder" in n)      # train.py:266-269


def test_linear2d_accepts_conv_checkpoints():
    import tramba_amd as ta
    lin = ta.Linear2d(8, 4, bias=False)
    w = torch.randn(4, 8, 1, 1)
    lin.load_state_dict({"weight": w})
    assert torch.equal(lin.weight, w.view(4, 8))


def test_get_model_build_surface():
    import types
    import tramba_amd as ta
    args = types.SimpleNamespace(img_size=384, pretrained_path="")
    for name in ("Tramba-R-TSOD", "Tramba-S-TSOD", "Tramba-P-SOD"):            # get_model.py:2-31
        assert isinstance(ta.build(name, args), ta.BaseUMambaEnc)
    with pytest.raises(NotImplementedError):
        ta.build("BaseUMamba-SOD", args)
    assert ta.build("no-such-model", args) is None


def test_graph_wrappers_validate_before_touching_the_device():
    """tramba_amd/graph.py: wrong mode / optimizer / tensor placement are host-side errors, not capture failures."""
    import tramba_amd as ta
    net = torch.nn.Linear(4, 2)
    with pytest.raises(RuntimeError, match="eval"):
        ta.GraphedForward(net)                                          # training mode
    fwd = ta.GraphedForward(net.eval())
    with pytest.raises(RuntimeError, match="device tensor"):
        fwd(torch.zeros(1, 4))                                          # no CPU path
    net.train()
    with pytest.raises(RuntimeError, match="training mode"):
        fwd(torch.zeros(1, 4))
    with pytest.raises(RuntimeError, match="capturable"):
        ta.GraphedTrainStep(net, torch.optim.Adam(net.parameters(), 1e-3))

    class TwoRanksHostRead:                                             # a reducer whose step reads flags on the host
        world, find_unused = 2, True
    with pytest.raises(RuntimeError, match="find_unused"):
        ta.GraphedTrainStep(net, torch.optim.Adam(net.parameters(), 1e-3, capturable=True), reducer=TwoRanksHostRead())


def test_load_pretrained_base_follows_the_reference_rules(tmp_path, capsys):
    """Models/vmamba.py:707-732 on a synthetic VMamba-style checkpoint: `layers.i.downsample.*` -> `downsample.i.*`,
    classifier keys skipped, unknown keys reported and ignored, (out,in,1,1) conv weights accepted for Linear2d, any
    other shape mismatch an AssertionError, uncovered parameters left alone and reported."""
    from tramba_amd.modules import Linear2d, VSSMEncoder, load_pretrained_Base
    torch.manual_seed(0)
    src = VSSMEncoder(depths=[1, 1, 1, 1], dims=16, imgsize=64)
    torch.manual_seed(1)
    dst = VSSMEncoder(depths=[1, 1, 1, 1], dims=16, imgsize=64)
    lin = {n + ".weight" for n, m in src.named_modules() if isinstance(m, Linear2d)}
    left_out = "layers.3.blocks.0.norm.weight"
    ck = {}
    for k, v in src.state_dict().items():
        if k == left_out:
            continue
        if k.startswith("downsample."):                                 # VMamba keeps the downsample inside the stage
            i, rest = k.split(".", 2)[1:]
            k = f"layers.{i}.downsample.{rest}"
        ck[k] = v.clone().view(*v.shape, 1, 1) if k in lin else v.clone()
    assert any(".downsample." in k for k in ck) and any(v.dim() == 4 and v.shape[2:] == (1, 1) for k, v in ck.items() if k in lin)
    ck["classifier.head.weight"] = torch.randn(1000, 128)
    ck["classifier.norm.bias"] = torch.randn(128)
    ck["not_a_module.weight"] = torch.randn(3)
    path = str(tmp_path / "vssm.pth")
    torch.save({"model": ck}, path)
    keep = dst.state_dict()[left_out].clone()
    assert load_pretrained_Base(dst, ckpt_path=path) is dst
    out = capsys.readouterr().out
    assert "Passing weights: classifier.head.weight" in out and "Module can not find: not_a_module.weight" in out
    assert f"Module {left_out} has not been inited!" in out and out.count("has not been inited") == 1
    got, want = dst.state_dict(), src.state_dict()
    assert list(got) == list(want)
    for k in want:
        assert torch.equal(got[k], keep if k == left_out else want[k]), k
    bad = dict(ck)
    bad["patch_embed.0.bias"] = torch.randn(ck["patch_embed.0.bias"].numel() + 1)
    torch.save({"model": bad}, path)
    with pytest.raises(AssertionError, match="Shape mismatch"):
        load_pretrained_Base(dst, ckpt_path=path)
    bad = dict(ck)
    bad["layers.3.downsample.1.weight"] = torch.randn(2)               # the last stage has no downsample to rename to
    torch.save({"model": bad}, path)
    with pytest.raises(AssertionError):
        load_pretrained_Base(dst, ckpt_path=path)


def test_optimizer_choice_and_no_cpu_fallback():
    """train.get_opt hands a host model the reference's own optimizer (host logic / gloo tests), a device model the library's;
    the library's Adam refuses host tensors instead of falling back (reference train.py:266-280)."""
    import torch
    from tramba_amd import hip, train
    m = torch.nn.Sequential(torch.nn.Linear(4, 3))
    opt = train.get_opt(1e-3, m)
    assert type(opt) is torch.optim.Adam and [g["lr"] for g in opt.param_groups] == [1e-4, 1e-3]
    assert issubclass(train.Adam, torch.optim.Adam)
    p = torch.nn.Parameter(torch.zeros(8))
    p.grad = torch.ones(8)
    lib_opt = train.Adam([p], 1e-3)
    assert lib_opt.state_dict()["param_groups"][0]["betas"] == (0.9, 0.999)
    with pytest.raises(hip.TrambaHipError):
        lib_opt.step()
    assert float(p.abs().sum()) == 0.0                     # nothing moved
    with pytest.raises(hip.TrambaHipError):
        hip.sod_loss([torch.zeros(1, 1, 4, 4)], torch.zeros(1, 1, 4, 4))


def test_bench_relay_keeps_a_printed_line_when_the_rank_job_dies(capfd):
    """bench.py --gpus N starts its ranks as a child job and relays its stdout: when rank 0 has printed the result line and the job
    then dies of an abort (a non-zero code from the launcher), the parent exits 0 -- the measurement stands; with no line the
    code is passed on (VERDICT r3 'next' 3c)."""
    import importlib.util
    import sys
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    ok = bench.relay_child([sys.executable, "-c", "import os, sys; print('{\"metric\": \"x\", \"value\": 1}', flush=True); os.abort()"])
    out = capfd.readouterr()
    assert ok == 0 and '"metric"' in out.out and "exiting 0" in out.err
    bad = bench.relay_child([sys.executable, "-c", "import sys; print('no result'); sys.exit(7)"])
    assert bad == 7
