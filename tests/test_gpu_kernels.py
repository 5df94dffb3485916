"""GPU parity tests, kernel level: every entry of the C ABI against the CPU oracle
(or the obvious fp64 torch formula) on seeded inputs.  Run with `-m gpu` on an MI355X."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ops as oo
from oracle import scan_tables as st
from oracle import selective_scan as oss

pytestmark = pytest.mark.gpu

DEV = "cuda"


def hip():
    from tramba_amd import hip as h
    return h


def _scan_inputs(nb, k, dper, n, l, dtype, seed=0):
    g = torch.Generator().manual_seed(seed)
    kd = k * dper
    r = lambda *s: torch.randn(*s, generator=g)
    a = dict(u=r(nb, kd, l), delta=0.5 * r(nb, kd, l) - 0.5, A=-(torch.rand(kd, n, generator=g) + 0.2),
             B=r(nb, k, n, l), C=r(nb, k, n, l), D=1 + 0.1 * r(kd), delta_bias=0.3 * r(kd))
    for key in ("u", "delta", "B", "C"):
        a[key] = a[key].to(dtype)
    return a


def _tol(dtype):
    # inputs are rounded to `dtype` BEFORE both paths; the kernel computes in fp32, the oracle in fp64
    return dict(rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(2, 4, 8, 1, 144), (1, 4, 6, 1, 576), (2, 8, 4, 1, 2304), (1, 2, 3, 1, 1000),
                                   (1, 4, 2, 1, 9216), (1, 1, 5, 1, 37), (2, 2, 3, 2, 200), (1, 2, 2, 4, 64)])
def test_selective_scan_fwd(dtype, shape):
    nb, k, dper, n, l = shape
    a = _scan_inputs(nb, k, dper, n, l, dtype)
    want = oss.selective_scan_fwd(a["u"].float(), a["delta"].float(), a["A"], a["B"].float(), a["C"].float(),
                                  a["D"], a["delta_bias"], True)
    g = {k_: v.to(DEV) for k_, v in a.items()}
    out, ckpt = hip().selective_scan_fwd(g["u"], g["delta"], g["A"], g["B"], g["C"], g["D"], g["delta_bias"], True, True)
    assert out.dtype == torch.float32 and out.shape == (nb, k * dper, l)
    np.testing.assert_allclose(out.cpu().double().numpy(), want.numpy(), **_tol(dtype))
    assert ckpt.shape == (nb, k * dper, hip().selective_scan_nchunk(l, dtype), n)
    # no softplus / no D / no bias / same-dtype output
    want2 = oss.selective_scan_fwd(a["u"].float(), a["delta"].float().abs(), a["A"], a["B"].float(), a["C"].float(),
                                   None, None, False)
    out2, _ = hip().selective_scan_fwd(g["u"], g["delta"].abs(), g["A"], g["B"], g["C"], None, None, False, False, False)
    assert out2.dtype == dtype
    lo = dict(rtol=2e-2, atol=2e-2) if dtype != torch.float32 else _tol(dtype)
    np.testing.assert_allclose(out2.cpu().double().numpy(), want2.numpy(), **lo)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 4, 8, 1, 144), (1, 4, 6, 1, 576), (1, 2, 3, 1, 1000), (1, 8, 2, 1, 2304), (1, 1, 5, 1, 37),
                                   (2, 2, 3, 2, 200), (1, 2, 2, 4, 64), (1, 4, 4, 4, 1153), (2, 1, 8, 2, 2304)])
def test_selective_scan_bwd(dtype, shape):
    """every gradient of selective_scan_cuda_oflex.bwd (csms6s.py:920-922) against the fp64 oracle, d_state 1, 2 and 4
    (dA (KD, N), dB / dC (B, K, N, L)), ragged and unaligned lengths"""
    nb, k, dper, n, l = shape
    a = _scan_inputs(nb, k, dper, n, l, dtype, seed=3)
    dout = torch.randn(nb, k * dper, l, generator=torch.Generator().manual_seed(9))
    f = lambda t: t.float()
    want = oss.selective_scan_bwd(f(a["u"]), f(a["delta"]), a["A"], f(a["B"]), f(a["C"]), a["D"], a["delta_bias"], dout, True)
    g = {k_: v.to(DEV) for k_, v in a.items()}
    _, ckpt = hip().selective_scan_fwd(g["u"], g["delta"], g["A"], g["B"], g["C"], g["D"], g["delta_bias"], True, True)
    got = hip().selective_scan_bwd(g["u"], g["delta"], g["A"], g["B"], g["C"], g["D"], g["delta_bias"], dout.to(DEV), ckpt, True)
    names = ("du", "ddelta", "dA", "dB", "dC", "dD", "dbias")
    for name, x, w in zip(names, got, want):
        w = w.numpy()
        scale = max(1.0, float(np.abs(w).max()))
        tol = 3e-4 if (dtype == torch.float32 or name not in ("du", "ddelta")) else 1.5e-2
        err = np.abs(x.cpu().double().numpy() - w).max() / scale
        assert err < tol, (name, err)


@pytest.mark.parametrize("fam", ["raster", "helix", "window", "dilation", "line"])
@pytest.mark.parametrize("h", [12, 24, 16])
def test_cross_scan_merge(fam, h):
    order = hip().scan_order(fam, h, h, torch.device(DEV))
    assert np.array_equal(order.table.cpu().numpy(), st.table(fam, h))
    x = torch.randn(2, 5, h, h)
    xs = hip().cross_scan(x.to(DEV), order)
    assert torch.equal(xs.cpu(), oo.cross_scan(x, fam))
    ys = torch.randn(2, order.k, 5, h * h)
    y = hip().cross_merge(ys.to(DEV), order)
    np.testing.assert_allclose(y.cpu().numpy(), oo.cross_merge(ys, fam, h, h).numpy(), rtol=1e-5, atol=1e-5)
    for dtype in (torch.bfloat16, torch.float16):
        xs16 = hip().cross_scan(x.to(DEV, dtype), order)
        assert torch.equal(xs16.cpu(), oo.cross_scan(x.to(dtype), fam))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("c", [16, 64, 128, 200, 256, 1024, 2048])
def test_layernorm_cl(dtype, c):
    x = torch.randn(3, 7, c) * 2 + 0.5
    w, b = 1 + 0.1 * torch.randn(c), 0.1 * torch.randn(c)
    xq = x.to(dtype)
    want = F.layer_norm(xq.double(), (c,), w.double(), b.double(), 1e-5)
    for act, fn in ((0, lambda t: t), (2, F.gelu)):
        got = hip().layernorm_cl(xq.to(DEV), w.to(DEV), b.to(DEV), 1e-5, act)
        tol = 1e-5 if dtype == torch.float32 else 2e-2
        np.testing.assert_allclose(got.cpu().double().numpy(), fn(want).numpy(), rtol=tol, atol=tol)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cfg", [(2, 6, 8, 2), (1, 5, 16, 4), (2, 3, 128, 2), (1, 4, 32, 4)])
def test_shuffle_norm_cl(dtype, cfg):
    b, h, c, p = cfg
    x = torch.randn(b, h, h, p * p * c).to(dtype)
    w, bb = 1 + 0.1 * torch.randn(c), 0.1 * torch.randn(c)
    xn = x.double().permute(0, 3, 1, 2)
    want = oo.layernorm2d(oo.pixel_shuffle_groups(xn, p), w.double(), bb.double()).permute(0, 2, 3, 1)
    got = hip().shuffle_norm_cl(x.to(DEV), w.to(DEV), bb.to(DEV), p)
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    np.testing.assert_allclose(got.cpu().double().numpy(), want.numpy(), rtol=tol, atol=tol)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cfg", [(2, 12, 32), (1, 7, 6), (1, 24, 64), (1, 5, 3)])
def test_dwconv_and_dwms_cl(dtype, cfg):
    b, h, c = cfg
    x = torch.randn(b, h, h, c).to(dtype)
    xn = x.double().permute(0, 3, 1, 2)
    ws = {k: 0.2 * torch.randn(c, 1, k, k) for k in (3, 5, 7)}
    bs = {k: 0.1 * torch.randn(c) for k in (3, 5, 7)}
    tol = 1e-5 if dtype == torch.float32 else 3e-2
    want = F.silu(F.conv2d(xn, ws[3].double(), None, padding=1, groups=c)).permute(0, 2, 3, 1)
    got = hip().dwconv_cl(x.to(DEV), *hip().dw_pack(ws[3].to(DEV)), 1)
    np.testing.assert_allclose(got.cpu().double().numpy(), want.numpy(), rtol=tol, atol=tol)
    for k in (5, 7):
        want = F.conv2d(xn, ws[k].double(), bs[k].double(), padding=k // 2, groups=c).permute(0, 2, 3, 1)
        got = hip().dwconv_cl(x.to(DEV), *hip().dw_pack(ws[k].to(DEV), bs[k].to(DEV)), 0)
        np.testing.assert_allclose(got.cpu().double().numpy(), want.numpy(), rtol=tol, atol=tol)
    acc = xn
    for k in (3, 5, 7):
        acc = acc + F.conv2d(xn, ws[k].double(), bs[k].double(), padding=k // 2, groups=c)
    want = F.gelu(acc).permute(0, 2, 3, 1)
    g = lambda t: t.to(DEV)
    got = hip().dwconv_cl(g(x), *hip().dw_pack(g(ws[7]), g(bs[7]), g(ws[3]), g(bs[3]), g(ws[5]), g(bs[5])), 2)
    np.testing.assert_allclose(got.cpu().double().numpy(), want.numpy(), rtol=tol, atol=tol)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cfg", [(2, 12, 16), (1, 24, 40), (1, 48, 64), (1, 96, 128), (1, 16, 8), (1, 112, 72), (1, 192, 64), (2, 20, 200)])
def test_dct_split_cl(dtype, cfg):
    b, n, c = cfg
    x = torch.randn(b, n, n, c).to(dtype)
    w = oo.dct_matrix(n)
    high, low = oo.dct2d_split(x.double().permute(0, 3, 1, 2), w.double(), w.double())
    gh, gl = hip().dct_split_cl(x.to(DEV), w.to(DEV), w.to(DEV))
    tol = 2e-5 if dtype == torch.float32 else 3e-2
    np.testing.assert_allclose(gh.cpu().double().numpy(), high.permute(0, 2, 3, 1).numpy(), rtol=tol, atol=tol)
    np.testing.assert_allclose(gl.cpu().double().numpy(), low.permute(0, 2, 3, 1).numpy(), rtol=tol, atol=tol)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("mnk", [(300, 40, 32), (1000, 264, 256), (144, 128, 1024), (37, 5, 16), (513, 130, 72), (64, 1, 128)])
def test_linear_cl(dtype, mnk):
    m, n, k = mnk
    # A = structured + random, asymmetric W: catches transposed fragment / C-layout mistakes
    x = (torch.randn(m, k) + torch.arange(k)[None, :] * 0.01).to(dtype)
    w = (torch.randn(n, k) * 0.2 + torch.arange(n)[:, None] * 0.003).to(dtype)
    bias = torch.randn(n)
    res = torch.randn(m, n).to(dtype)
    base = x.double() @ w.double().T
    tol = 2e-5 * k ** 0.5 if dtype == torch.float32 else 2e-2
    got = hip().linear_cl(x.to(DEV), w.to(DEV), None, None, 0, torch.float32)
    np.testing.assert_allclose(got.cpu().double().numpy(), base.numpy(), rtol=1e-4 if dtype == torch.float32 else 1e-3,
                               atol=tol if dtype == torch.float32 else 1e-3 * float(base.abs().max()))
    want = F.gelu(base + bias.double()) + res.double()
    got = hip().linear_cl(x.to(DEV), w.to(DEV), bias.to(DEV), res.to(DEV), 2)
    assert got.dtype == dtype
    np.testing.assert_allclose(got.cpu().double().numpy(), want.numpy(), rtol=tol, atol=tol * max(1.0, float(want.abs().max())))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("mnk", [(576, 512, 1024), (2304, 512, 2048), (600, 200, 4096), (2304, 1024, 512), (1000, 520, 1024),
                                 (2304, 136, 1024), (70, 64, 1280)])
def test_linear_cl_small_grid_long_k(dtype, mnk):
    """The 24x24 / 12x12 stages' GEMMs (about one tile per CU, K up to 4096; deep prefetch ring): fp64 on the same
    16-bit inputs, ragged M / N included, bitwise reproducible, and the two-source A operand on the same path."""
    m, n, k = mnk
    H = hip()
    g = torch.Generator().manual_seed(m + n + k)
    x = (torch.randn(m, k, generator=g) + torch.arange(k)[None, :] * 0.002).to(dtype).to(DEV)
    w = (torch.randn(n, k, generator=g) * k ** -0.5).to(dtype).to(DEV)
    bias = torch.randn(n, generator=g).to(DEV)
    res = torch.randn(m, n, generator=g).to(dtype).to(DEV)
    base = x.double() @ w.double().T
    got = H.linear_cl(x, w, None, None, 0, torch.float32)
    np.testing.assert_allclose(got.cpu().double().numpy(), base.cpu().numpy(), rtol=1e-3, atol=1e-3 * float(base.abs().max()))
    assert torch.equal(got, H.linear_cl(x, w, None, None, 0, torch.float32))
    want = F.gelu(base + bias.double()) + res.double()
    got = H.linear_cl(x, w, bias, res, 2)
    np.testing.assert_allclose(got.cpu().double().numpy(), want.cpu().numpy(), rtol=2e-2, atol=2e-2 * max(1.0, float(want.abs().max())))
    k1 = 64 * ((k // 64) // 3)                                   # two-source A operand through the same path
    assert torch.equal(H.linear2_cl(x[:, :k1].contiguous(), x[:, k1:].contiguous(), w, bias, res, 2), got)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("mnk", [(9216, 512, 128), (36864, 128, 256), (8200, 520, 256), (9216, 512, 320)])
def test_linear_cl_short_k_on_two_lds_stages(dtype, mnk):
    """Grids of >= 1024 tiles with K <= 256 run linear_dma_kernel on 2 LDS stages (5 workgroups per CU): fp64 on the same
    16-bit inputs, ragged M / N included, and bit-identical to the 3-stage form (TRAMBA_TUNE_GEMM_TILE 14) -- the K loop
    adds the same products in the same order whatever the ring depth; K = 320 stays on 3 stages either way.  Also the
    dual-output (training fc1) form."""
    m, n, k = mnk
    H = hip()
    g = torch.Generator().manual_seed(m + n + k)
    x = (torch.randn(m, k, generator=g) + torch.arange(k)[None, :] * 0.002).to(dtype).to(DEV)
    w = (torch.randn(n, k, generator=g) * k ** -0.5).to(dtype).to(DEV)
    bias = torch.randn(n, generator=g).to(DEV)
    res = torch.randn(m, n, generator=g).to(dtype).to(DEV)
    want = F.gelu(x.double() @ w.double().T + bias.double()) + res.double()
    got = H.linear_cl(x, w, bias, res, 2)
    np.testing.assert_allclose(got.cpu().double().numpy(), want.cpu().numpy(), rtol=2e-2, atol=2e-2 * max(1.0, float(want.abs().max())))
    H.tune_set(H.TUNE_GEMM_TILE, 14)
    try:
        assert torch.equal(H.linear_cl(x, w, bias, res, 2), got)
    finally:
        H.tune_set(H.TUNE_GEMM_TILE, 0)
    if n % 8 == 0 and H.linear_dual_ok(x, w):
        pre, act = H.linear_dual_cl(x, w, bias, 2)
        assert torch.equal(act, H.linear_cl(x, w, bias, None, 2)) and torch.equal(pre, H.linear_cl(x, w, bias, None, 0))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("form", [16, 17])
@pytest.mark.parametrize("mnk", [(2304, 512, 1024), (2304, 512, 2048), (2304, 2048, 512), (4608, 512, 2048), (2300, 520, 640),
                                 (100, 64, 576), (96, 72, 64), (576, 1024, 2048), (36864, 512, 128), (9216, 256, 192)])
def test_linear_cl_producer_consumer_form(dtype, form, mnk):
    """linear_pc_kernel (r04: 512-thread workgroups, waves 0-3 multiply, waves 4-7 fetch by LDS-DMA; 3 / 4 LDS stages =
    TRAMBA_TUNE_GEMM_TILE 16 / 17): fp64 on the same 16-bit inputs, ragged M / N and every K-loop tail (K / 64 = 1, 2, 3, 8, 9,
    10, 16, 32), and BIT-identical to the r03 kernel (form 18: the same products added in the same order) for the plain, fp32-out,
    LayerNorm-folded and dual-output launches; run twice (bitwise run to run)."""
    m, n, k = mnk
    H = hip()
    g = torch.Generator().manual_seed(m + n + k)
    x = (torch.randn(m, k, generator=g) + torch.arange(k)[None, :] * 0.002).to(dtype).to(DEV)
    w = (torch.randn(n, k, generator=g) * k ** -0.5).to(dtype).to(DEV)
    bias = torch.randn(n, generator=g).to(DEV)
    res = torch.randn(m, n, generator=g).to(dtype).to(DEV)
    want = F.gelu(x.double() @ w.double().T + bias.double()) + res.double()
    colsum = w.float().sum(dim=1).contiguous()

    def run():
        out = [H.linear_cl(x, w, bias, res, 2), H.linear_cl(x, w, None, None, 0, out_dtype=torch.float32)]
        if n % 8 == 0:
            out.append(H.linear_ln_cl(x, w, colsum, bias, 1e-5, res, 2))
        if H.linear_dual_ok(x, w):
            out += list(H.linear_dual_cl(x, w, bias, 2))
        return out

    try:
        H.tune_set(H.TUNE_GEMM_TILE, 18)
        old = run()
        H.tune_set(H.TUNE_GEMM_TILE, form)
        got = run()
        again = run()
    finally:
        H.tune_set(H.TUNE_GEMM_TILE, 0)
    np.testing.assert_allclose(got[0].cpu().double().numpy(), want.cpu().numpy(), rtol=2e-2,
                               atol=2e-2 * max(1.0, float(want.abs().max())))
    if n % 8 == 0:      # LayerNorm folded in: LN(x) W^T with gamma = 1, beta = 0
        xn = torch.nn.functional.layer_norm(x.double(), (k,))
        want_ln = F.gelu(xn @ w.double().T + bias.double()) + res.double()
        np.testing.assert_allclose(got[2].cpu().double().numpy(), want_ln.cpu().numpy(), rtol=3e-2,
                                   atol=3e-2 * max(1.0, float(want_ln.abs().max())))
    assert len(got) == len(old) >= 2
    for a, b, c in zip(got, old, again):
        assert torch.equal(a, b) and torch.equal(a, c)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("mnk", [(36864, 512, 128), (36864, 256, 128), (36864, 128, 256), (9216, 1024, 256), (9216, 512, 256),
                                 (4100, 256, 128), (9216, 384, 128), (33, 128, 256), (73728, 128, 128), (2304, 512, 256)])
def test_linear_cl_weight_stationary_form(dtype, mnk):
    """linear_ws_kernel (r04: persistent workgroups, the weights of a 128 NCW-column panel held in registers as MFMA fragments,
    32-row activation tiles on an LDS-DMA ring, epilogue straight from the accumulators; TRAMBA_TUNE_GEMM_TILE 19 forces it
    wherever it can run): fp64 on the same 16-bit inputs, NCW = 4 / 2 / 1, several panels, ragged M (4100 = 128 tiles + 4 rows,
    33 = one tile + 1 row), work lists of 1 .. 5 tiles per workgroup.  The plain (bias + GELU + residual), SiLU, bias-free and
    dual-output launches are BIT-identical to the r03 kernel (form 18: same products, same order, same epilogue formulas);
    the LayerNorm-folded launch sums its row statistics in another order (from the MFMA fragments) and is held to fp64
    within the GEMM tolerance and to the r03 kernel within 2 ulp of the output dtype; twice = bitwise run to run."""
    m, n, k = mnk
    H = hip()
    g = torch.Generator().manual_seed(m + n + k)
    x = (torch.randn(m, k, generator=g) + torch.arange(k)[None, :] * 0.002).to(dtype).to(DEV)
    w = (torch.randn(n, k, generator=g) * k ** -0.5).to(dtype).to(DEV)
    bias = torch.randn(n, generator=g).to(DEV)
    res = torch.randn(m, n, generator=g).to(dtype).to(DEV)
    colsum = w.float().sum(dim=1).contiguous()

    def run():
        out = [H.linear_cl(x, w, bias, res, 2), H.linear_cl(x, w, bias, None, 1), H.linear_cl(x, w, None, None, 0),
               H.linear_cl(x, w, bias, res, 0)]
        if H.linear_dual_ok(x, w):
            out += list(H.linear_dual_cl(x, w, bias, 2))
        out.append(H.linear_ln_cl(x, w, colsum, bias, 1e-5, None, 2))
        out.append(H.linear_ln_cl(x, w, colsum, bias, 1e-5, res, 0))
        return out

    try:
        H.tune_set(H.TUNE_GEMM_TILE, 18)
        old = run()
        H.tune_set(H.TUNE_GEMM_TILE, 19)
        got = run()
        again = run()
    finally:
        H.tune_set(H.TUNE_GEMM_TILE, 0)
    torch.cuda.synchronize()
    H.device_error()
    want = F.gelu(x.double() @ w.double().T + bias.double()) + res.double()
    np.testing.assert_allclose(got[0].cpu().double().numpy(), want.cpu().numpy(), rtol=2e-2,
                               atol=2e-2 * max(1.0, float(want.abs().max())))
    xn = torch.nn.functional.layer_norm(x.double(), (k,))
    want_ln = F.gelu(xn @ w.double().T + bias.double())
    np.testing.assert_allclose(got[-2].cpu().double().numpy(), want_ln.cpu().numpy(), rtol=3e-2,
                               atol=3e-2 * max(1.0, float(want_ln.abs().max())))
    want_ln2 = xn @ w.double().T + bias.double() + res.double()
    np.testing.assert_allclose(got[-1].cpu().double().numpy(), want_ln2.cpu().numpy(), rtol=3e-2,
                               atol=3e-2 * max(1.0, float(want_ln2.abs().max())))
    assert len(got) == len(old)
    for i, (a, b, c) in enumerate(zip(got, old, again)):
        assert torch.equal(a, c), i
        if i < len(got) - 2:
            assert torch.equal(a, b), (i, float((a.float() - b.float()).abs().max()))
        else:       # LayerNorm folded in: statistics summed in another order
            d = (a.float() - b.float()).abs()
            ulp = (2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10) * b.float().abs().clamp_min(1.0)
            assert bool((d <= 2 * ulp).all()), (i, float(d.max()))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("mnk", [(2304, 512, 1024), (2304, 2048, 512), (4608, 512, 2048), (2300, 520, 640), (100, 64, 576), (96, 72, 64),
                                 (576, 1024, 2048)])
def test_linear_cl_on_96_row_tiles(dtype, mnk):
    """linear_dma96_kernel (96 x 64 tiles: 3 compute waves + a loader wave; a measurement form behind TRAMBA_TUNE_GEMM_TILE 15 --
    it loses to the 64 x 64 form): fp64 on the same 16-bit inputs, ragged M / N and every K-loop tail
    (K / 64 = 1, 8, 9, 10, 16, 32), and BIT-identical to the library's own form (the same products added in the same
    order), for the plain and the dual-output launches."""
    m, n, k = mnk
    H = hip()
    g = torch.Generator().manual_seed(m + n + k)
    x = (torch.randn(m, k, generator=g) + torch.arange(k)[None, :] * 0.002).to(dtype).to(DEV)
    w = (torch.randn(n, k, generator=g) * k ** -0.5).to(dtype).to(DEV)
    bias = torch.randn(n, generator=g).to(DEV)
    res = torch.randn(m, n, generator=g).to(dtype).to(DEV)
    want = F.gelu(x.double() @ w.double().T + bias.double()) + res.double()

    def run():
        out = [H.linear_cl(x, w, bias, res, 2), H.linear_cl(x, w, None, None, 0, out_dtype=torch.float32)]
        if H.linear_dual_ok(x, w):
            out += list(H.linear_dual_cl(x, w, bias, 2))
        return out

    own = run()
    try:
        H.tune_set(H.TUNE_GEMM_TILE, 15)
        got96 = run()
    finally:
        H.tune_set(H.TUNE_GEMM_TILE, 0)
    np.testing.assert_allclose(got96[0].cpu().double().numpy(), want.cpu().numpy(), rtol=2e-2,
                               atol=2e-2 * max(1.0, float(want.abs().max())))
    assert len(got96) == len(own) >= 2
    for a, b in zip(got96, own):
        assert torch.equal(a, b)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("fam", ["raster", "helix", "window", "dilation"])
@pytest.mark.parametrize("cfg", [(2, 12, 32, 2), (1, 24, 64, 4), (1, 16, 40, 3), (1, 48, 128, 8), (1, 12, 96, 40), (1, 24, 32, 64), (2, 96, 64, 8),
                                 (1, 12, 1024, 16), (1, 8, 2048, 8)])
def test_ss2d_fused_core(dtype, fam, cfg):
    """fused channels-last scan + merge/LayerNorm/GELU against the oracle's NCHW composition."""
    b, h, d, r = cfg
    k = 8 if fam == "helix" else 4
    g = torch.Generator().manual_seed(h * d + k)
    x = torch.randn(b, d, h, h, generator=g).to(dtype)
    wx = (torch.randn(k, r + 2, d, generator=g) * d ** -0.5).to(dtype)
    wdt = torch.randn(k, d, r, generator=g) * r ** -0.5
    dtb = torch.randn(k, d, generator=g) * 0.5 - 2.0
    a_logs = torch.log(0.5 + torch.rand(k * d, 1, generator=g))
    ds = 1 + 0.1 * torch.randn(k * d, generator=g)
    lw, lb = 1 + 0.1 * torch.randn(d, generator=g), 0.1 * torch.randn(d, generator=g)
    y = oo.ss2d_core(x.double(), wx.double(), wdt.double(), dtb.double(), a_logs.double(), ds.double(), fam)
    want = F.gelu(oo.layernorm2d(y, lw.double(), lb.double())).permute(0, 2, 3, 1)

    H = hip()
    dev = torch.device(DEV)
    order = H.scan_order(fam, h, h, dev)
    xc = x.permute(0, 2, 3, 1).contiguous().view(b, h * h, d).to(dev)
    xdbl = H.linear_cl(xc, H.pad_x_proj_weight(wx.to(dev)), out_dtype=torch.float32)
    for ys_dtype in ([torch.float32] if dtype == torch.float32 else [torch.float32, dtype]):
        for segmented in (True, False):  # library-chosen form with workspace / chained form without
            ys = H.ss2d_scan_cl(xc, xdbl, order, wdt.to(dev), dtb.reshape(-1).to(dev),
                                (-torch.exp(a_logs)).reshape(-1).to(dev), ds.to(dev), ys_dtype, segmented=segmented)
            out = H.ss2d_merge_norm_cl(ys, order, lw.to(dev), lb.to(dev), 1e-5, 2, dtype)
            tol = 2e-4 if dtype == torch.float32 else 4e-2
            np.testing.assert_allclose(out.view(b, h, h, d).cpu().double().numpy(), want.numpy(), rtol=tol, atol=tol)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("fam,b,h,d,r", [("helix", 4, 96, 256, 8), ("raster", 4, 96, 256, 8), ("helix", 4, 48, 512, 16),
                                         ("window", 8, 48, 512, 16), ("helix", 2, 96, 256, 8)])
def test_ss2d_lds_dma_scan_at_the_benchmarked_launches(dtype, fam, b, h, d, r):
    """The launches bench.py's `roofline` object times -- ss2d_scan_dma_kernel + the deep / streaming merge at B=4, 96x96,
    D=256, K=8 (BASELINE configs[1]), its K=4 sibling, the 48x48 rank-16 form and the 8-wave workgroups of a smaller batch --
    DIRECTLY against the fp64 oracle (vmamba.py:230-257 composed with csms6s.py:161-216), 16-bit `ys` as the model runs it.
    test_ss2d_fused_core's maps stay below the 2048-wave rule that selects this kernel."""
    k = 8 if fam == "helix" else 4
    g = torch.Generator().manual_seed(h * d + k + b)
    x = torch.randn(b, d, h, h, generator=g).to(dtype)
    wx = (torch.randn(k, r + 2, d, generator=g) * d ** -0.5).to(dtype)
    wdt = torch.randn(k, d, r, generator=g) * r ** -0.5
    dtb = torch.randn(k, d, generator=g) * 0.5 - 2.0
    a_logs = torch.log(0.5 + torch.rand(k * d, 1, generator=g))
    ds = 1 + 0.1 * torch.randn(k * d, generator=g)
    lw, lb = 1 + 0.1 * torch.randn(d, generator=g), 0.1 * torch.randn(d, generator=g)
    # the oracle's per-direction outputs (B,K,L,D) before the merge, then its merge + out_norm + GELU
    ys_want = oo.ss2d_core(x.double(), wx.double(), wdt.double(), dtb.double(), a_logs.double(), ds.double(), fam, merge=False)
    y = oo.cross_merge(ys_want.permute(0, 1, 3, 2), fam, h, h).reshape(b, d, h, h)
    want = F.gelu(oo.layernorm2d(y, lw.double(), lb.double())).permute(0, 2, 3, 1)
    H = hip()
    dev = torch.device(DEV)
    order = H.scan_order(fam, h, h, dev)
    xc = x.permute(0, 2, 3, 1).contiguous().view(b, h * h, d).to(dev)
    xdbl = H.linear_cl(xc, H.pad_x_proj_weight(wx.to(dev)), out_dtype=torch.float32)
    ys = H.ss2d_scan_cl(xc, xdbl, order, wdt.to(dev), dtb.reshape(-1).to(dev), (-torch.exp(a_logs)).reshape(-1).to(dev),
                        ds.to(dev), dtype)
    err = (ys.cpu().double() - ys_want).abs()
    scale = float(ys_want.abs().max())
    assert float(err.max()) <= 2e-2 * scale, (float(err.max()), scale)          # 16-bit rounding of ys: 2^-9 relative
    assert float(err.pow(2).mean().sqrt()) <= 3e-3 * float(ys_want.pow(2).mean().sqrt())
    out = H.ss2d_merge_norm_cl(ys, order, lw.to(dev), lb.to(dev), 1e-5, 2, dtype)
    np.testing.assert_allclose(out.view(b, h, h, d).cpu().double().numpy(), want.numpy(), rtol=4e-2, atol=4e-2)
    assert torch.equal(ys, H.ss2d_scan_cl(xc, xdbl, order, wdt.to(dev), dtb.reshape(-1).to(dev),
                                          (-torch.exp(a_logs)).reshape(-1).to(dev), ds.to(dev), dtype))   # run-to-run bitwise


@pytest.mark.parametrize("form", [0, 1])      # the library's choice for this shape (chained on LDS-DMA) / the register ring: both carry by mailbox
def test_scan_mailbox_timeout_is_reported_not_returned_as_a_result(form):
    """a starved carry-mailbox poll used to end the launch with NaN in `ys` and rc 0.  With the poll budget
    forced to 1 (TRAMBA_TUNE_MAILBOX_SKIP) the waves that find their predecessor's mailbox empty give up at once: the launch
    still ends, its output holds NaN, and the device error word makes the next library call after it return TRAMBA_ERR_HIP
    with the cause in tramba_last_error(); the word is cleared by the report and the same launch at the default budget is
    clean."""
    H = hip()
    dev = torch.device(DEV)
    b, h, d, r, k, dtype = 4, 96, 256, 8, 8, torch.bfloat16     # the Helix launch of BASELINE configs[1]
    g = torch.Generator().manual_seed(77)
    order = H.scan_order("helix", h, h, dev)
    xc = torch.randn(b, h * h, d, generator=g).to(dtype).to(dev)
    wx = (torch.randn(k, r + 2, d, generator=g) * d ** -0.5).to(dtype)
    xdbl = H.linear_cl(xc, H.pad_x_proj_weight(wx.to(dev)), out_dtype=torch.float32)
    args = (xc, xdbl, order, (torch.randn(k, d, r, generator=g) * r ** -0.5).to(dev),
            (torch.randn(k * d, generator=g) * 0.5 - 2.0).to(dev), (-0.5 - torch.rand(k * d, generator=g)).to(dev),
            torch.ones(k * d).to(dev), dtype)
    H.tune_set(H.TUNE_SCAN_FORM, form)
    try:
        good = H.ss2d_scan_cl(*args)
        torch.cuda.synchronize()
        H.device_error()                                   # nothing pending
        assert torch.isfinite(good).all()
        H.tune_set(H.TUNE_MAILBOX_SKIP, 3)
        try:
            bad = H.ss2d_scan_cl(*args)                    # the launch itself is issued without complaint ...
            torch.cuda.synchronize()
        finally:
            H.tune_set(H.TUNE_MAILBOX_SKIP, 0)
        assert not torch.isfinite(bad).all()               # ... its output is poisoned, not plausible ...
        with pytest.raises(H.TrambaHipError, match="mailbox"):
            H.layernorm_cl(xc, torch.ones(d, device=dev), torch.zeros(d, device=dev))   # ... and the NEXT call says so
        H.device_error()                                   # reported once, then cleared
        again = H.ss2d_scan_cl(*args)
        torch.cuda.synchronize()
        H.device_error()
        assert torch.equal(again, good)
    finally:
        H.tune_set(H.TUNE_SCAN_FORM, 0)
        H.tune_set(H.TUNE_MAILBOX_SKIP, 0)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cfg", [(2, 12, 64, 40), (1, 24, 128, 256), (1, 13, 64, 64), (1, 48, 256, 130)])
def test_conv3x3s2_cl(dtype, cfg):
    """implicit-GEMM 3x3 / stride 2 / pad 1 conv (patch_embed[5], downsample) against F.conv2d in fp64."""
    b, h, cin, cout = cfg
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(b, h, h, cin, generator=g).to(dtype)
    w = (torch.randn(cout, cin, 3, 3, generator=g) * (9 * cin) ** -0.5).to(dtype)
    bias = torch.randn(cout, generator=g)
    want = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), bias.double(), stride=2, padding=1).permute(0, 2, 3, 1)
    wk = w.permute(0, 2, 3, 1).reshape(cout, -1).contiguous()
    got = hip().conv3x3s2_cl(x.to(DEV), wk.to(DEV), bias.to(DEV))
    assert got.shape == want.shape
    np.testing.assert_allclose(got.cpu().double().numpy(), want.numpy(), rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cfg", [(2, 24), (1, 37), (1, 96)])
def test_stem_conv_ln_gelu(dtype, cfg):
    """fused patch_embed[0..4]: conv 3->64 s2 + LayerNorm2d + GELU, NCHW fp32 image in."""
    b, h = cfg
    g = torch.Generator().manual_seed(h)
    img = torch.randn(b, 3, h, h, generator=g)
    w = torch.randn(64, 3, 3, 3, generator=g) * 27 ** -0.5
    bias, lw, lb = 0.1 * torch.randn(64, generator=g), 1 + 0.1 * torch.randn(64, generator=g), 0.1 * torch.randn(64, generator=g)
    y = F.conv2d(img.double(), w.double(), bias.double(), stride=2, padding=1)
    want = F.gelu(oo.layernorm2d(y, lw.double(), lb.double())).permute(0, 2, 3, 1)
    got = hip().stem_conv_ln_gelu(img.to(DEV), w.to(DEV), bias.to(DEV), lw.to(DEV), lb.to(DEV), 1e-5, dtype)
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    np.testing.assert_allclose(got.cpu().double().numpy(), want.numpy(), rtol=tol, atol=tol)


def test_two_stream_concurrency_is_bitwise_stable():
    """Kernels of two HIP streams running concurrently must not change each other's results (the decoder's guide
    branch runs beside the encoder, tramba_amd/models.py).  Regression: a 16-byte buffer store with an SGPR
    offset lost its data to the next VALU write when the memory system was busy (fp32 streaming merge)."""
    H = hip()
    dev = torch.device(DEV)
    g = torch.Generator().manual_seed(5)

    def scan_ops(d, r, fam, dtype):
        order = H.scan_order(fam, 96, 96, dev)
        k, l = order.k, 96 * 96
        x = torch.randn(1, l, d, generator=g).to(dev, dtype)
        wx = (torch.randn(k, r + 2, d, generator=g) * d ** -0.5).to(dev, dtype)
        xdbl = H.linear_cl(x, H.pad_x_proj_weight(wx), out_dtype=torch.float32)
        dt_w = (torch.randn(k, d, r, generator=g) * r ** -0.5).to(dev)
        dt_b = (torch.randn(k * d, generator=g) * 0.5 - 3).to(dev)
        a = -torch.rand(k * d, generator=g).to(dev) - 0.5
        ds = torch.ones(k * d, device=dev)
        lw, lb = torch.ones(d, device=dev), torch.zeros(d, device=dev)
        ys0 = H.ss2d_scan_cl(x, xdbl, order, dt_w, dt_b, a, ds, torch.float32)
        return {"scan": lambda: H.ss2d_scan_cl(x, xdbl, order, dt_w, dt_b, a, ds, torch.float32),
                "merge": lambda: H.ss2d_merge_norm_cl(ys0, order, lw, lb, 1e-5, 2, dtype)}

    xs = torch.randn(1, 192, 192, 128, generator=g).to(dev)
    ws = (torch.randn(256, 128, generator=g) * 0.1).to(dev)
    side_ops = [lambda: H.linear_cl(xs, ws, None, None, 2), scan_ops(256, 8, "window", torch.float32)["merge"]]
    side = torch.cuda.Stream()
    for dtype in (torch.float32, torch.bfloat16):
        for name, op in scan_ops(512, 16, "raster", dtype).items():
            ref = op().clone()
            torch.cuda.synchronize()
            for sop in side_ops:
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    keep = [sop() for _ in range(30)]
                outs = [op() for _ in range(10)]
                torch.cuda.synchronize()
                assert all(torch.equal(o, ref) for o in outs), (name, dtype)
                del keep


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("rc", [(1000, 128), (77, 512), (4096, 256), (13, 1024), (300, 40)])
def test_rowdot_cl(dtype, rc):
    """nn.Conv2d(C, 1, 1) on channels-last rows (decoder seg heads, Trambav6.py:62,67)."""
    rows, c = rc
    g = torch.Generator().manual_seed(rows + c)
    x = torch.randn(rows, c, generator=g).to(dtype)
    w = torch.randn(c, generator=g) * c ** -0.5
    got = hip().rowdot_cl(x.to(DEV), w.to(DEV), 0.37)
    want = x.double() @ w.double() + 0.37
    np.testing.assert_allclose(got.cpu().double().numpy(), want.numpy(), rtol=1e-5, atol=1e-5 if dtype == torch.float32 else 1e-4)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cfg", [(2, 6, 128, 4), (1, 5, 64, 2), (1, 12, 128, 4)])
def test_shuffle_norm_head_cl(dtype, cfg):
    """FinalPatchExpand_X4's shuffle + LayerNorm fused with the 1x1 head == shuffle_norm_cl followed by rowdot."""
    b, h, c, p = cfg
    H = hip()
    g = torch.Generator().manual_seed(h * c + p)
    x = torch.randn(b, h, h, p * p * c, generator=g).to(dtype).to(DEV)
    lw, lb = (1 + 0.1 * torch.randn(c, generator=g)).to(DEV), (0.1 * torch.randn(c, generator=g)).to(DEV)
    hw = (torch.randn(c, generator=g) * c ** -0.5).to(DEV)
    got = H.shuffle_norm_head_cl(x, lw, lb, hw, -0.21, p)
    mid = H.shuffle_norm_cl(x, lw, lb, p)                      # (B, H*P, W*P, C), rounded to dtype like the fused path
    want = H.rowdot_cl(mid, hw, -0.21)
    assert got.shape == (b, h * p, h * p)
    np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=1e-5)
    # and against the oracle's composition in fp64
    y = x.double().cpu().view(b, h, h, p, p, c).permute(0, 1, 3, 2, 4, 5).reshape(b, h * p, h * p, c)
    y = F.layer_norm(y, (c,), lw.double().cpu(), lb.double().cpu(), 1e-5)
    ref = y @ hw.double().cpu() - 0.21
    np.testing.assert_allclose(got.cpu().double().numpy(), ref.numpy(), rtol=2e-2 if dtype != torch.float32 else 1e-4,
                               atol=2e-2 if dtype != torch.float32 else 1e-4)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cfg", [(2, 6, 7, 128, 4), (1, 5, 5, 64, 2), (4, 96, 96, 128, 4), (1, 13, 3, 192, 1)])
def test_expand_norm_head_cl(dtype, cfg):
    """The whole final decoder stage in one kernel == expand GEMM -> pixel shuffle -> LayerNorm(128) -> 1x1 head
    (Trambav6.py:132-137), in fp64 on the same 16-bit inputs; and close to the three-kernel product path."""
    b, h, wd, cin, p = cfg
    H = hip()
    g = torch.Generator().manual_seed(h * cin + p)
    x = torch.randn(b, h, wd, cin, generator=g).to(dtype).to(DEV)
    w = (torch.randn(p * p * 128, cin, generator=g) * cin ** -0.5).to(dtype).to(DEV)
    lw, lb = (1 + 0.1 * torch.randn(128, generator=g)).to(DEV), (0.1 * torch.randn(128, generator=g)).to(DEV)
    hw = (torch.randn(128, generator=g) * 128 ** -0.5).to(DEV)
    got = H.expand_norm_head_cl(x, w, lw, lb, hw, 0.4, p)
    assert got.shape == (b, h * p, wd * p)
    xe = x.double() @ w.double().T
    y = xe.view(b, h, wd, p, p, 128).permute(0, 1, 3, 2, 4, 5).reshape(b, h * p, wd * p, 128)
    y = F.layer_norm(y, (128,), lw.double(), lb.double(), 1e-5)
    ref = y @ hw.double() + 0.4
    np.testing.assert_allclose(got.cpu().double().numpy(), ref.cpu().numpy(), rtol=2e-4, atol=2e-4)
    unfused = H.shuffle_norm_head_cl(H.linear_cl(x, w), lw, lb, hw, 0.4, p)
    np.testing.assert_allclose(got.cpu().numpy(), unfused.cpu().numpy(), rtol=3e-2, atol=3e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cfg", [(2, 12, 64, 3), (1, 24, 130, 7), (2, 9, 32, 5), (1, 48, 256, 7)])
def test_dwconv_training_path_matches_autograd(dtype, cfg):
    """_DwConvCL (HIP forward, flipped-tap input gradient, atomic weight gradient) against F.conv2d autograd in fp64."""
    from tramba_amd import modules as M
    b, h, c, ks = cfg
    g = torch.Generator().manual_seed(h * c + ks)
    conv = torch.nn.Conv2d(c, c, ks, padding=ks // 2, groups=c, bias=True)
    x = torch.randn(b, h, h, c, generator=g).to(dtype)
    gy = torch.randn(b, h, h, c, generator=g).to(dtype)
    # reference in fp64 on the (rounded) inputs
    xr = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    cr = torch.nn.Conv2d(c, c, ks, padding=ks // 2, groups=c, bias=True).double()
    cr.load_state_dict({k: v.double() for k, v in conv.state_dict().items()})
    yr = cr(xr)
    yr.backward(gy.double().permute(0, 3, 1, 2))
    conv = conv.to(DEV)
    xg = x.to(DEV).requires_grad_(True)
    y = M._dwconv_train_cl(xg, conv)
    y.backward(gy.to(DEV))
    tol = 2e-5 if dtype == torch.float32 else 3e-2
    np.testing.assert_allclose(y.detach().cpu().double().numpy(), yr.detach().permute(0, 2, 3, 1).numpy(), rtol=tol, atol=tol)
    np.testing.assert_allclose(xg.grad.cpu().double().numpy(), xr.grad.permute(0, 2, 3, 1).numpy(), rtol=tol, atol=tol)
    wtol = 1e-4 if dtype == torch.float32 else 3e-2
    scale = float(cr.weight.grad.abs().max())
    np.testing.assert_allclose(conv.weight.grad.cpu().double().numpy(), cr.weight.grad.numpy(), rtol=wtol, atol=wtol * scale)
    np.testing.assert_allclose(conv.bias.grad.cpu().double().numpy(), cr.bias.grad.numpy(), rtol=wtol,
                               atol=wtol * float(cr.bias.grad.abs().max()))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32, torch.float16])
@pytest.mark.parametrize("cfg", [(2, 96, 96, 512, 0), (1, 48, 48, 1024, 0), (2, 24, 24, 2048, 0), (1, 13, 50, 130, 0), (3, 7, 5, 6, 0),
                                 (1, 96, 96, 128, 5), (1, 30, 22, 64, 7), (2, 24, 24, 256, 100), (2, 13, 50, 136, 0), (1, 9, 6, 8, 0)])
def test_dwconv7_marching_kernel_is_bit_identical_to_the_row_kernel(dtype, cfg):
    """dwconv7_march_kernel (r04: a lane = 2 channels x 4 columns walking a band of rows, every input row converted once, the 49
    tap pairs in registers) does the products of dwconv4_cl_kernel<7> in the same order: the plain, the dual-store (training
    forward) and the flipped-tap (input gradient) launches are BIT-identical to that kernel (TRAMBA_TUNE_DW_FORM 1) at the
    decoder's three DWMSMlp shapes, on ragged maps (13 x 50 with 136 channels: a channel tile of 8; channel counts that are
    not a multiple of 8 -- 130, 6 -- stay with the row kernel) and with forced band heights
    (5, 7, 100 rows: bands that end inside the map, a single band); fp64 on the same inputs holds both."""
    b, h, w, c, rows = cfg
    if dtype == torch.float32 and h * w * c > 96 * 96 * 128:
        pytest.skip("fp32 covered on the small maps")
    H = hip()
    g = torch.Generator().manual_seed(h * w + c)
    x = torch.randn(b, h, w, c, generator=g).to(dtype).to(DEV)
    wt = (0.15 * torch.randn(49, c, generator=g)).to(DEV)
    bt = (0.1 * torch.randn(c, generator=g)).to(DEV)

    def run():
        return [H.dwconv_cl(x, wt, bt, 2), H.dwconv_cl(x, wt, bt, 0), *H.dwconv_dual_cl(x, wt, bt, 2, True, False),
                H.dwconv_dual_cl(x, wt, bt, 0, False, True)[1], H.dwconv_cl(x, wt, bt, 1)]

    try:
        H.tune_set(H.TUNE_DW_FORM, 1)
        old = run()
        H.tune_set(H.TUNE_DW_FORM, 0)
        H.tune_set(H.TUNE_DW_ROWS, rows)
        new = run()
    finally:
        H.tune_set(H.TUNE_DW_FORM, 0)
        H.tune_set(H.TUNE_DW_ROWS, 0)
    torch.cuda.synchronize()
    for i, (o, n) in enumerate(zip(old, new)):
        assert torch.equal(o, n), f"launch {i}: {int((o != n).sum())} of {o.numel()} values differ"
    xn = x.double().permute(0, 3, 1, 2)
    k = wt.double().view(7, 7, c).permute(2, 0, 1).unsqueeze(1)
    want = F.conv2d(xn, k, bt.double(), padding=3, groups=c).permute(0, 2, 3, 1)
    tol = 1e-5 if dtype == torch.float32 else (4e-3 if dtype == torch.float16 else 3e-2)
    np.testing.assert_allclose(new[1].cpu().double().numpy(), want.cpu().numpy(), rtol=tol, atol=tol)
    wantf = F.conv2d(xn, k.flip(2, 3), bt.double(), padding=3, groups=c).permute(0, 2, 3, 1)
    np.testing.assert_allclose(new[4].cpu().double().numpy(), wantf.cpu().numpy(), rtol=tol, atol=tol)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32, torch.float16])
@pytest.mark.parametrize("cfg", [(2, 96, 96, 256, 0), (2, 48, 48, 512, 0), (2, 24, 24, 1024, 0), (1, 13, 50, 130, 0), (3, 7, 5, 6, 0),
                                 (1, 40, 36, 64, 5), (1, 30, 100, 64, 7), (2, 24, 24, 128, 100), (2, 13, 50, 136, 0), (1, 9, 6, 8, 0)])
def test_dwconv7_marching_weight_gradient(dtype, cfg):
    """dwconv7_wgrad_march_kernel (r04: an x row converted once meets the seven gy rows it pairs with from a rotating register
    window) against fp64 on the same inputs and against the tap-row-outer kernel (TRAMBA_TUNE_DW_FORM 1; another summation
    order: fp32 rounding apart); ragged maps, more column groups than waves (W = 100: 4 column ranges), forced band heights;
    twice = bitwise run to run (ordered fold, no atomics)."""
    b, h, w, c, rows = cfg
    H = hip()
    g = torch.Generator().manual_seed(h * w + c + 1)
    x = torch.randn(b, h, w, c, generator=g).to(dtype).to(DEV)
    gy = torch.randn(b, h, w, c, generator=g).to(dtype).to(DEV)
    try:
        H.tune_set(H.TUNE_DW_FORM, 1)
        gw_old, gb_old = H.dwconv_wgrad_cl(x, gy, 7)
        H.tune_set(H.TUNE_DW_FORM, 0)
        H.tune_set(H.TUNE_DW_ROWS, rows)
        gw, gb = H.dwconv_wgrad_cl(x, gy, 7)
        gw2, gb2 = H.dwconv_wgrad_cl(x, gy, 7)
    finally:
        H.tune_set(H.TUNE_DW_FORM, 0)
        H.tune_set(H.TUNE_DW_ROWS, 0)
    torch.cuda.synchronize()
    assert torch.equal(gw, gw2) and torch.equal(gb, gb2)
    xp = F.pad(x.double().permute(0, 3, 1, 2), (3, 3, 3, 3))
    gd = gy.double().permute(0, 3, 1, 2)
    want = torch.stack([(xp[:, :, dy:dy + h, dx:dx + w] * gd).sum(dim=(0, 2, 3)) for dy in range(7) for dx in range(7)])
    scale = float(want.abs().max())
    np.testing.assert_allclose(gw.cpu().double().numpy(), want.cpu().numpy(), rtol=1e-4, atol=2e-5 * scale)
    np.testing.assert_allclose(gb.cpu().double().numpy(), gd.sum(dim=(0, 2, 3)).cpu().numpy(), rtol=1e-4,
                               atol=2e-5 * float(gd.sum(dim=(0, 2, 3)).abs().max()))
    np.testing.assert_allclose(gw.cpu().numpy(), gw_old.cpu().numpy(), rtol=1e-4, atol=2e-5 * scale)


def test_dwms_training_fold_matches_reference_sum():
    """h + dw3(h) + dw5(h) + dw7(h) as one folded stencil: outputs and all six parameter gradients."""
    from tramba_amd import modules as M
    c, h = 64, 12
    g = torch.Generator().manual_seed(3)
    convs = [torch.nn.Conv2d(c, c, k, padding=k // 2, groups=c, bias=True).double() for k in (3, 5, 7)]
    x = torch.randn(2, h, h, c, generator=g)
    gy = torch.randn(2, h, h, c, generator=g)
    xr = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    yr = xr + convs[0](xr) + convs[1](xr) + convs[2](xr)
    yr.backward(gy.double().permute(0, 3, 1, 2))
    dev = [torch.nn.Conv2d(c, c, k, padding=k // 2, groups=c, bias=True) for k in (3, 5, 7)]
    for d, r in zip(dev, convs):
        d.load_state_dict({k: v.float() for k, v in r.state_dict().items()})
        d.to(DEV)
    xg = x.to(DEV).requires_grad_(True)
    y = M._dwms_train_cl(xg, dev[0], dev[1], dev[2])
    y.backward(gy.to(DEV))
    np.testing.assert_allclose(y.detach().cpu().double().numpy(), yr.detach().permute(0, 2, 3, 1).numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(xg.grad.cpu().double().numpy(), xr.grad.permute(0, 2, 3, 1).numpy(), rtol=1e-4, atol=1e-4)
    for d, r in zip(dev, convs):
        np.testing.assert_allclose(d.weight.grad.cpu().double().numpy(), r.weight.grad.numpy(), rtol=1e-3,
                                   atol=1e-3 * float(r.weight.grad.abs().max()))
        np.testing.assert_allclose(d.bias.grad.cpu().double().numpy(), r.bias.grad.numpy(), rtol=1e-3,
                                   atol=1e-3 * float(r.bias.grad.abs().max()))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cfg", [(2, 3, 64, 96, 2), (8, 64, 128, 48, 2), (2, 16, 24, 20, 1), (16, 8, 8, 64, 2)])
def test_dense_conv_training_path_matches_autograd(dtype, cfg):
    """_ConvIm2colCL (channels-last; im2col around tramba_linear_cl for the forward / input gradient and tramba_wgrad_cl
    for the weight / bias gradient; fp32 through bf16 hi + lo splits) against F.conv2d autograd in fp64.  Cin = 3 takes
    the padded-K route (27 -> 32 columns), the last cases split the token range over several workgroups."""
    from tramba_amd import modules as M
    b, cin, cout, hw, stride = cfg
    g = torch.Generator().manual_seed(cin * cout + hw)
    x = torch.randn(b, cin, hw, hw, generator=g).to(dtype)
    w = (torch.randn(cout, cin, 3, 3, generator=g) * (9 * cin) ** -0.5).to(dtype)
    bias = (0.1 * torch.randn(cout, generator=g)).to(dtype)
    gy_shape = F.conv2d(x.float(), w.float(), stride=stride, padding=1).shape
    gy = torch.randn(gy_shape, generator=g).to(dtype)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, bias))
    yr = F.conv2d(xr, wr, br, stride=stride, padding=1)
    yr.backward(gy.double())
    xg = x.permute(0, 2, 3, 1).contiguous().to(DEV).requires_grad_(True)                 # channels-last (B,H,W,C)
    wg, bg = (t.float().to(DEV).requires_grad_(True) for t in (w, bias))                  # fp32 master parameters
    y = M._ConvIm2colCL.apply(xg, wg, bg, (stride, stride), (1, 1))
    y.backward(gy.permute(0, 2, 3, 1).contiguous().to(DEV))
    tol = 2e-4 if dtype == torch.float32 else 3e-2
    pairs = ((y.permute(0, 3, 1, 2), yr.detach()), (xg.grad.permute(0, 3, 1, 2), xr.grad), (wg.grad, wr.grad), (bg.grad, br.grad))
    for got, want in pairs:
        np.testing.assert_allclose(got.detach().cpu().double().numpy(), want.numpy(), rtol=tol, atol=tol * float(want.abs().max()))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rc", [(500, 128), (77, 512), (4100, 256), (33, 1024), (64, 2048), (50, 40), (100003, 128), (70001, 64),
                                (9001, 320), (301, 1536), (4608, 1024), (1153, 2048), (700, 520)])
def test_layernorm_training_path_matches_autograd(dtype, rc):
    """_LayerNormCL (HIP forward + HIP backward with atomic dgamma/dbeta) against F.layer_norm autograd in fp64."""
    from tramba_amd import modules as M
    rows, c = rc
    g = torch.Generator().manual_seed(rows + c)
    x = (torch.randn(rows, c, generator=g) * 1.5 + 0.3).to(dtype)
    gy = torch.randn(rows, c, generator=g).to(dtype)
    w = 1 + 0.2 * torch.randn(c, generator=g)
    b = 0.1 * torch.randn(c, generator=g)
    xr = x.double().requires_grad_(True)
    wr, br = w.double().requires_grad_(True), b.double().requires_grad_(True)
    yr = F.layer_norm(xr, (c,), wr, br, 1e-5)
    yr.backward(gy.double())
    xg = x.to(DEV).requires_grad_(True)
    wg, bg = w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    y = M._LayerNormCL.apply(xg, wg, bg, 1e-5)
    y.backward(gy.to(DEV))
    tol = 2e-5 if dtype == torch.float32 else 3e-2
    np.testing.assert_allclose(y.detach().cpu().double().numpy(), yr.detach().numpy(), rtol=tol, atol=tol)
    np.testing.assert_allclose(xg.grad.cpu().double().numpy(), xr.grad.numpy(), rtol=tol, atol=tol)
    ptol = 1e-4 if dtype == torch.float32 else 3e-2
    np.testing.assert_allclose(wg.grad.cpu().double().numpy(), wr.grad.numpy(), rtol=ptol, atol=ptol * float(wr.grad.abs().max()))
    np.testing.assert_allclose(bg.grad.cpu().double().numpy(), br.grad.numpy(), rtol=ptol, atol=ptol * float(br.grad.abs().max()))


def _ref_core_fp64(x, xdbl, table, dt_w, dt_bias, a_neg, ds, r):
    """forward_corev2's scan + merge restated in fp64 torch ops (sequential in l), differentiable."""
    b, l, d = x.shape
    k = table.shape[0]
    rg = xdbl.shape[-1] // k
    r8 = rg - 4
    xd = xdbl.view(b, l, k, rg)
    ym = torch.zeros(b, l, d, dtype=torch.float64)
    for i in range(k):
        idx = table[i].long()
        u = x[:, idx, :]
        rows = xd[:, idx, i, :]
        raw = rows[..., :r] @ dt_w[i].t()
        bv, cv = rows[..., r8], rows[..., r8 + 1]
        dt = F.softplus(raw + dt_bias[i])
        a = torch.exp(dt * a_neg[i])
        h = torch.zeros(b, d, dtype=torch.float64)
        ys = []
        for t in range(l):
            h = a[:, t] * h + dt[:, t] * bv[:, t, None] * u[:, t]
            ys.append(cv[:, t, None] * h + ds[i] * u[:, t])
        ym = ym.index_add(1, idx, torch.stack(ys, 1))
    return ym


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("fam,h,d,r,b", [("raster", 24, 64, 32, 2), ("helix", 96, 256, 8, 1), ("raster", 96, 256, 8, 2),
                                         ("helix", 48, 512, 16, 1), ("window", 24, 40, 3, 1), ("raster", 12, 2048, 64, 1),
                                         ("dilation", 13, 96, 5, 1)])
def test_ss2d_saved_states_equal_the_recomputed_ones(dtype, fam, h, d, r, b):
    """Training: the forward launch saves the state entering every tile (ss2d_scan_cl(states=...), register-ring and LDS-DMA
    forms, ragged sequence ends) and the backward launch skips the sweep that recomputes them: same ys as the plain forward,
    and every output of the backward bit-identical to the run that recomputes."""
    H = hip()
    dev = torch.device(DEV)
    if dtype == torch.float32 and h >= 48:
        pytest.skip("fp32 at the large maps: covered at the small ones")
    order = H.scan_order(fam, h, h, dev)
    k, l = order.k, h * h
    rg = H.ss2d_group_stride(r)
    g = torch.Generator().manual_seed(h + d + r)
    x = torch.randn(b, l, d, generator=g).to(dtype).to(dev)
    xdbl = torch.zeros(b, l, k, rg)
    xdbl[..., :r] = 0.5 * torch.randn(b, l, k, r, generator=g)
    xdbl[..., rg - 4:rg - 2] = torch.randn(b, l, k, 2, generator=g)
    xdbl = xdbl.view(b, l, k * rg).to(dev)
    dt_w = (torch.randn(k, d, r, generator=g) * r ** -0.5).to(dev)
    dt_b = (torch.randn(k * d, generator=g) * 0.5 - 1.0).to(dev)
    a_neg = (-(torch.rand(k * d, generator=g) * 0.8 + 0.2)).to(dev)
    ds = (1 + 0.1 * torch.randn(k * d, generator=g)).to(dev)
    gym = torch.randn(b, l, d, generator=g).to(dtype).to(dev)
    states = H.ss2d_scan_states(x, order)
    states.fill_(0xFF)                                           # NaN patterns: an unwritten state would poison the result
    ys_s = H.ss2d_scan_cl(x, xdbl, order, dt_w, dt_b, a_neg, ds, dtype, states=states)
    ys_p = H.ss2d_scan_cl(x, xdbl, order, dt_w, dt_b, a_neg, ds, dtype, segmented=False)
    assert torch.equal(ys_s, ys_p)
    with_states = H.ss2d_scan_bwd_cl(x, xdbl, order, dt_w, dt_b, a_neg, ds, gym, states=states)
    recomputed = H.ss2d_scan_bwd_cl(x, xdbl, order, dt_w, dt_b, a_neg, ds, gym)
    for name, u, v in zip(("gu", "graw", "gB", "gC", "gpar"), with_states, recomputed):
        assert torch.isfinite(u.float()).all(), name
        if name in ("gB", "gC"):                                 # accumulated by float atomics over the channel tiles
            np.testing.assert_allclose(u.cpu().numpy(), v.cpu().numpy(), rtol=1e-4, atol=1e-4 * float(v.abs().max()), err_msg=name)
        else:
            assert torch.equal(u, v), name


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("fam,h,d,r,b", [("raster", 12, 64, 4, 2), ("helix", 12, 32, 8, 1), ("window", 24, 40, 3, 1),
                                         ("dilation", 16, 96, 16, 2), ("raster", 24, 64, 40, 1)])
def test_ss2d_core_training_gradients(dtype, fam, h, d, r, b):
    """_SS2DCoreCL (fused forward + tramba_ss2d_scan_bwd_cl + projection GEMMs) against fp64 autograd of the
    restated graph: merged output and the gradients of x, the x_proj rows, dt_w, dt_bias, A and D."""
    from tramba_amd import modules as M
    H = hip()
    dev = torch.device(DEV)
    order = H.scan_order(fam, h, h, dev)
    k, l = order.k, h * h
    rg = H.ss2d_group_stride(r)
    g = torch.Generator().manual_seed(h * d + r)
    x = torch.randn(b, l, d, generator=g).to(dtype)
    xdbl = torch.zeros(b, l, k, rg)
    xdbl[..., :r] = 0.5 * torch.randn(b, l, k, r, generator=g)
    xdbl[..., rg - 4:rg - 2] = torch.randn(b, l, k, 2, generator=g)
    xdbl = xdbl.view(b, l, k * rg)
    dt_w = torch.randn(k, d, r, generator=g) * r ** -0.5
    dt_b = torch.randn(k, d, generator=g) * 0.5 - 1.0
    a_neg = -(torch.rand(k, d, generator=g) * 0.8 + 0.2)
    ds = 1 + 0.1 * torch.randn(k, d, generator=g)
    gym = torch.randn(b, l, d, generator=g)
    # reference
    leaves = [t.double().requires_grad_(True) for t in (x.float(), xdbl, dt_w, dt_b, a_neg, ds)]
    ymr = _ref_core_fp64(leaves[0], leaves[1], order.table.cpu(), leaves[2], leaves[3], leaves[4], leaves[5], r)
    ymr.backward(gym.double())
    # device
    dl = [x.to(dev).requires_grad_(True), xdbl.to(dev).requires_grad_(True), dt_w.to(dev).requires_grad_(True),
          dt_b.reshape(-1).to(dev).requires_grad_(True), a_neg.reshape(-1).to(dev).requires_grad_(True),
          ds.reshape(-1).to(dev).requires_grad_(True)]
    ym = M._SS2DCoreCL.apply(*dl, order)
    ym.backward(gym.to(dev))
    f32 = dtype == torch.float32

    def close(got, want, name, rel):
        got, want = got.detach().double().cpu().reshape(want.shape), want.detach()
        scale = float(want.abs().max()) + 1e-12
        err = float((got - want).abs().max()) / scale
        assert err < rel, (name, err)

    close(ym, ymr, "ym", 2e-4 if f32 else 2e-2)
    close(dl[0].grad, leaves[0].grad, "gx", 3e-4 if f32 else 3e-2)
    close(dl[1].grad, leaves[1].grad, "gxdbl", 5e-4 if f32 else 5e-2)
    close(dl[2].grad, leaves[2].grad, "gdt_w", 5e-4 if f32 else 5e-2)
    close(dl[3].grad, leaves[3].grad, "gbias", 5e-4 if f32 else 5e-2)
    close(dl[4].grad, leaves[4].grad, "gA", 5e-4 if f32 else 5e-2)
    close(dl[5].grad, leaves[5].grad, "gD", 5e-4 if f32 else 5e-2)


@pytest.mark.parametrize("fam,b,h,d,r", [("helix", 8, 96, 256, 8), ("raster", 8, 96, 256, 8), ("helix", 8, 48, 512, 16),
                                         ("window", 8, 48, 512, 16)])
def test_ss2d_scan_backward_at_the_benchmarked_launches(fam, b, h, d, r):
    """VERDICT r3 'missing' #3: the backward launches of a BASELINE configs[2] training step -- ss2d_scan_bwd_cl_kernel at B = 8,
    bf16, as _SS2DInnerCL issues it (states saved by the forward launch, A_logs in, dB / dC as per-channel-tile partials):
    96x96 K = 8 D = 256 (bench.py's `roofline_scan_bwd.helix_top_stage`), its K = 4 sibling and the 48x48 rank-16 launches --
    every output against the fp64 oracle of the op it replaces DIRECTLY (csms6s.py:914-923: selective_scan_cuda_oflex.bwd on the
    gathered operands, oracle/selective_scan_ref.c), image by image so that the oracle's fp64 tensors stay at 150 MB each."""
    from oracle import selective_scan as oss
    H = hip()
    dev = torch.device(DEV)
    dtype = torch.bfloat16
    order = H.scan_order(fam, h, h, dev)
    k, l = order.k, h * h
    rg = H.ss2d_group_stride(r)
    g = torch.Generator().manual_seed(h * d + r + k)
    x = torch.randn(b, l, d, generator=g).to(dtype)
    xdbl = torch.zeros(b, l, k, rg)
    xdbl[..., :r] = 0.5 * torch.randn(b, l, k, r, generator=g)
    xdbl[..., rg - 4:rg - 2] = torch.randn(b, l, k, 2, generator=g)
    xdbl = xdbl.view(b, l, k * rg)
    dt_w = torch.randn(k, d, r, generator=g) * r ** -0.5
    dt_b = torch.randn(k, d, generator=g) * 0.5 - 1.0
    a_logs = torch.log(torch.rand(k, d, generator=g) * 0.8 + 0.2)
    ds = 1 + 0.1 * torch.randn(k, d, generator=g)
    gym = torch.randn(b, l, d, generator=g).to(dtype)          # the merged map's gradient arrives in the activation dtype
    xd, xdd, gd = x.to(dev), xdbl.to(dev), gym.to(dev)
    par = (dt_w.to(dev), dt_b.reshape(-1).to(dev), a_logs.reshape(-1).to(dev), ds.reshape(-1).to(dev))
    states = H.ss2d_scan_states(xd, order)
    H.ss2d_scan_cl(xd, xdd, order, *par, dtype, states=states, a_log=True)
    gu, graw, gB, gC, gpar = H.ss2d_scan_bwd_cl(xd, xdd, order, *par, gd, states=states, a_log=True, bc_partials=True)
    torch.cuda.synchronize()
    H.device_error()
    gu, graw = gu.cpu(), graw.cpu()
    gB, gC, gpar = gB.sum(dim=2).cpu().double(), gC.sum(dim=2).cpu().double(), gpar.cpu().double()
    tbl = order.table.cpu().long()                                # (K, L)
    a_neg = -torch.exp(a_logs.double()).reshape(k * d, 1)
    worst = {}

    def close(got, want, name, rel, rms):
        got, want = got.double().reshape(want.shape), want.double()
        e_max = float((got - want).abs().max()) / (float(want.abs().max()) + 1e-30)
        e_rms = float((got - want).pow(2).mean().sqrt()) / (float(want.pow(2).mean().sqrt()) + 1e-30)
        worst[name] = max(worst.get(name, (0, 0)), (e_max, e_rms))
        assert e_max <= rel and e_rms <= rms, (fam, name, e_max, e_rms)

    for i in range(b):
        xi, ri = x[i].double(), xdbl[i].double().view(l, k, rg)
        u = torch.stack([xi[tbl[j]].t() for j in range(k)]).reshape(1, k * d, l)                # (1, KD, L)
        rows = torch.stack([ri[tbl[j], j] for j in range(k)])                                    # (K, L, RG)
        delta = torch.einsum("klr,kdr->kdl", rows[..., :r], dt_w.double()).reshape(1, k * d, l)
        Bm, Cm = rows[..., rg - 4].reshape(1, k, 1, l), rows[..., rg - 3].reshape(1, k, 1, l)
        dout = torch.stack([gym[i].double()[tbl[j]].t() for j in range(k)]).reshape(1, k * d, l)
        du, dd, dA, dB, dC, dD, dbias = oss.selective_scan_bwd(u, delta.contiguous(), a_neg, Bm.contiguous(), Cm.contiguous(),
                                                               ds.double().reshape(-1), dt_b.double().reshape(-1), dout)
        # 16-bit outputs: 2^-9 relative rounding on top of the kernel's fp32 arithmetic on bf16-rounded rank rows.  Measured on
        # MI355X (r04): max 0.003-0.006 of the largest element, RMS 0.0018 (gu, graw) / 0.001 (the sums); bounds = ~2.5x that
        close(gu[i].permute(0, 2, 1), du.reshape(k, d, l), "gu", 1.5e-2, 4.5e-3)
        close(graw[i].permute(0, 2, 1), dd.reshape(k, d, l), "graw", 1.2e-2, 4.5e-3)
        close(gB[i], dB.reshape(k, l), "gB", 8e-3, 2.5e-3)
        close(gC[i], dC.reshape(k, l), "gC", 5e-3, 2.5e-3)
        close(gpar[i, 0], (dA.reshape(k, d) * a_neg.reshape(k, d)), "gA_log", 5e-3, 2.5e-3)
        close(gpar[i, 1], dD.reshape(k, d), "gD", 5e-3, 2.5e-3)
        close(gpar[i, 2], dbias.reshape(k, d), "gbias", 5e-3, 2.5e-3)
    print("scan backward vs fp64 oracle (max / rms relative):", fam, h, {n: (round(a, 5), round(c, 5)) for n, (a, c) in worst.items()})


@pytest.mark.parametrize("partner", ["linear_pc", "linear_ws", "scan_dma"])
def test_wgrad_is_bitwise_stable_beside_another_stream(partner):
    """r04: tramba_wgrad_cl (wgrad_dma_kernel: LDS-DMA staged token tiles read back TRANSPOSED by ds_read_b64_tr_b16) while a
    second stream keeps the CUs busy with an LDS-heavy kernel of another kind -- the producer / consumer and weight-stationary
    projections, the LDS-DMA scan.  With r03's counted lgkmcnt waits on the transposed reads 16-28 of 180 such launches
    differed from the launch running alone (a few elements, NaN now and then: the data-parallel step overlaps RCCL kernels
    with backward the same way); the kernel now waits for all reads of a step before its first MFMA."""
    H = hip()
    dev = torch.device(DEV)
    g = torch.Generator().manual_seed(3)
    shapes = [(4608, 512, 2048), (4608, 1024, 512), (18432, 256, 1024), (73728, 128, 512), (1152, 1024, 4096)]
    ops = [(torch.randn(m, n, generator=g).bfloat16().to(dev), torch.randn(m, k, generator=g).bfloat16().to(dev)) for m, n, k in shapes]
    if partner == "scan_dma":
        b, h, d, r, kk = 8, 96, 256, 8, 8
        order = H.scan_order("helix", h, h, dev)
        xc = torch.randn(b, h * h, d, generator=g).bfloat16().to(dev)
        wx = (torch.randn(kk, r + 2, d, generator=g) * d ** -0.5).bfloat16().to(dev)
        xdbl = H.linear_cl(xc, H.pad_x_proj_weight(wx), out_dtype=torch.float32)
        sargs = (xc, xdbl, order, (torch.randn(kk, d, r, generator=g) * r ** -0.5).to(dev),
                 (torch.randn(kk * d, generator=g) * 0.5 - 2.0).to(dev), (-0.5 - torch.rand(kk * d, generator=g)).to(dev),
                 torch.ones(kk * d).to(dev), torch.bfloat16)
        other = lambda: H.ss2d_scan_cl(*sargs)
    elif partner == "linear_ws":
        a = torch.randn(73728, 128, generator=g).bfloat16().to(dev)
        w = (torch.randn(512, 128, generator=g) * 0.1).bfloat16().to(dev)
        other = lambda: H.linear_cl(a, w, None, None, 2)
    else:
        a = torch.randn(4608, 2048, generator=g).bfloat16().to(dev)
        w = (torch.randn(512, 2048, generator=g) * 0.02).bfloat16().to(dev)
        other = lambda: H.linear_cl(a, w, None, None, 0)
    ref = [H.wgrad_cl(gy, x, True) for gy, x in ops]
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    bad = 0
    for _ in range(20):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(6):
                other()
        outs = [H.wgrad_cl(gy, x, True) for gy, x in ops]
        torch.cuda.synchronize()
        bad += sum(int(not (torch.equal(o[0], r_[0]) and torch.equal(o[1], r_[1]))) for o, r_ in zip(outs, ref))
    assert bad == 0, f"{bad} of {20 * len(ops)} weight-gradient launches differ beside {partner}"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cfg", [(2304, 512, 512, 512), (300, 128, 64, 192), (9216, 256, 256, 256), (1000, 72, 128, 64)])
def test_linear2_cl_matches_concatenated_gemm(dtype, cfg):
    """tramba_linear2_cl(x1, x2) == tramba_linear_cl(cat(x1, x2)) bit for bit (same K order, same kernel), with the
    bias / GELU / residual epilogue and with the sigmoid gate (freq_mamba.py:52-56)."""
    m, n, k1, k2 = cfg
    H = hip()
    g = torch.Generator().manual_seed(m + n + k1)
    x1 = torch.randn(m, k1, generator=g).to(dtype).to(DEV)
    x2 = torch.randn(m, k2, generator=g).to(dtype).to(DEV)
    w = (torch.randn(n, k1 + k2, generator=g) * (k1 + k2) ** -0.5).to(dtype).to(DEV)
    bias = torch.randn(n, generator=g).to(DEV)
    res = torch.randn(m, n, generator=g).to(dtype).to(DEV)
    xc = torch.cat((x1, x2), dim=-1)
    assert torch.equal(H.linear2_cl(x1, x2, w, bias, res, 2), H.linear_cl(xc, w, bias, res, 2))
    assert torch.equal(H.linear2_cl(x1, x2, w, None, None, 0, torch.float32), H.linear_cl(xc, w, None, None, 0, torch.float32))
    gate = H.linear2_cl(x1, x2, w, None, res, H.ACT_SIGMOID_GATE)
    want = torch.sigmoid(xc.double() @ w.double().T) * res.double()
    np.testing.assert_allclose(gate.cpu().double().numpy(), want.cpu().numpy(), rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("mnk", [(2304, 512, 2048), (576, 1024, 512), (333, 72, 136), (100, 30, 50)])
def test_linear_gelu_grad_epilogue(dtype, mnk):
    """TRAMBA_ACT_GELU_GRAD_MUL: y = (x @ w^T + b) * gelu'(h) -- the input gradient of Linear(GELU(h)) in one launch --
    against fp64 autograd of the exact (erf) GELU; tiled and fallback kernels (ragged / unaligned shapes), and the gate /
    gradient modes refuse to run without their second operand."""
    m, n, k = mnk
    H = hip()
    g = torch.Generator().manual_seed(m + n)
    x = torch.randn(m, k, generator=g).to(dtype).to(DEV)
    w = (torch.randn(n, k, generator=g) * k ** -0.5).to(dtype).to(DEV)
    h = (torch.randn(m, n, generator=g) * 1.5).to(dtype).to(DEV)
    hd = h.double().requires_grad_(True)
    (dg,) = torch.autograd.grad(F.gelu(hd).sum(), hd)
    want = (x.double() @ w.double().T) * dg
    got = H.linear_cl(x, w, None, h, H.ACT_GELU_GRAD_MUL)
    assert got.dtype == dtype
    tol = 1e-4 if dtype == torch.float32 else 2e-2
    np.testing.assert_allclose(got.cpu().double().numpy(), want.cpu().numpy(), rtol=tol, atol=tol * max(1.0, float(want.abs().max())))
    for act in (H.ACT_GELU_GRAD_MUL, H.ACT_SIGMOID_GATE):
        with pytest.raises(H.TrambaHipError, match="residual"):
            H.linear_cl(x, w, None, None, act)


@pytest.mark.parametrize("shape", [(576, 2, 512), (1024, 2, 128), (192, 10, 1024), (8, 4, 3, 1024), (7, 12), (300, 36), (5, 3),
                                   (1, 64)])
def test_slab_sum_fixed_order(shape):
    """tramba_slab_sum: the partial-sum tables of the backward kernels summed over their first axis -- equal to an fp64 sum
    to fp32 rounding, bitwise reproducible, any slab count (the row lanes of a block see ragged tails)."""
    H = hip()
    part = torch.randn(shape, generator=torch.Generator().manual_seed(sum(shape))).to(DEV)
    got = H.slab_sum(part)
    assert got.shape == part.shape[1:] and got.dtype == torch.float32
    want = part.double().sum(0)
    np.testing.assert_allclose(got.cpu().double().numpy(), want.cpu().numpy(), rtol=1e-5, atol=1e-5 * shape[0] ** 0.5)
    assert torch.equal(got, H.slab_sum(part))


@pytest.mark.parametrize("cfg", [(8, 48, 384), (8, 96, 384), (2, 192, 384), (3, 13, 37), (1, 7, 7), (2, 24, 25)])
def test_upsample_bilinear_backward_is_the_adjoint(cfg):
    """tramba_upsample_bilinear_bwd (the resize of the deep-supervision outputs in the loss, train.py:76-85) == autograd of
    F.interpolate(mode="bilinear"): every input pixel gathers its weights (borders, non-integer scales included)."""
    from tramba_amd import train
    b, n, m = cfg
    g = torch.Generator().manual_seed(n * m)
    x = torch.randn(b, 1, n, n, generator=g).to(DEV).requires_grad_(True)
    gy = torch.randn(b, 1, m, m, generator=g).to(DEV)
    F.interpolate(x, (m, m), mode="bilinear").backward(gy)
    want = x.grad.clone()
    x.grad = None
    y = train._resize_bilinear(x, (m, m))
    assert torch.equal(y, F.interpolate(x.detach(), (m, m), mode="bilinear"))
    y.backward(gy)
    np.testing.assert_allclose(x.grad.cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=1e-5 * float(want.abs().max()))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("mnk", [(4608, 2048, 512), (576, 4096, 1024), (1000, 72, 128), (73, 8, 64)])
def test_linear_dual_output(dtype, mnk):
    """tramba_linear_dual_cl: the pre-activation and its GELU from one launch == the two single-output launches bit for bit
    (same kernel, same accumulation order), ragged M / N included."""
    m, n, k = mnk
    H = hip()
    g = torch.Generator().manual_seed(m + n)
    x = torch.randn(m, k, generator=g).to(dtype).to(DEV)
    w = (torch.randn(n, k, generator=g) * k ** -0.5).to(dtype).to(DEV)
    bias = torch.randn(n, generator=g).to(DEV)
    pre, act = H.linear_dual_cl(x, w, bias, H.ACT_GELU)
    assert torch.equal(pre, H.linear_cl(x, w, bias)) and torch.equal(act, H.linear_cl(x, w, bias, None, H.ACT_GELU))
    pre, act = H.linear_dual_cl(x, w, None, H.ACT_SILU)
    assert torch.equal(pre, H.linear_cl(x, w)) and torch.equal(act, H.linear_cl(x, w, None, None, H.ACT_SILU))


def test_shadow_cast_multi_matches_cast_and_transpose():
    """tramba_shadow_cast_multi: one launch writes the 16-bit copy and the 16-bit transpose of every matrix in a device
    table (ragged shapes, shapes below one tile, a skipped destination) == .to(dtype) / .t() bit for bit."""
    H = hip()
    g = torch.Generator().manual_seed(5)
    shapes = [(512, 2048), (2048, 512), (136, 1024), (1, 128), (40, 8), (65, 67), (3, 5), (1024, 1024)]
    for dtype in (torch.bfloat16, torch.float16):
        src = [torch.randn(s, generator=g).to(DEV) for s in shapes]
        dst = [torch.full(s, 7.0, dtype=dtype, device=DEV) for s in shapes]
        dst_t = [torch.full((s[1], s[0]), 7.0, dtype=dtype, device=DEV) for s in shapes]
        rows, first = [], 0
        for i, (a, b, c) in enumerate(zip(src, dst, dst_t)):
            r, cc = a.shape
            rows.append([a.data_ptr(), 0 if i == 3 else b.data_ptr(), 0 if i == 4 else c.data_ptr(), r, cc, first, cc, r])
            first += ((r + 63) // 64) * ((cc + 63) // 64)
        table = torch.tensor(rows, dtype=torch.int64).to(DEV)
        H.shadow_cast_multi(table, len(rows), first, dtype)
        for i, (a, b, c) in enumerate(zip(src, dst, dst_t)):
            assert torch.equal(b, torch.full_like(b, 7.0) if i == 3 else a.to(dtype)), (i, dtype)
            assert torch.equal(c, torch.full_like(c, 7.0) if i == 4 else a.to(dtype).t()), (i, dtype)
        # blocks of a padded layout: two (R, D) matrices into rows 0.. and 12.. of a (24, D) buffer, transposes into columns
        # 0.. and 12.. of a (D, 24) buffer (the x_proj weight of the fused scan)
        for r_, d_ in ((8, 256), (3, 40)):
            blocks = [torch.randn(r_, d_, generator=g).to(DEV) for _ in range(2)]
            pad = torch.zeros(24, d_, dtype=dtype, device=DEV)
            pad_t = torch.zeros(d_, 24, dtype=dtype, device=DEV)
            es = pad.element_size()
            rows2 = [[blk.data_ptr(), pad.data_ptr() + 12 * j * d_ * es, pad_t.data_ptr() + 12 * j * es, r_, d_, j * ((d_ + 63) // 64),
                      d_, 24] for j, blk in enumerate(blocks)]
            H.shadow_cast_multi(torch.tensor(rows2, dtype=torch.int64).to(DEV), 2, 2 * ((d_ + 63) // 64), dtype)
            want = torch.zeros(24, d_, dtype=dtype, device=DEV)
            for j, blk in enumerate(blocks):
                want[12 * j:12 * j + r_] = blk.to(dtype)
            assert torch.equal(pad, want) and torch.equal(pad_t, want.t()), (r_, d_, dtype)


# ----------------------------------------------------------------------------- training-path GEMMs (train_gemm.hip)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("m,n,k", [(576, 128, 512), (2304, 256, 128), (1000, 8, 136), (33, 264, 72), (9216, 48, 256),
                                   (4608, 1024, 256)])
def test_wgrad_tn_against_fp64(m, n, k, dtype):
    """gw = gy^T x and gb = column sums of gy (the weight / bias gradient of Linear2d, modules.py:10-19 under autograd)
    against an fp64 contraction of the same rounded operands: ragged token counts (tail steps, several splits), channel
    counts that are not multiples of the 128 tile, strided rows."""
    H = hip()
    g = torch.Generator().manual_seed(m + n + k)
    wide = torch.randn(m, n + 16, generator=g).to(DEV, dtype)
    gy = wide[:, 8:8 + n]                                         # strided rows, 16-byte aligned start
    x = torch.randn(m, k, generator=g).to(DEV, dtype)
    gw, gb = H.wgrad_cl(gy, x, want_bias=True)
    want = gy.double().t() @ x.double()
    scale = float((gy.double().abs().t() @ x.double().abs()).max())
    assert gw.shape == (n, k) and gw.dtype == torch.float32
    assert float((gw.double() - want).abs().max()) <= 2e-6 * scale + 1e-6
    wb = gy.double().sum(0)
    assert float((gb.double() - wb).abs().max()) <= 2e-6 * float(gy.double().abs().sum(0).max()) + 1e-6
    gw2, none = H.wgrad_cl(gy.contiguous(), x)
    assert none is None and torch.equal(gw2, gw)                  # strides do not change the summation order


@pytest.mark.parametrize("b,kk,l,d,r", [(2, 4, 144, 64, 8), (1, 8, 576, 128, 16), (2, 8, 2304, 256, 8), (1, 4, 100, 2048, 64),
                                        (1, 4, 576, 1024, 32), (2, 4, 150, 512, 16),
                                        (1, 4, 576, 40, 8)])
def test_ss2d_backward_small_contractions(b, kk, l, d, r):
    """the per-direction projections of the SS2D backward (vmamba.py:233-236 under autograd): d(dt_projs_weight)[k] =
    sum_{b,l} graw^T ranks (grouped TN GEMM) and d(ranks) = graw @ dt_projs_weight[k] (rows_gemm into strided fp32 rows)"""
    H = hip()
    g = torch.Generator().manual_seed(b * l + d)
    graw = torch.randn(b, kk, l, d, generator=g).to(DEV, torch.bfloat16)
    ranks = torch.randn(b, kk, l, r, generator=g).to(DEV, torch.bfloat16)
    dt_w = (torch.randn(kk, d, r, generator=g) * d ** -0.5).to(DEV)
    got = H.wgrad_grouped_cl(graw, ranks)
    want = torch.einsum("bkld,bklr->kdr", graw.double(), ranks.double())
    scale = float(torch.einsum("bkld,bklr->kdr", graw.double().abs(), ranks.double().abs()).max())
    assert got.shape == (kk, d, r) and float((got.double() - want).abs().max()) <= 2e-6 * scale + 1e-6
    rg = r + 4
    y = torch.full((b * kk, l, rg), 7.0, device=DEV)
    wt = dt_w.transpose(1, 2).contiguous().to(torch.bfloat16)     # (K, R, D)
    H.rows_gemm_cl(graw.view(b * kk, l, d), wt, y, r)
    want2 = torch.einsum("bkld,krd->bklr", graw.double(), wt.double()).reshape(b * kk, l, r)
    assert float((y[..., :r].double() - want2).abs().max()) <= 1e-5 * float(want2.abs().max()) + 1e-5
    assert torch.all(y[..., r:] == 7.0)                           # columns past N untouched


@pytest.mark.parametrize("fam,h,d,r,b", [("helix", 32, 64, 8, 2), ("helix", 40, 96, 8, 1), ("raster", 48, 32, 4, 3),
                                         ("window", 96, 256, 8, 1), ("raster", 48, 64, 16, 2), ("dilation", 24, 1024, 32, 4),
                                         ("helix", 48, 128, 32, 1), ("window", 48, 512, 12, 4)])
@pytest.mark.parametrize("ys_dtype", [torch.float32, torch.bfloat16])
def test_ss2d_scan_forms_agree(fam, h, d, r, b, ys_dtype):
    """The three schedules of the fused scan -- chained on a register ring (<= 8 waves per sequence), wave-segment (two
    passes), chained on LDS-DMA staged operands (16 or 8 waves per sequence, padded dt_rank 8 / 16 / 32) -- compute the same
    recurrence; only the order in which tile aggregates are folded differs.  Ragged last super-chunks included (40x40 =
    1600 positions = 3.125 super-chunks of 512).  The chained form itself is checked against the fp64 oracle above."""
    H = hip()
    g = torch.Generator().manual_seed(h * d + r)
    order = H.scan_order(fam, h, h, torch.device(DEV))
    k, l = order.k, h * h
    x = torch.randn(b, l, d, generator=g).to(DEV, torch.bfloat16)
    wx = (torch.randn(k, r + 2, d, generator=g) * d ** -0.5).to(DEV, torch.bfloat16)
    xdbl = H.linear_cl(x, H.pad_x_proj_weight(wx), out_dtype=torch.float32)
    dt_w = (torch.randn(k, d, r, generator=g) * r ** -0.5).to(DEV)
    dt_b = (torch.randn(k * d, generator=g) * 0.5 - 3).to(DEV)
    a = -(0.5 + torch.rand(k * d, generator=g)).to(DEV)
    ds = (1 + 0.1 * torch.randn(k * d, generator=g)).to(DEV)
    out = {}
    try:
        for form in (1, 2, 3):
            H.tune_set(H.TUNE_SCAN_FORM, form)
            out[form] = H.ss2d_scan_cl(x, xdbl, order, dt_w, dt_b, a, ds, ys_dtype).float()
    finally:
        H.tune_set(H.TUNE_SCAN_FORM, 0)
    scale = float(out[1].abs().max())
    tol = 2e-5 * scale if ys_dtype == torch.float32 else 1e-2 * scale
    for form in (2, 3):
        assert float((out[form] - out[1]).abs().max()) <= tol, (form, float((out[form] - out[1]).abs().max()), scale)
    assert torch.isfinite(out[3]).all()


def test_linear_epilogue_operands_are_validated():
    """a residual in another dtype / shape, or a non-fp32 bias, would be misread by the kernel: host error."""
    H = hip()
    x = torch.randn(64, 128, device=DEV).to(torch.bfloat16)
    w = torch.randn(64, 128, device=DEV).to(torch.bfloat16)
    ok = H.linear_cl(x, w, torch.zeros(64, device=DEV), torch.zeros(64, 64, device=DEV, dtype=torch.bfloat16))
    assert ok.shape == (64, 64)
    with pytest.raises(H.TrambaHipError, match="residual"):
        H.linear_cl(x, w, None, torch.zeros(64, 64, device=DEV), out_dtype=torch.float32)     # fp32 residual
    with pytest.raises(H.TrambaHipError, match="residual"):
        H.linear_cl(x, w, None, torch.zeros(64, 32, device=DEV, dtype=torch.bfloat16))        # wrong shape
    with pytest.raises(H.TrambaHipError, match="bias"):
        H.linear_cl(x, w, torch.zeros(64, device=DEV, dtype=torch.bfloat16))                  # 16-bit bias


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("m,n,k,act,res", [(2304, 1024, 512, 0, False), (2304, 2048, 512, 2, False), (576, 4096, 1024, 2, False),
                                           (36864, 256, 128, 0, False), (1000, 72, 192, 0, True), (9216, 512, 256, 2, True)])
def test_layernorm_folded_into_the_gemm(m, n, k, act, res, dtype):
    """tramba_linear_ln_cl = LayerNorm2d then Linear2d (VSSBlock norm -> in_proj, norm2 -> fc1; vmamba.py:384-396) in one
    launch, against an fp64 evaluation of the same rounded inputs, and against the two-launch path it replaces.  Rows with a
    large mean (the cancellation case of the folded form) included."""
    H = hip()
    from tramba_amd.modules import _fold_layernorm
    import tramba_amd as ta
    g = torch.Generator().manual_seed(m + n + k)
    x = torch.randn(m, k, generator=g) * 1.3 + 0.2
    x[::7] += 6.0                                                        # rows whose mean is 5x their spread
    x = x.to(DEV, dtype)
    lin = ta.Linear2d(k, n, bias=True).to(DEV)
    norm = ta.LayerNorm2d(k).to(DEV)
    with torch.no_grad():
        lin.weight.copy_(torch.randn(n, k, generator=g) * k ** -0.5)
        lin.bias.copy_(0.1 * torch.randn(n, generator=g))
        norm.weight.copy_(1 + 0.2 * torch.randn(k, generator=g))
        norm.bias.copy_(0.1 * torch.randn(k, generator=g))
    r = torch.randn(m, n, generator=g).to(DEV, dtype) if res else None
    wf, cs, tb = _fold_layernorm(lin, norm, dtype)
    got = H.linear_ln_cl(x, wf, cs, tb, norm.eps, r, act).double().cpu()
    xd = x.double().cpu()
    ln = torch.nn.functional.layer_norm(xd, (k,), norm.weight.double().cpu(), norm.bias.double().cpu(), norm.eps)
    want = ln @ lin.weight.double().cpu().t() + lin.bias.double().cpu()
    if act == 2:
        want = torch.nn.functional.gelu(want)
    if res:
        want = want + r.double().cpu()
    two = H.linear_cl(H.layernorm_cl(x, norm.weight.detach(), norm.bias.detach(), norm.eps), lin.weight.detach().to(dtype),
                      lin.bias.detach(), r, act).double().cpu()
    scale = float(want.abs().max())
    tol = (2e-2 if dtype == torch.bfloat16 else 3e-3) * scale
    e_fused, e_two = float((got - want).abs().max()), float((two - want).abs().max())
    assert e_fused <= tol, (e_fused, e_two, scale)
    assert e_fused <= 2.0 * e_two + 1e-3 * scale, (e_fused, e_two)          # no worse than the path it replaces


def test_refreshing_the_depthwise_packs_invalidates_a_forward_that_saved_them():
    """refresh_dw_packs rewrites the standing packs through raw pointers after an optimizer step; a backward whose forward saved
    a pack BEFORE the refresh must raise autograd's in-place error instead of silently using the new weights' stencil
    (forward - forward - step - backward, a second loss on an earlier graph)"""
    import tramba_amd as ta
    from tramba_amd import modules as M, train
    torch.manual_seed(7)
    m = ta.bulid_model(use_pretrain=False, img_size=384).to(DEV).train()
    m.compute_dtype = torch.bfloat16
    opt = train.get_opt(1e-4, m)
    x = torch.randn(1, 3, 384, 384, device=DEV)
    y = (torch.rand(1, 1, 384, 384, device=DEV) > 0.7).float()
    train.train_step(m, opt, x, y)             # (the standing packs exist from the first step's refresh on)
    outs = m(x)
    assert M.refresh_dw_packs(m) > 0           # as after an optimizer step
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        train.tramba_loss(outs, y).backward()
    opt.zero_grad(set_to_none=True)
    loss = train.train_step(m, opt, x, y)      # the normal order is untouched
    assert bool(torch.isfinite(loss))
