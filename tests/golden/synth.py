"""Closed-form synthetic weights and inputs shared by the golden generator and the tests.

No weight blob is shipped: every parameter is a deterministic function of its NAME and
SHAPE (numpy RandomState = frozen legacy stream, bit-stable across numpy versions and
hosts), scaled so activations stay O(1) through a 100-layer network.
"""
import zlib

import numpy as np
import torch


def _rs(name: str, salt: int = 0) -> np.random.RandomState:
    return np.random.RandomState((zlib.crc32(name.encode()) + salt) % (2 ** 31))


def synth_tensor(name: str, shape, dtype=torch.float32) -> torch.Tensor:
    shape = tuple(int(s) for s in shape)
    rs = _rs(name)
    leaf = name.split(".")[-1]
    n = int(np.prod(shape)) if shape else 1

    def randn():
        return rs.standard_normal(n).reshape(shape)

    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.long)
    if leaf == "running_mean":
        v = 0.1 * randn()
    elif leaf == "running_var":
        v = 1.0 + 0.2 * np.abs(randn())
    elif leaf == "A_logs":
        v = np.log(0.5 + 1.5 * rs.random_sample(n).reshape(shape))  # A in [-2, -0.5]
    elif leaf == "Ds":
        v = 1.0 + 0.1 * randn()
    elif leaf == "dt_projs_bias":
        dt = np.exp(rs.random_sample(n).reshape(shape) * (np.log(0.1) - np.log(1e-3)) + np.log(1e-3))
        v = dt + np.log(-np.expm1(-dt))  # inverse softplus, like mamba_init.py:20-24
    elif leaf == "dt_projs_weight":
        v = randn() * (shape[-1] ** -0.5)
    elif leaf == "x_proj_weight":
        v = randn() * (shape[-1] ** -0.5)
    elif leaf == "bias":
        v = 0.02 * randn()
    elif leaf == "weight" and len(shape) == 1:  # LayerNorm / BatchNorm scale
        v = 1.0 + 0.1 * randn()
    elif leaf == "weight":
        fan_in = int(np.prod(shape[1:]))
        v = randn() * (fan_in ** -0.5)
    else:
        v = 0.1 * randn()
    return torch.from_numpy(np.asarray(v)).to(dtype)


def synth_state_dict(manifest, keep=(), dtype=torch.float32):
    """manifest: iterable of (name, shape).  Names containing any of ``keep`` are skipped
    (e.g. the DCT buffers, which are constants of the architecture)."""
    out = {}
    for name, shape in manifest:
        if any(k in name for k in keep):
            continue
        out[name] = synth_tensor(name, shape, dtype)
    return out


def synth_input(tag: str, shape, scale=1.0) -> torch.Tensor:
    rs = _rs("input:" + tag)
    return torch.from_numpy(rs.standard_normal(int(np.prod(shape))).reshape(shape) * scale).float()


DCT_KEYS = ("DCT2D.dct_x.weight", "DCT2D.dct_y.weight")
# constants of the architecture (never synthesised): the DCT bases, Swin's shifted-window masks and index tables
CONST_KEYS = DCT_KEYS + ("attn_mask", "relative_position_index")


# --------------------------------------------------------------------------- saliency-metric cases
def metric_cases():
    """(name, pred float32 (h, w) in [0, 1], gt float32 (h, w) of 0/1) for the evaluation-metric fixtures: smooth
    blobs with noise (the usual case), an empty mask, a full mask, a constant prediction, a non-square map and a
    prediction that is already binary.  Closed-form in the case name, like every other synthetic input here."""
    def blob(name, h, w, n, sharp):
        rs = _rs(name)
        yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
        z = np.zeros((h, w))
        for _ in range(n):
            cy, cx = rs.uniform(0.15, 0.85) * h, rs.uniform(0.15, 0.85) * w
            sy, sx = rs.uniform(0.05, 0.25) * h, rs.uniform(0.05, 0.25) * w
            z += rs.uniform(0.5, 1.5) * np.exp(-((yy - cy) ** 2 / (2 * sy * sy) + (xx - cx) ** 2 / (2 * sx * sx)))
        return z * sharp, rs

    cases = []
    for name, h, w in (("blobs_a", 96, 96), ("blobs_b", 120, 72), ("blobs_c", 384, 384)):
        z, rs = blob("metric_" + name, h, w, 3, 1.0)
        gt = (z > 0.55).astype(np.float32)
        logit = 6.0 * (z - 0.55) + 1.2 * rs.standard_normal((h, w)) + 0.8 * np.sin(np.arange(w) / 7.0)[None, :]
        pred = (1.0 / (1.0 + np.exp(-logit))).astype(np.float32)
        cases.append((name, pred, gt))
    z, rs = blob("metric_empty", 64, 64, 2, 1.0)
    pred = (1.0 / (1.0 + np.exp(-(3.0 * (z - 0.8) + 0.5 * rs.standard_normal((64, 64)))))).astype(np.float32)
    cases.append(("empty_gt", pred, np.zeros((64, 64), np.float32)))
    cases.append(("full_gt", pred, np.ones((64, 64), np.float32)))
    z, rs = blob("metric_const", 48, 80, 2, 1.0)
    cases.append(("const_pred", np.full((48, 80), 0.37, np.float32), (z > 0.5).astype(np.float32)))
    z, rs = blob("metric_binary", 72, 72, 3, 1.0)
    gt = (z > 0.6).astype(np.float32)
    flip = rs.random_sample((72, 72)) < 0.08
    cases.append(("binary_pred", np.where(flip, 1.0 - gt, gt).astype(np.float32), gt))
    return cases


def image_pair(tag: str, w: int, h: int):
    """A seeded RGB image (smooth colour ramps + blobs + noise) and a binary blob mask of the same size, as PIL images:
    the input of the data-loader cases (make_golden_data.py / tests/test_data.py)."""
    from PIL import Image
    rs = _rs("img_" + tag)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    cy, cx, r = rs.uniform(0.3, 0.7) * h, rs.uniform(0.3, 0.7) * w, rs.uniform(0.15, 0.3) * min(h, w)
    blob = ((yy - cy) ** 2 + (xx - cx) ** 2) < r * r
    blob |= (np.abs(yy - 0.2 * h) < 0.06 * h) & (np.abs(xx - 0.75 * w) < 0.1 * w)
    rgb = np.stack([xx / w, yy / h, (xx + yy) / (w + h)], -1) * 160 + blob[..., None] * 70 + rs.uniform(0, 25, (h, w, 3))
    return (Image.fromarray(np.clip(rgb, 0, 255).astype(np.uint8), "RGB"),
            Image.fromarray((blob * 255).astype(np.uint8), "L"))


DATA_SIZE = 32                                              # img_size of the loader cases
DATA_SOURCES = [(70, 60), (41, 57), (32, 32), (96, 50)]     # (W, H) of the synthetic originals, cycled
DATA_TRAIN_SAMPLES = 12                                     # consecutive samples through ONE transform instance
