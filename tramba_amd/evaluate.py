"""Evaluation harness of the reference (train.py:101-152 test_one_epoch, test_TSOD.py:46-68, Evaluation/metrics.py)
with the per-image reductions on the GPU.

The reference pulls every prediction to the host and runs five numpy metric objects on it.  Here one HIP kernel
(`tramba_saliency_stats`, csrc/saliency_eval.hip) reduces a batch of predictions against their masks to ~570 numbers
per image -- min / max, sums, two 256-bin histograms, centroid and quadrant moments -- and MAE, F-measure, E-measure
and S-measure are finished from those on the host in fp64.  The classes keep the reference's names and
`step(pred, gt)` / `get_results()` interface (and result dictionaries), so `test_one_epoch` reads like the
reference's; stepping the five objects with the same arrays launches the kernel once.

WeightedFmeasure needs an exact Euclidean distance transform with nearest-pixel indices (scipy's, in the
reference); it is host-side post-processing here exactly as it is there -- there is no HIP variant of it, and it is
not on the path bench.py measures.
"""
import os
import struct
import zlib

import numpy as np
import torch
import torch.nn.functional as F

from . import hip

_EPS = 1e-16


# ----------------------------------------------------------------------------- statistics -> metrics
class ImageStats:
    """Host view of one row of tramba_saliency_stats (layout: include/tramba_hip.h)."""

    def __init__(self, ints: np.ndarray, dbl: np.ndarray, shape):
        self.i, self.d, self.h, self.w = ints, dbl, int(shape[0]), int(shape[1])
        self.n = int(ints[7])
        self.area = int(ints[0])
        # cumulative counts at thresholds 255 .. 0, the order Evaluation/metrics.py:62-66 keeps
        self.tp = np.cumsum(ints[8:264][::-1])
        self.fp = np.cumsum(ints[264:520][::-1])

    # Evaluation/metrics.py:88-104
    def mae(self):
        return self.d[3] / self.n

    # Evaluation/metrics.py:26-86
    def fmeasure(self, beta=0.3):
        inter, nbin = int(self.i[5]), int(self.i[5] + self.i[6])
        if inter == 0:
            adp = 0.0
        else:
            p, r = inter / nbin, inter / self.area
            adp = (1 + beta) * p * r / (beta * p + r)
        pos = self.tp + self.fp
        pos = np.where(pos == 0, 1, pos)
        prec = self.tp / pos
        rec = self.tp / max(self.area, 1)
        num = (1 + beta) * prec * rec
        den = np.where(num == 0, 1, beta * prec + rec)
        return adp, prec, rec, num / den

    # Evaluation/metrics.py:265-376
    def _em(self, ff, fb):
        n, nfg = self.n, self.area
        pf = ff + fb
        pb = n - pf
        if nfg == 0:
            tot = pb
        elif nfg == n:
            tot = pf
        else:
            bf = nfg - ff
            bb = pb - bf
            mp, mg = pf / n, nfg / n
            tot = 0
            for cnt, dp, dg in ((ff, 1 - mp, 1 - mg), (fb, 1 - mp, 0 - mg), (bf, 0 - mp, 1 - mg), (bb, 0 - mp, 0 - mg)):
                align = 2 * (dp * dg) / (dp ** 2 + dg ** 2 + _EPS)
                tot = tot + (align + 1) ** 2 / 4 * cnt
        return tot / (n - 1 + _EPS)

    def emeasure(self):
        return self._em(int(self.i[5]), int(self.i[6])), self._em(self.tp, self.fp)

    # Evaluation/metrics.py:152-262
    def smeasure(self, alpha=0.5):
        d, n, area = self.d, self.n, self.area
        if area == 0:
            return 1 - d[2] / n
        if area == n:
            return d[2] / n
        frac = area / n

        def obj(s1, c2, cnt):
            m = s1 / cnt
            return 2 * m / (m * m + 1 + np.sqrt(c2 / cnt) + _EPS)

        score = alpha * (frac * obj(d[4], d[44], area) + (1 - frac) * obj(d[6], d[45], n - area))
        cx, cy, h, w = int(self.i[3]), int(self.i[4]), self.h, self.w
        sizes = [cy * cx, cy * (w - cx), (h - cy) * cx, (h - cy) * (w - cx)]
        wts = [cx * cy / n, cy * (w - cx) / n, (h - cy) * cx / n]
        wts.append(1 - sum(wts))
        reg = 0.0
        with np.errstate(all="ignore"):
            for q in range(4):
                nq = np.float64(sizes[q])
                mp, mg = d[8 + 4 * q] / nq, d[8 + 4 * q + 2] / nq
                vp, vg, cov = d[32 + 3 * q] / (nq - 1), d[32 + 3 * q + 1] / (nq - 1), d[32 + 3 * q + 2] / (nq - 1)
                a = 4 * mp * mg * cov
                b = (mp * mp + mg * mg) * (vp + vg)
                ssim = a / (b + _EPS) if a != 0 else (1.0 if b == 0 else 0.0)
                reg = reg + wts[q] * ssim
        return max(0, score + (1 - alpha) * reg)


def image_stats(pred, gt):
    """pred, gt: (H, W) or (B, H, W), numpy or torch (any device) -> list of ImageStats.  pred = sigmoid(logits)."""
    dev = torch.device("cuda", torch.cuda.current_device())
    p = torch.as_tensor(pred).to(dev, torch.float32)
    g = torch.as_tensor(np.asarray(gt).astype(bool) if isinstance(gt, np.ndarray) else gt).to(dev)
    if p.dim() == 2:
        p, g = p[None], g[None]
    ints, dbl = hip.saliency_stats(p, g != 0)
    ints, dbl = ints.cpu().numpy(), dbl.cpu().numpy()
    return [ImageStats(ints[i], dbl[i], p.shape[1:]) for i in range(p.shape[0])]


_memo = {"key": None, "stats": None}


def _stats_for(pred, gt):
    """one launch for the five metric objects of test_one_epoch stepped with the same (pred, gt)"""
    key = (id(pred), id(gt))
    if _memo["key"] != key:
        _memo["key"], _memo["stats"] = key, image_stats(pred, gt)
        _memo["hold"] = (pred, gt)          # keep the ids alive for as long as the memo entry
    return _memo["stats"]


# ----------------------------------------------------------------------------- the reference's metric objects
class MAE:
    def __init__(self):
        self.maes = []

    def step(self, pred, gt):
        self.maes.extend(s.mae() for s in _stats_for(pred, gt))

    def get_results(self):
        return dict(mae=np.mean(np.array(self.maes, np.float64)))


class Fmeasure_and_FNR:
    def __init__(self, beta: float = 0.3):
        self.beta = beta
        self.precisions, self.recalls, self.fnrs, self.adaptive_fms, self.changeable_fms = [], [], [], [], []

    def step(self, pred, gt):
        for s in _stats_for(pred, gt):
            adp, prec, rec, curve = s.fmeasure(self.beta)
            self.adaptive_fms.append(adp)
            self.precisions.append(prec)
            self.recalls.append(rec)
            self.fnrs.append(1 - rec)
            self.changeable_fms.append(curve)

    def get_results(self):
        mean0 = lambda v: np.mean(np.array(v, dtype=np.float64), axis=0)
        return dict(fm=dict(adp=np.mean(np.array(self.adaptive_fms, np.float64)), curve=mean0(self.changeable_fms)),
                    pr=dict(p=mean0(self.precisions), r=mean0(self.recalls))), np.mean(self.fnrs, dtype=np.float64)


class Smeasure:
    def __init__(self, alpha: float = 0.5):
        self.alpha = alpha
        self.sms = []

    def step(self, pred, gt):
        self.sms.extend(s.smeasure(self.alpha) for s in _stats_for(pred, gt))

    def get_results(self):
        return dict(sm=np.mean(np.array(self.sms, dtype=np.float64)))


class Emeasure:
    def __init__(self):
        self.adaptive_ems, self.changeable_ems = [], []

    def step(self, pred, gt):
        for s in _stats_for(pred, gt):
            adp, curve = s.emeasure()
            self.adaptive_ems.append(adp)
            self.changeable_ems.append(curve)

    def get_results(self):
        return dict(em=dict(adp=np.mean(np.array(self.adaptive_ems, np.float64)),
                            curve=np.mean(np.array(self.changeable_ems, dtype=np.float64), axis=0)))


class WeightedFmeasure:
    """Evaluation/metrics.py:379-441.  Host-side (scipy distance transform), as in the reference."""

    def __init__(self, beta: float = 1):
        self.beta = beta
        self.weighted_fms = []

    def step(self, pred, gt):
        pred = pred.detach().cpu().numpy() if isinstance(pred, torch.Tensor) else np.asarray(pred)
        gt = gt.detach().cpu().numpy() if isinstance(gt, torch.Tensor) else np.asarray(gt)
        if pred.ndim == 2:
            pred, gt = pred[None], gt[None]
        for p, g in zip(pred, gt):
            self.weighted_fms.append(self._one(p, g.astype(bool)))

    def _one(self, pred, gt):
        from scipy.ndimage import convolve, distance_transform_edt
        lo, hi = pred.min(), pred.max()
        if hi != lo:
            pred = (pred - lo) / (hi - lo)
        if not gt.any():
            return 0
        dist, idx = distance_transform_edt(~gt, return_indices=True)
        err = np.abs(pred - gt)
        near = err.copy()
        near[~gt] = err[idx[0][~gt], idx[1][~gt]]
        r = np.arange(-3, 4, dtype=np.float64)
        ker = np.exp(-(r[:, None] ** 2 + r[None, :] ** 2) / 50.0)
        ker[ker < np.finfo(ker.dtype).eps * ker.max()] = 0
        ker /= ker.sum()
        smooth = convolve(near, weights=ker, mode="constant", cval=0)
        ew = np.where(gt & (smooth < err), smooth, err) * np.where(~gt, 2 - np.exp(np.log(0.5) / 5 * dist), np.ones_like(gt))
        tp = np.sum(gt) - np.sum(ew[gt])
        rec = 1 - np.mean(ew[gt])
        prec = tp / (tp + np.sum(ew[~gt]) + _EPS)
        return (1 + self.beta) * rec * prec / (rec + self.beta * prec + _EPS)

    def get_results(self):
        return dict(wfm=np.mean(np.array(self.weighted_fms, dtype=np.float64)))


# ----------------------------------------------------------------------------- loops
def _forward_fn(model, graph):
    """`model` itself, or its hipGraph replay (tramba_amd/graph.py) when `graph` is set: the weights are fixed for the
    duration of an evaluation pass, so a fresh capture per pass is safe."""
    if not graph:
        return model
    from .graph import GraphedForward
    return GraphedForward(model)


def test_one_epoch(model, batches, methods="SOD", weighted=True, graph=False):
    """train.py:101-152.  `batches` yields dicts with 'image' (B,3,S,S) and 'gt' (B,1,S,S) (what RGB_Dataset's
    loader yields; the reference uses B = 1, any B works here).  Returns the reference's results dictionary.
    `graph=True` replays the forward as a hipGraph (one capture per input shape, same kernels, same results)."""
    fm, wfm, sm, em, mae = Fmeasure_and_FNR(), WeightedFmeasure(), Smeasure(), Emeasure(), MAE()
    dev = next(model.parameters()).device
    was_training = model.training
    model.eval()
    forward = _forward_fn(model, graph)
    with torch.no_grad():
        for batch in batches:
            images = batch["image"].to(dev, non_blocking=True)
            gt = batch["gt"].to(dev).reshape(images.shape[0], *batch["gt"].shape[-2:]) != 0
            pred = torch.sigmoid(forward(images)[-1].float()).reshape(gt.shape)
            for m in (fm, sm, em, mae):
                m.step(pred=pred, gt=gt)
            if weighted:
                wfm.step(pred=pred, gt=gt)
    model.train(was_training)
    f, fnr = fm.get_results()
    e = em.get_results()["em"]
    r4 = lambda v: np.round(v, 4)
    return {
        "dataset_setname": methods,
        "Smeasure_r": r4(sm.get_results()["sm"]),
        "Wmeasure_r": r4(wfm.get_results()["wfm"]) if weighted else None,
        "MAE_r": r4(mae.get_results()["mae"]),
        "adpEm_r": r4(e["adp"]), "meanEm_r": r4(e["curve"].mean()), "maxEm_r": r4(e["curve"].max()),
        "adpFm_r": r4(f["fm"]["adp"]), "meanFm_r": r4(f["fm"]["curve"].mean()), "maxFm_r": r4(f["fm"]["curve"].max()),
        "fnr_r": r4(fnr),
    }


def write_png_gray8(path, img: np.ndarray):
    """8-bit greyscale PNG (what cv2.imwrite produces for a uint8 (H, W) array, test_TSOD.py:66-68), stdlib only."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    raw = np.concatenate([np.zeros((h, 1), np.uint8), img], axis=1).tobytes()      # filter type 0 per scanline

    def chunk(tag, data):
        body = tag + data
        return struct.pack(">I", len(data)) + body + struct.pack(">I", zlib.crc32(body) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def save_predictions(model, batches, save_path, graph=False):
    """test_TSOD.py:46-68: forward, bilinear resize of the full-resolution logits to the image's original size
    (`shape` = (W, H) as the loader reports it), sigmoid, *255 -> uint8, <name>.png."""
    os.makedirs(save_path, exist_ok=True)
    dev = next(model.parameters()).device
    model.eval()
    forward = _forward_fn(model, graph)
    written = []
    with torch.no_grad():
        for batch in batches:
            res = forward(batch["image"].to(dev))[-1].float()
            for i in range(res.shape[0]):
                shape, name = batch["shape"], batch["name"]
                wd, ht = (int(shape[0][i]), int(shape[1][i])) if isinstance(shape[0], (list, tuple, torch.Tensor)) \
                    and len(shape[0]) > 1 else (int(shape[0]), int(shape[1]))
                pred = F.interpolate(res[i:i + 1], size=(ht, wd), mode="bilinear", align_corners=False)
                pred = (torch.sigmoid(pred)[0, 0] * 255).to(torch.uint8).cpu().numpy()
                out = os.path.join(save_path, name[i] + ".png")
                write_png_gray8(out, pred)
                written.append(out)
    return written
