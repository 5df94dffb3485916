"""Training-step pieces of the reference's train.py, device-resident (no .cpu() sync per step).

loss:       train.py:76-85 -- every output bilinearly resized to the label size, then
            binary_cross_entropy_with_logits + iou_loss (utils/loss.py:6-11), weights 1.
optimizer:  train.py:266-280 -- Adam, parameters whose name contains "encoder" at 0.1 x lr.
lr decay:   utils/lr.py:1-17.
epoch loop, resume / best-MAE checkpoint files: train.py:212-263.
"""
import os

import torch
import torch.nn.functional as F

from . import hip
from .modules import refresh_lowp_shadows


def iou_loss(pred, mask):
    """utils/loss.py:6-11."""
    pred = torch.sigmoid(pred)
    inter = (pred * mask).sum(dim=(2, 3))
    union = (pred + mask).sum(dim=(2, 3))
    return (1 - (inter + 1) / (union - inter + 1)).mean()


class _UpsampleBilinearHIP(torch.autograd.Function):
    """F.interpolate(x, size, mode="bilinear") with the library's backward (a gather per input pixel; torch's scatters with
    float atomics, 183 us per deep-supervision output at batch 8)."""

    @staticmethod
    def forward(ctx, x, size):
        ctx.in_hw = tuple(x.shape[-2:])
        return F.interpolate(x, size, mode="bilinear")

    @staticmethod
    def backward(ctx, g):
        from . import hip
        return hip.upsample_bilinear_bwd(g, *ctx.in_hw), None


def _resize_bilinear(o, size):
    """train.py:78-79: the deep-supervision outputs resized to the label"""
    if o.is_cuda and o.dtype == torch.float32 and o.requires_grad and size[0] >= o.shape[-2] and size[1] >= o.shape[-1]:
        return _UpsampleBilinearHIP.apply(o, tuple(size))
    return F.interpolate(o, size, mode="bilinear")


def tramba_loss(outputs, label, loss_weights=None):
    """Sum over the deep-supervision outputs (3 for Tramba-R, 4 otherwise) of BCE-with-logits + IoU."""
    h, w = label.shape[-2:]
    total = None
    for i, o in enumerate(outputs):
        o = o.float()
        if o.shape[-2:] != (h, w):
            o = _resize_bilinear(o, (h, w))
        term = F.binary_cross_entropy_with_logits(o, label) + iou_loss(o, label)
        if loss_weights is not None:
            term = term * loss_weights[i]
        total = term if total is None else total + term
    return total


def get_opt(lr, model, capturable=False):
    """train.py:266-280: two Adam groups, encoder parameters at lr/10.  `capturable`: step counters on the device, so
    that the whole step can be replayed as a hipGraph (tramba_amd.graph.GraphedTrainStep).  On the GPU the update runs as
    torch's single-pass multi-tensor Adam (`fused=True`: the same arithmetic and state_dict as the reference's default
    optimizer, one read and one write of p / exp_avg / exp_avg_sq instead of ten passes over the 446 MB of each)."""
    base = [p for n, p in model.named_parameters() if "encoder" in n]
    other = [p for n, p in model.named_parameters() if "encoder" not in n]
    fused = all(p.is_cuda and p.is_floating_point() for p in base + other) and len(base + other) > 0
    return torch.optim.Adam([{"params": base, "lr": lr * 0.1}, {"params": other, "lr": lr}], lr, capturable=capturable,
                            fused=True if fused else None)


def adjust_learning_rate(optimizer, epoch, decay_epochs, base_lr, decay_factors):
    """utils/lr.py:1-17: at a listed epoch set lr = base_lr * factor (encoder group at a tenth)."""
    assert len(decay_epochs) == len(decay_factors)
    if epoch in decay_epochs:
        f = decay_factors[decay_epochs.index(epoch)]
        optimizer.param_groups[1]["lr"] = base_lr * f
        optimizer.param_groups[0]["lr"] = base_lr * f * 0.1
    return optimizer.param_groups[1]["lr"]


DEFER_SUMS = True   # (scripts/graph_train.py times the step both ways)


def train_step(model, opt, images, label, reducer=None):
    """One optimisation step (train.py:74-89).  `reducer` (tramba_amd.parallel.GradBucketReducer)
    averages gradients across data-parallel ranks; its all-reduces overlap the backward."""
    outputs = model(images)
    loss = tramba_loss(outputs, label)
    if reducer is not None:
        reducer.prepare()
    else:
        opt.zero_grad(set_to_none=True)
    if DEFER_SUMS:
        with hip.deferred_sums():  # the parameter-gradient partial sums of the pass run as a few batched launches at the exit
            loss.backward()
    else:
        loss.backward()
    if reducer is not None:
        reducer.finish()
    opt.step()
    refresh_lowp_shadows(model, getattr(model, "compute_dtype", None))   # next forward's bf16 weights: one fused cast
    return loss.detach()


# ----------------------------------------------------------------------------- checkpoints / epoch loop
# File formats and names of the reference's `fit` (train.py:212-263), so checkpoints interoperate both ways:
#   <save_model>/<method>/<method>_resume.pth           {"model": state_dict, "optimizer": state_dict, "epoch": e}
#   <save_model>/<method>/<method>_MAE_<mae>_<e+1>.pth   bare state_dict of a best-MAE epoch
def _ckpt_dir(save_model, method):
    return os.path.join(save_model, method)


def save_resume(save_model, method, model, opt, epoch):
    """train.py:254-262 (written every 5th epoch there)."""
    d = _ckpt_dir(save_model, method)
    os.makedirs(d, exist_ok=True)
    path = os.path.join(d, f"{method}_resume.pth")
    torch.save({"model": model.state_dict(), "optimizer": opt.state_dict(), "epoch": epoch}, path)
    return path


def save_best(save_model, method, model, mae, epoch):
    """train.py:250-253: `<method>_MAE_<mae>_<epoch+1>.pth` holds the bare state_dict."""
    d = _ckpt_dir(save_model, method)
    os.makedirs(d, exist_ok=True)
    path = os.path.join(d, f"{method}_MAE_{mae}_{epoch + 1}.pth")
    torch.save(model.state_dict(), path)
    return path


def load_resume(resume, save_model, method, model, opt, map_location=None):
    """train.py:214-229.  resume=None -> 0; "last" -> the resume file (model + optimizer, continues at epoch+1);
    anything else is a state_dict path whose file name ends in `_<epoch>.pth` (continues at that number)."""
    if resume is None:
        return 0
    if resume == "last":
        ck = torch.load(os.path.join(_ckpt_dir(save_model, method), f"{method}_resume.pth"), map_location=map_location)
        model.load_state_dict(ck["model"], strict=True)
        opt.load_state_dict(ck["optimizer"])
        return ck["epoch"] + 1
    model.load_state_dict(torch.load(resume, map_location=map_location), strict=True)
    return int(os.path.basename(resume).split("_")[-1].split(".")[0])


def fit(model, opt, batches, epochs, base_lr, decay_epochs, decay_factors, save_model, method, start_epoch=0,
        evaluate=None, see=0, best_mae=None, reducer=None, is_main=True, log=None, graph=False):
    """Epoch loop of train.py:212-263 around `train_step`.  `batches(epoch)` yields (images, label) device tensors;
    `evaluate(model, epoch) -> MAE` runs from epoch `see` on (train.py:237); rank 0 (`is_main`) writes the files.
    `graph=True` (single process, optimizer from `get_opt(..., capturable=True)`): every step is a hipGraph replay
    (tramba_amd.graph.GraphedTrainStep), re-captured by itself when the learning rate steps."""
    step_fn = train_step
    if graph:
        from .graph import GraphedTrainStep
        graphed = GraphedTrainStep(model, opt, reducer=reducer)
        step_fn = lambda m_, o_, images, label, reducer=None: graphed(images, label)  # noqa: E731
    history = []
    for epoch in range(start_epoch, epochs):
        lr = adjust_learning_rate(opt, epoch, decay_epochs, base_lr, decay_factors)
        total, n = None, 0
        for images, label in batches(epoch):
            loss = step_fn(model, opt, images, label, reducer=reducer)
            total = loss.clone() if total is None else total + loss     # clone: a graphed step reuses its loss buffer
            n += 1
        mean_loss = float(total / max(n, 1)) if total is not None else float("nan")   # one host sync per epoch
        mae = None
        if evaluate is not None and epoch + 1 >= see:
            mae = evaluate(model, epoch)
            if is_main and (best_mae is None or mae < best_mae):
                save_best(save_model, method, model, mae, epoch)
        if is_main and (epoch + 1) % 5 == 0:
            save_resume(save_model, method, model, opt, epoch)
        history.append({"epoch": epoch, "lr": lr, "loss": mean_loss, "mae": mae})
        if log is not None:
            log(history[-1])
    return history
