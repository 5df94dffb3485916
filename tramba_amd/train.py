"""Training-step pieces of the reference's train.py, device-resident (no .cpu() sync per step).

loss:       train.py:76-85 -- every output bilinearly resized to the label size, then
            binary_cross_entropy_with_logits + iou_loss (utils/loss.py:6-11), weights 1.
optimizer:  train.py:266-280 -- Adam, parameters whose name contains "encoder" at 0.1 x lr.
lr decay:   utils/lr.py:1-17.
"""
import torch
import torch.nn.functional as F


def iou_loss(pred, mask):
    """utils/loss.py:6-11."""
    pred = torch.sigmoid(pred)
    inter = (pred * mask).sum(dim=(2, 3))
    union = (pred + mask).sum(dim=(2, 3))
    return (1 - (inter + 1) / (union - inter + 1)).mean()


def tramba_loss(outputs, label, loss_weights=None):
    """Sum over the deep-supervision outputs (3 for Tramba-R, 4 otherwise) of BCE-with-logits + IoU."""
    h, w = label.shape[-2:]
    total = None
    for i, o in enumerate(outputs):
        o = o.float()
        if o.shape[-2:] != (h, w):
            o = F.interpolate(o, (h, w), mode="bilinear")
        term = F.binary_cross_entropy_with_logits(o, label) + iou_loss(o, label)
        if loss_weights is not None:
            term = term * loss_weights[i]
        total = term if total is None else total + term
    return total


def get_opt(lr, model):
    """train.py:266-280: two Adam groups, encoder parameters at lr/10."""
    base = [p for n, p in model.named_parameters() if "encoder" in n]
    other = [p for n, p in model.named_parameters() if "encoder" not in n]
    return torch.optim.Adam([{"params": base, "lr": lr * 0.1}, {"params": other, "lr": lr}], lr)


def adjust_learning_rate(optimizer, epoch, decay_epochs, base_lr, decay_factors):
    """utils/lr.py:1-17: at a listed epoch set lr = base_lr * factor (encoder group at a tenth)."""
    assert len(decay_epochs) == len(decay_factors)
    if epoch in decay_epochs:
        f = decay_factors[decay_epochs.index(epoch)]
        optimizer.param_groups[1]["lr"] = base_lr * f
        optimizer.param_groups[0]["lr"] = base_lr * f * 0.1
    return optimizer.param_groups[1]["lr"]


def train_step(model, opt, images, label, reducer=None):
    """One optimisation step (train.py:74-89).  `reducer` (tramba_amd.parallel.GradBucketReducer)
    averages gradients across data-parallel ranks; its all-reduces overlap the backward."""
    outputs = model(images)
    loss = tramba_loss(outputs, label)
    if reducer is not None:
        reducer.prepare()
    else:
        opt.zero_grad(set_to_none=True)
    loss.backward()
    if reducer is not None:
        reducer.finish()
    opt.step()
    return loss.detach()
