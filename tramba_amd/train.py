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
from .modules import refresh_dw_packs, refresh_lowp_shadows


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


class _SodLossHIP(torch.autograd.Function):
    """The whole loss of train.py:76-85 in the library: three sums per image and output in one pass each (the resized logit
    map is never stored), one finishing block, and in the backward one gather per output that lands on the output's own
    resolution.  ~10 launches where the framework issued ~140."""

    @staticmethod
    def forward(ctx, label, weights, *outs):
        loss, coefs = hip.sod_loss(outs, label, weights)
        ctx.save_for_backward(label, coefs, *outs)
        return loss

    @staticmethod
    def backward(ctx, gl):
        label, coefs, *outs = ctx.saved_tensors
        gl = gl.to(torch.float32).contiguous()
        return (None, None) + tuple(hip.sod_loss_grad(o, label, coefs[i], gl) if ctx.needs_input_grad[2 + i] else None
                                    for i, o in enumerate(outs))


def _loss_on_device(outputs, label):
    """the library's loss takes logit maps no larger than the label, plane for plane"""
    # (a resized output's gradient kernel holds rows of the label in LDS: labels wider than 4096 only with outputs at label size)
    return (label.is_cuda and label.dim() == 4 and 0 < len(outputs) <= 8 and label.shape[0] * label.shape[1] <= 512
            and all(o.is_cuda and o.dim() == 4 and o.shape[:2] == label.shape[:2] and o.shape[-2] <= label.shape[-2]
                    and o.shape[-1] <= label.shape[-1] for o in outputs)
            and (label.shape[-1] <= 4096 or all(o.shape[-2:] == label.shape[-2:] for o in outputs)))


def tramba_loss(outputs, label, loss_weights=None):
    """Sum over the deep-supervision outputs (3 for Tramba-R, 4 otherwise) of BCE-with-logits + IoU."""
    outputs = list(outputs)
    if loss_weights is not None and len(loss_weights) != len(outputs):
        raise ValueError(f"tramba_loss: {len(loss_weights)} loss weights for {len(outputs)} outputs")
    if _loss_on_device(outputs, label):
        weights = None if loss_weights is None else tuple(float(w) for w in loss_weights)
        return _SodLossHIP.apply(label.float().contiguous(), weights, *[o.float().contiguous() for o in outputs])
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


class Adam(torch.optim.Adam):
    """torch.optim.Adam -- constructor, param_groups, per-parameter state {step, exp_avg, exp_avg_sq} and state_dict as the
    reference's optimizer (train.py:266-280) -- whose step() is the library's multi-tensor kernel (tramba_adam_step): one
    read of the gradient and one read + write of p / exp_avg / exp_avg_sq at HBM speed (torch's `fused=True` form ran the
    111 M parameters of Tramba-V at 2.7 TB/s in 27 launches), the bias corrections in fp64 as torch computes them.  The step
    counters live on the device (what torch does for `capturable` / `fused`), so a step can be recorded into a hipGraph.
    fp32 parameters on a HIP device only; amsgrad / maximize / differentiable are not implemented (the reference uses none)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, capturable=False):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, capturable=capturable)
        self._plans = {}

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._plans = {}          # the state tensors have been replaced

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        self._plans = {}

    def _plan(self, gi, ps):
        """pointer arrays of the parameters and their state for group gi: rebuilt when the set of parameters with a
        gradient or an address changes (the state tensors only change through load_state_dict)"""
        ptrs = [p.data_ptr() for p in ps]
        plan = self._plans.get(gi)
        if plan is not None and plan[0] == ptrs:
            # the state entries may have been replaced behind the optimizer's back (opt.state.clear(), a hand-made restore):
            # the plan must never keep updating orphaned tensors
            ms, vs, steps = plan[6]
            if all((st := self.state.get(p)) and st.get("exp_avg") is m and st.get("exp_avg_sq") is v and st.get("step") is t
                   for p, m, v, t in zip(ps, ms, vs, steps)):
                return plan
        ms, vs, steps = [], [], []
        for p in ps:
            if p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous():
                raise hip.TrambaHipError("Adam: contiguous fp32 parameters on a HIP device only (no CPU fallback)")
            st = self.state[p]
            if len(st) == 0:
                st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            if not torch.is_tensor(st["step"]) or st["step"].device != p.device or st["step"].dtype != torch.float32:
                st["step"] = torch.as_tensor(st["step"], dtype=torch.float32).to(p.device).reshape(())   # a non-capturable checkpoint
            for k in ("exp_avg", "exp_avg_sq"):
                if st[k].dtype != torch.float32 or st[k].device != p.device or not st[k].is_contiguous() or st[k].numel() != p.numel():
                    raise hip.TrambaHipError(f"Adam: state '{k}' does not match its parameter {tuple(p.shape)}")
            ms.append(st["exp_avg"])
            vs.append(st["exp_avg_sq"])
            steps.append(st["step"])
        import ctypes
        n = len(ps)
        plan = (ptrs, hip.pointer_array(ps), hip.pointer_array(ms), hip.pointer_array(vs), hip.pointer_array(steps),
                (ctypes.c_int64 * n)(*[p.numel() for p in ps]), (ms, vs, steps))
        self._plans[gi] = plan
        return plan

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            if group.get("amsgrad") or group.get("maximize") or group.get("differentiable"):
                raise hip.TrambaHipError("Adam: amsgrad / maximize / differentiable are not implemented")
            ps = [p for p in group["params"] if p.grad is not None and p.numel() > 0]
            if not ps:
                continue
            grads = [p.grad for p in ps]
            for p, g in zip(ps, grads):
                if g.dtype != torch.float32 or g.layout != torch.strided or not g.is_contiguous() or g.numel() != p.numel():
                    raise hip.TrambaHipError("Adam: dense contiguous fp32 gradients only")
            plan = self._plan(gi, ps)
            b1, b2 = group["betas"]
            hip.adam_step_raw(plan[1], hip.pointer_array(grads), plan[2], plan[3], plan[4], plan[5], len(ps), group["lr"], b1, b2,
                              group["eps"], group["weight_decay"])
            # the kernel writes through raw pointers: bump the version counters as an in-place torch op would (every
            # derived-weight cache of the inference path is keyed on them)
            torch.autograd.graph.increment_version(ps)
        return loss


def get_opt(lr, model, capturable=False):
    """train.py:266-280: two Adam groups, encoder parameters at lr/10.  `capturable`: the whole step can be replayed as a
    hipGraph (tramba_amd.graph.GraphedTrainStep).  On the GPU the update is the library's one-pass multi-tensor kernel
    (`Adam` above: the same arithmetic and state_dict as the reference's default optimizer)."""
    base = [p for n, p in model.named_parameters() if "encoder" in n]
    other = [p for n, p in model.named_parameters() if "encoder" not in n]
    groups = [{"params": base, "lr": lr * 0.1}, {"params": other, "lr": lr}]
    if len(base + other) > 0 and all(p.is_cuda and p.dtype == torch.float32 for p in base + other):
        return Adam(groups, lr, capturable=capturable)
    return torch.optim.Adam(groups, lr, capturable=capturable)


def adjust_learning_rate(optimizer, epoch, decay_epochs, base_lr, decay_factors):
    """utils/lr.py:1-17: at a listed epoch set lr = base_lr * factor (encoder group at a tenth)."""
    assert len(decay_epochs) == len(decay_factors)
    if epoch in decay_epochs:
        f = decay_factors[decay_epochs.index(epoch)]
        optimizer.param_groups[1]["lr"] = base_lr * f
        optimizer.param_groups[0]["lr"] = base_lr * f * 0.1
    return optimizer.param_groups[1]["lr"]


DEFER_SUMS = True   # (scripts/graph_train.py times the step both ways)


def _has_standing_grads(opt):
    """any parameter of the optimizer whose .grad is set (backward would accumulate into it)"""
    for g in opt.param_groups:
        for p in g["params"]:
            if p.grad is not None:
                return True
    return False


def train_step(model, opt, images, label, reducer=None):
    """One optimisation step (train.py:74-89).  `reducer` (tramba_amd.parallel.GradBucketReducer)
    averages gradients across data-parallel ranks; its all-reduces overlap the backward."""
    outputs = model(images)
    loss = tramba_loss(outputs, label)
    if reducer is not None:
        reducer.prepare()
    else:
        opt.zero_grad(set_to_none=True)
    # Deferred partial sums hand autograd gradient tensors that are FILLED at the exit of the context: safe only where the engine
    # stores them as they are -- an fp32 leaf whose .grad is None (the call sites defer for fp32 parameters only; a gradient that is
    # already set would be accumulated into, i.e. read, before the flush: gradient accumulation, zero_grad(set_to_none=False))
    if DEFER_SUMS and not _has_standing_grads(opt):
        with hip.deferred_sums():  # the parameter-gradient partial sums of the pass run as a few batched launches at the exit
            loss.backward()
    else:
        loss.backward()
    if reducer is not None:
        reducer.finish()
    opt.step()
    refresh_lowp_shadows(model, getattr(model, "compute_dtype", None))   # next forward's bf16 weights: one fused cast
    if images.is_cuda:
        refresh_dw_packs(model)                                          # ... and its packed depth-wise stencils: one launch
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
