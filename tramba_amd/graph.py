"""hipGraph replay of an inference forward.

A Tramba-V forward is ~400 dependent launches of 5-100 us each: run eagerly, Python and the launch path cost more than
the kernels at batch 1.  `GraphedForward` captures `model(x)` once per input shape into a hipGraph
(after eager warm-up passes, which also fill the per-module scan-table / weight caches on the capturing thread) and
replays it: one launch call per image.  This is the deployment form of the reference's evaluation loops
(test_TSOD.py:53-64, train.py:116-125), which call `model(images)` at batch 1 in a Python loop.

The graph holds raw pointers to the weights and to the low-precision weight shadows: capture AFTER loading a checkpoint
and build a new `GraphedForward` (or call `reset()`) whenever the weights change (e.g. per evaluation epoch in training).
Outputs are the graph's static buffers, valid until the next call with the same input shape; clone what must outlive it.
"""
import torch


class GraphedForward:
    def __init__(self, model, warmup=2, strict=False):
        if model.training:
            raise RuntimeError("GraphedForward captures an inference forward: call model.eval() first")
        self.model = model
        self.warmup = warmup
        self.strict = strict          # True: a failed capture raises instead of falling back to eager launches
        self._graphs = {}

    def reset(self):
        self._graphs.clear()

    def _capture(self, x):
        static_in = x.clone()
        with torch.no_grad():
            for _ in range(self.warmup):
                self.model(static_in)
        torch.cuda.synchronize(x.device)
        graph = torch.cuda.CUDAGraph()
        try:
            with torch.no_grad(), torch.cuda.graph(graph):
                out = self.model(static_in)
        except Exception:
            if self.strict:
                raise
            torch.cuda.synchronize(x.device)
            return None
        return graph, static_in, out

    def __call__(self, x):
        if self.model.training:
            raise RuntimeError("GraphedForward: the model was switched back to training mode")
        if not x.is_cuda:
            raise RuntimeError("GraphedForward needs a device tensor (there is no CPU path)")
        key = (tuple(x.shape), x.dtype, x.device)
        if key not in self._graphs:
            self._graphs[key] = self._capture(x)
        entry = self._graphs[key]
        if entry is None:                              # capture unavailable: same kernels, launched eagerly
            with torch.no_grad():
                return self.model(x)
        graph, static_in, out = entry
        static_in.copy_(x, non_blocking=True)
        graph.replay()
        return out


class GraphedTrainStep:
    """One optimisation step (tramba_amd.train.train_step: forward, loss, backward, Adam, weight-shadow refresh) captured
    as ONE hipGraph and replayed per batch -- the eager step issues ~6000 launches from Python.

    Requirements: an optimizer built with `capturable=True` (`train.get_opt(lr, model, capturable=True)`: Adam's step
    counters live on the device).  With a data-parallel `reducer` (world size > 1) the bucketed RCCL all-reduces launched
    from the autograd hooks are captured with the step (RCCL's stream joins the capture through the usual event edges) and
    replayed as graph nodes: every rank must capture and replay the same sequence (same batch shapes per step), and the
    reducer must not use `find_unused` (its host read of the flag vector cannot be captured).  One graph per batch shape (a short last batch of an epoch gets its
    own).  Learning rates are baked into the captured kernels: the graphs are dropped and re-captured when
    `adjust_learning_rate` changes them.  A capture needs eager warm-up steps (optimizer state and lazy caches must exist
    before the stream is captured); parameters and optimizer state are saved before and restored after them, so every call
    -- capturing or not -- advances the optimisation by exactly one step on the batch it was given.  Stochastic depth draws
    from torch's device generator, which hipGraph capture advances per replay.  Returns the loss (a static buffer)."""

    def __init__(self, model, opt, reducer=None, warmup=2):
        if reducer is not None and getattr(reducer, "world", 1) > 1 and getattr(reducer, "find_unused", False):
            raise RuntimeError("GraphedTrainStep: a reducer with find_unused=True reads flags on the host every step and "
                               "cannot be captured")
        if not all(g.get("capturable", False) for g in opt.param_groups):
            raise RuntimeError("GraphedTrainStep needs an optimizer with capturable=True")
        self.model, self.opt, self.reducer, self.warmup = model, opt, reducer, warmup
        self._graphs = {}
        self._lr_key = None
        self._trainable_key = None

    def _lrs(self):
        return tuple(float(g["lr"]) for g in self.opt.param_groups)

    def _snapshot(self):
        params = [p for g in self.opt.param_groups for p in g["params"]]
        saved = []
        for p in params:
            st = self.opt.state.get(p)
            saved.append((p, p.detach().clone(),
                          None if not st else {k: v.clone() for k, v in st.items() if torch.is_tensor(v)}))
        return saved, [(b, b.detach().clone()) for b in self.model.buffers()]     # buffers: BatchNorm running statistics

    def _restore(self, snapshot):
        from .modules import refresh_dw_packs, refresh_lowp_shadows
        saved, buffers = snapshot
        with torch.no_grad():
            for b, value in buffers:
                b.copy_(value)
            for p, value, st in saved:
                p.copy_(value)
                for k, v in self.opt.state.get(p, {}).items():
                    if torch.is_tensor(v):               # in place: the graph holds pointers to these tensors
                        if st is not None and k in st:
                            v.copy_(st[k])
                        else:
                            v.zero_()                    # state created by the warm-up: back to "never stepped"
        refresh_lowp_shadows(self.model, getattr(self.model, "compute_dtype", None))
        if any(p.is_cuda for p in self.model.parameters()):
            refresh_dw_packs(self.model)

    def _touched(self):
        """Every tensor a replay rewrites in place behind autograd's back: parameters, optimizer state, buffers."""
        ts = [p for g in self.opt.param_groups for p in g["params"]]
        ts += [v for p in list(ts) for v in self.opt.state.get(p, {}).values() if torch.is_tensor(v)]
        ts += list(self.model.buffers())
        return ts

    def _mark_modified(self):
        """A graph replay updates the weights without bumping their version counters, and every derived-weight cache of
        the inference path (packed stencils, -exp(A_logs), fp32 views, head biases, the low-precision shadows) is keyed
        on those counters: bump them, then re-stamp the shadows the replay has just refreshed as current."""
        from .modules import restamp_lowp_shadows
        torch.autograd.graph.increment_version(self._touched())
        restamp_lowp_shadows(self.model)

    def _capture(self, images, label):
        from .train import train_step
        sx, sy = images.clone(), label.clone()
        saved = self._snapshot()
        dev = images.device
        rng = torch.cuda.get_rng_state(dev)                 # the warm-up steps draw stochastic-depth masks: undone below
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                       # optimizer state and all lazy caches exist before capture
            for _ in range(self.warmup):
                train_step(self.model, self.opt, sx, sy, reducer=self.reducer)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        from .modules import model_mask_pool
        pool = model_mask_pool(self.model)
        pool.forget_draw()                                  # the step's one stochastic-depth draw must be IN the graph
        graph = torch.cuda.CUDAGraph()
        # With collectives in the step, ProcessGroupNCCL's watchdog / heartbeat threads keep making runtime calls (event
        # queries on the warm-up steps' work objects) while this thread records: under the default "global" capture mode
        # any such call from ANOTHER thread invalidates the capture.  "thread_local" flags only this thread's own unsafe
        # calls -- what PyTorch documents for capturing NCCL collectives.
        mode = "thread_local" if (self.reducer is not None and getattr(self.reducer, "world", 1) > 1) else "global"
        with torch.cuda.graph(graph, capture_error_mode=mode):   # records, executes nothing
            loss = train_step(self.model, self.opt, sx, sy, reducer=self.reducer)
        keep_alive = (pool.probs, pool.buf, pool.buf32)              # the keep-probabilities the captured bernoulli / divide nodes read
        pool.forget_draw()                                  # (the table drawn during capture lives in the graph's pool)
        self._restore(saved)
        torch.cuda.set_rng_state(rng, dev)
        return graph, sx, sy, loss, keep_alive

    def __call__(self, images, label):
        if not self.model.training:
            raise RuntimeError("GraphedTrainStep: the model is in eval mode (an evaluation callback must switch it back "
                               "with model.train() before the next step)")
        # learning rates AND the set of trainable parameters are baked into a capture (freeze_encoder / unfreeze_encoder
        # flip requires_grad: the eager step follows them, a stale graph would keep updating frozen weights)
        lrs, trainable = self._lrs(), tuple(p.requires_grad for g in self.opt.param_groups for p in g["params"])
        if lrs != self._lr_key or trainable != self._trainable_key:
            self._graphs.clear()
            self._lr_key, self._trainable_key = lrs, trainable
        key = (tuple(images.shape), images.dtype, tuple(label.shape), label.dtype)
        entry = self._graphs.get(key)
        if entry is None:
            entry = self._graphs[key] = self._capture(images, label)
        graph, sx, sy, loss = entry[:4]
        sx.copy_(images, non_blocking=True)
        sy.copy_(label, non_blocking=True)
        graph.replay()
        self._mark_modified()
        return loss
