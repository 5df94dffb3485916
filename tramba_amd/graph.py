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
