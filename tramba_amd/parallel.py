"""Data parallelism for the Tramba training step: gradients only, one process per GPU.

The reference has no distributed code (`--parallel` only prints, run.py:46-50).  The path shards by
image, so the single exchange step is an all-reduce(SUM)/world of the gradients.  Design for xGMI
(point-to-point links, ring collectives are per-link bound): few large flat buckets (default 32 MB)
instead of one call per tensor, launched from autograd hooks in reverse registration order so RCCL
works on bucket i while backward still produces bucket i+1; one flat buffer per bucket and the
gradients are VIEWS into it (no gather/scatter copies).  Backend: "nccl" (= RCCL) on GPUs, "gloo" in
the CPU tests.
"""
import torch
import torch.distributed as dist


def broadcast_parameters(model, src=0, process_group=None):
    """Seed-synchronised replicas: rank `src`'s parameters and buffers overwrite everyone's."""
    for t in list(model.parameters()) + list(model.buffers()):
        dist.broadcast(t.data, src=src, group=process_group)


class GradBucketReducer:
    def __init__(self, model, bucket_mb=32.0, process_group=None):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # backward produces gradients roughly in reverse parameter order: bucket in that order
        params = [p for p in model.parameters() if p.requires_grad][::-1]
        cap = int(bucket_mb * 1024 * 1024)
        self.buckets, cur, cur_bytes = [], [], 0
        for p in params:
            nbytes = p.numel() * p.element_size()
            if cur and (cur_bytes + nbytes > cap or p.dtype != cur[0].dtype or p.device != cur[0].device):
                self.buckets.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            self.buckets.append(cur)
        self.flat, self._where = [], {}
        for bi, bucket in enumerate(self.buckets):
            flat = torch.zeros(sum(p.numel() for p in bucket), dtype=bucket[0].dtype, device=bucket[0].device)
            off = 0
            for p in bucket:
                self._where[p] = (bi, flat[off:off + p.numel()].view_as(p))
                off += p.numel()
            self.flat.append(flat)
        self._pending = [0] * len(self.buckets)
        self._handles = []
        for p in self._where:
            p.register_post_accumulate_grad_hook(self._on_grad)
        self.prepare()

    def prepare(self):
        """Zero the buckets and point every .grad at its slice; call before each backward
        (it replaces optimizer.zero_grad())."""
        self._handles = []
        for bi, bucket in enumerate(self.buckets):
            self.flat[bi].zero_()
            self._pending[bi] = len(bucket)
            for p in bucket:
                p.grad = self._where[p][1]

    def _on_grad(self, p):
        bi, view = self._where[p]
        if p.grad is not view:  # autograd installed a fresh tensor: move it into the bucket
            view.copy_(p.grad)
            p.grad = view
        self._pending[bi] -= 1
        if self._pending[bi] == 0 and self.world > 1:
            self._handles.append(dist.all_reduce(self.flat[bi], op=dist.ReduceOp.SUM, group=self.group,
                                                 async_op=True))

    def finish(self):
        """Wait for the outstanding all-reduces, also reduce buckets whose hooks did not all fire
        (parameters unused in this step), then average."""
        if self.world > 1:
            for bi, left in enumerate(self._pending):
                if left > 0:
                    self._handles.append(dist.all_reduce(self.flat[bi], op=dist.ReduceOp.SUM, group=self.group,
                                                         async_op=True))
                    self._pending[bi] = 0
            for h in self._handles:
                h.wait()
            for flat in self.flat:
                flat.div_(self.world)
        self._handles = []

    def bytes_per_step(self):
        return sum(f.numel() * f.element_size() for f in self.flat)
