"""Data parallelism for the Tramba training step: gradients only, one process per GPU.

The reference has no distributed code (`--parallel` only prints, run.py:46-50).  The path shards by
image, so the single exchange step is an all-reduce(SUM)/world of the gradients.  Design for xGMI
(point-to-point links, ring collectives are per-link bound): few large flat buckets (default 32 MB)
instead of one call per tensor, launched from autograd hooks in reverse registration order so RCCL
works on bucket i while backward still produces bucket i+1; one flat buffer per bucket, filled by one
multi-tensor copy when the bucket's last gradient arrives, after which the .grad fields are VIEWS into it.  Backend: "nccl" (= RCCL) on GPUs, "gloo" in
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
        """Call before each backward (it replaces optimizer.zero_grad()).  Gradients start as None, so autograd hands
        each parameter a fresh tensor instead of launching one `+=` kernel per parameter into a zeroed bucket (673 adds
        and as many slices of zero-fill per step on Tramba-V); a completed bucket is filled by ONE multi-tensor copy."""
        self._handles = []
        for bi, bucket in enumerate(self.buckets):
            self._pending[bi] = len(bucket)
            for p in bucket:
                p.grad = None

    def _flush(self, bi):
        """Every gradient of bucket `bi` exists (or its parameter was unused): move them into the flat buffer, point
        .grad at the slices, start the all-reduce."""
        bucket = self.buckets[bi]
        views = [self._where[p][1] for p in bucket]
        dst = [v for v, p in zip(views, bucket) if p.grad is not None and p.grad is not v]
        src = [p.grad for v, p in zip(views, bucket) if p.grad is not None and p.grad is not v]
        unused = [v for v, p in zip(views, bucket) if p.grad is None]
        if dst:
            torch._foreach_copy_(dst, src)
        if unused:
            torch._foreach_zero_(unused)
        for p, v in zip(bucket, views):
            p.grad = v
        if self.world > 1:
            self._handles.append(dist.all_reduce(self.flat[bi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _on_grad(self, p):
        bi, _ = self._where[p]
        self._pending[bi] -= 1
        if self._pending[bi] == 0:
            self._flush(bi)

    def finish(self):
        """Flush buckets whose hooks did not all fire (parameters unused in this step), wait for the outstanding
        all-reduces, then average."""
        for bi, left in enumerate(self._pending):
            if left > 0:
                self._flush(bi)
                self._pending[bi] = 0
        if self.world > 1:
            for h in self._handles:
                h.wait()
            for flat in self.flat:
                flat.div_(self.world)
        self._handles = []

    def bytes_per_step(self):
        return sum(f.numel() * f.element_size() for f in self.flat)
