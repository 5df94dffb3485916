"""Data parallelism for the Tramba training step: gradients only, one process per GPU.

The reference has no distributed code (`--parallel` only prints, run.py:46-50).  The path shards by
image, so the single exchange step is an all-reduce of the gradients, averaged over the ranks.  Design
for xGMI (point-to-point links, ring collectives are per-link bound): few large flat buckets (default
32 MB) instead of one call per tensor, launched from autograd hooks in reverse registration order so RCCL
works on bucket i while backward still produces bucket i+1; one flat buffer per bucket, filled by one
multi-tensor copy when the bucket's last gradient arrives, after which the .grad fields are VIEWS into it.
The 1/world average rides inside the collective (RCCL's ncclAvg pre-multiplied sum; gloo, which has no AVG,
gets the scale folded into the bucket fill), so there is no extra pass over the 446 MB of gradients.
`bucket_dtype=torch.bfloat16` halves the bytes on the links (222.9 MB per step for Tramba-V): gradients are
rounded once into the bucket, reduced in bf16 and widened back into the fp32 .grad tensors.
Backend: "nccl" (= RCCL) on GPUs, "gloo" in the CPU tests.

Which parameters take part is decided per step from `requires_grad` (freeze_encoder / unfreeze_encoder rebuild
the buckets -- on every rank alike, these are model-level calls).  A parameter that produced no gradient in a
step keeps `.grad = None` when that is known to hold on every rank (world size 1, or `find_unused=True`, which
costs one host read of a small flag vector per step, like DDP's find_unused_parameters); otherwise it
contributes zeros to the average and receives the other ranks' mean.
"""
import torch
import torch.distributed as dist


def broadcast_parameters(model, src=0, process_group=None):
    """Seed-synchronised replicas: rank `src`'s parameters and buffers overwrite everyone's."""
    for t in list(model.parameters()) + list(model.buffers()):
        dist.broadcast(t.data, src=src, group=process_group)


class GradBucketReducer:
    def __init__(self, model, bucket_mb=32.0, process_group=None, bucket_dtype=None, find_unused=False):
        self.model = model
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.bucket_mb = float(bucket_mb)
        self.bucket_dtype = bucket_dtype
        self.find_unused = bool(find_unused)
        self.enabled = True
        self._armed = False
        self._hooks = []
        self._sig = None
        self._params = None
        backend = dist.get_backend(process_group) if dist.is_initialized() else None
        self._native_avg = backend == "nccl"          # RCCL: ncclAvg; gloo has SUM only
        # no process group at all and full-precision buckets: nothing to exchange, nothing to round -- the gradients stay
        # where autograd put them (no bucket fill: 446 MB read + written per step on Tramba-V)
        self._passthrough = (not dist.is_initialized()) and bucket_dtype is None
        self._build()

    # ------------------------------------------------------------------ bucket layout
    def _signature(self):
        # (the parameter OBJECTS of a model do not change; walking the module tree for them every step cost 4 ms of host time)
        if self._params is None:
            self._params = list(self.model.parameters())
        return tuple(p.requires_grad for p in self._params)

    def _build(self):
        self.remove_hooks()
        self._sig = self._signature()
        # backward produces gradients roughly in reverse parameter order: bucket in that order
        self._params = list(self.model.parameters())
        params = [p for p in self._params if p.requires_grad][::-1]
        cap = int(self.bucket_mb * 1024 * 1024)
        self.buckets, cur, cur_bytes = [], [], 0
        for p in params:
            nbytes = p.numel() * (p.element_size() if self.bucket_dtype is None else
                                  torch.empty((), dtype=self.bucket_dtype).element_size())
            if cur and (cur_bytes + nbytes > cap or p.dtype != cur[0].dtype or p.device != cur[0].device):
                self.buckets.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            self.buckets.append(cur)
        self.flat, self._where, self._views, self._flags = [], {}, [], []
        for bi, bucket in enumerate(self.buckets):
            dtype = bucket[0].dtype if self.bucket_dtype is None else self.bucket_dtype
            # every view starts on a 16-byte boundary (<= 12 bytes of zero padding per parameter): the `.grad` views are what
            # the optimizer reads, and tramba_adam_step takes its 16-byte path only on aligned tensors
            al = max(1, 16 // torch.empty((), dtype=dtype).element_size())
            offs, n = [], 0
            for p in bucket:
                offs.append(n)
                n += (p.numel() + al - 1) // al * al
            # the bucket's tail carries one "this rank produced a gradient" flag per parameter through the same collective
            flat = torch.zeros(n + len(bucket), dtype=dtype, device=bucket[0].device)
            views = []
            for p, off in zip(bucket, offs):
                self._where[p] = bi
                views.append(flat[off:off + p.numel()].view_as(p))
            self.flat.append(flat)
            self._views.append(views)
            self._flags.append(flat[n:])
        # fp32 landing tensors for the widened gradients of a low-precision bucket (.grad must match the parameter dtype)
        self._wide = None
        if self.bucket_dtype is not None:
            self._wide = [[torch.zeros_like(p) for p in bucket] for bucket in self.buckets]
        self._pending = [0] * len(self.buckets)
        self._streams = [set() for _ in self.buckets]
        self._handles = []
        self._have_local = {}
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self._where]
        self._armed = False

    def remove_hooks(self):
        """Detach from the model (the hooks otherwise live as long as the parameters)."""
        for h in self._hooks:
            h.remove()
        self._hooks = []

    remove = remove_hooks

    # ------------------------------------------------------------------ one step
    def prepare(self):
        """Call before each backward (it replaces optimizer.zero_grad()).  Gradients start as None, so autograd hands
        each parameter a fresh tensor instead of launching one `+=` kernel per parameter into a zeroed bucket (673 adds
        and as many slices of zero-fill per step on Tramba-V); a completed bucket is filled by ONE multi-tensor copy."""
        if self._signature() != self._sig:      # freeze_encoder() / unfreeze_encoder() since the last step
            self._build()
        self._handles = []
        self._have_local = {}
        self._streams = [set() for _ in self.buckets]     # the streams the gradients of a bucket were finished on (see _flush)
        for bi, bucket in enumerate(self.buckets):
            self._pending[bi] = len(bucket)
            for p in bucket:
                p.grad = None
        for p in self._params:
            if not p.requires_grad:
                p.grad = None
        self._armed = self.enabled

    def _flush(self, bi):
        """Every gradient of bucket `bi` exists (or its parameter was unused): move them into the flat buffer, start the
        all-reduce.  The .grad fields are pointed at the result in finish()."""
        bucket, views = self.buckets[bi], self._views[bi]
        have = [p.grad is not None for p in bucket]
        if self._passthrough:
            self._have_local[bi] = have
            return
        if bucket[0].is_cuda:
            # The model runs the decoder's guide branches on a side stream, under autograd too (models.OVERLAP_TRAINING): the
            # gradients of one bucket are then finished on DIFFERENT streams, and the hook that completes the bucket runs on only
            # one of them.  The autograd engine drives a device from one thread, so everything the other streams need has been
            # enqueued by now: an event at the tail of each of them, awaited here, orders the fill (and the collective, which
            # ProcessGroupNCCL orders behind the current stream) after every gradient of the bucket.
            cur = torch.cuda.current_stream(bucket[0].device)
            for st in self._streams[bi]:
                if st != cur:
                    ev = torch.cuda.Event()
                    ev.record(st)
                    cur.wait_event(ev)
            # gradients whose partial sums are still pending (hip.deferred_sums() around backward) must hold their values
            # before they are copied into the bucket
            from . import hip
            hip.flush_sums()
        dst = [v for v, p, h in zip(views, bucket, have) if h and p.grad is not v]
        src = [p.grad for v, p, h in zip(views, bucket, have) if h and p.grad is not v]
        unused = [v for v, h in zip(views, have) if not h]
        if dst:
            if self.world > 1 and not self._native_avg:     # gloo: SUM only -- fold the average into the fill
                src = torch._foreach_mul(src, 1.0 / self.world)
            torch._foreach_copy_(dst, src)
        if unused:
            torch._foreach_zero_(unused)
        if self.find_unused and self.world > 1:
            self._flags[bi].copy_(torch.tensor([1.0 if h else 0.0 for h in have], dtype=self._flags[bi].dtype))
        self._have_local[bi] = have
        if self.world > 1:
            op = dist.ReduceOp.AVG if self._native_avg else dist.ReduceOp.SUM
            self._handles.append(dist.all_reduce(self.flat[bi], op=op, group=self.group, async_op=True))

    def _on_grad(self, p):
        if not self._armed:
            return
        bi = self._where.get(p)
        if bi is None:
            return
        if p.is_cuda:
            self._streams[bi].add(torch.cuda.current_stream(p.device))
        self._pending[bi] -= 1
        if self._pending[bi] == 0:
            self._flush(bi)

    def finish(self):
        """Flush buckets whose hooks did not all fire (parameters unused in this step), wait for the outstanding
        all-reduces, hand every parameter its averaged gradient."""
        if not self._armed:
            return
        for bi, left in enumerate(self._pending):
            if left > 0:
                self._flush(bi)
                self._pending[bi] = 0
        for h in self._handles:
            h.wait()
        self._handles = []
        if self._passthrough:
            self._have_local = {}
            self._armed = False
            return
        for bi, bucket in enumerate(self.buckets):
            have = self._have_local.get(bi, [True] * len(bucket))
            if self.world > 1:
                if self.find_unused:
                    got = self._flags[bi].float().tolist()            # one small host read per bucket (DDP does the same)
                    have = [g > 0.0 for g in got]
                else:
                    have = [True] * len(bucket)                       # unknown for the other ranks: everyone takes the mean
            views = self._views[bi]
            if self._wide is not None:
                wide = self._wide[bi]
                dst = [w for w, h in zip(wide, have) if h]
                if dst:       # (a bucket none of whose parameters produced a gradient: world 1 or find_unused)
                    torch._foreach_copy_(dst, [v for v, h in zip(views, have) if h])
                views = wide
            for p, v, h in zip(bucket, views, have):
                p.grad = v if h else None
        self._have_local = {}
        self._armed = False

    def bytes_per_step(self):
        """Bytes this rank hands to the collective per step (gradients; the per-parameter flags are ~3 KB)."""
        return sum(sum(p.numel() for p in b) * f.element_size() for f, b in zip(self.flat, self.buckets))
