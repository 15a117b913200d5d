"""Data-parallel plumbing: one process per GPU, RCCL over xGMI through torch.distributed.

The reference's intended semantics: every rank runs the model on its own
batch, gradients are MEAN-all-reduced, codebook statistics are SUM-all-reduced
(models/vqvae/bottleneck.py, in this build).  ``GradSync`` keeps all gradients in one flat
fp32 buffer, cut into a few large buckets in reverse parameter order; a bucket's all-reduce
is launched asynchronously from the autograd hook of its last-arriving gradient, so the
collectives overlap the rest of backward.  xGMI is point-to-point (7 links per GPU): a few
multi-megabyte buckets keep every link busy without per-tensor launch latency.
"""
import torch
import torch.distributed as dist


def broadcast_module(module, src=0):
    """Make parameters and buffers identical on every rank (DDP's constructor does this)."""
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t, src)          # in place on the parameter itself: bumps its version (pack cache)


class GradSync:
    def __init__(self, params, bucket_bytes=8 << 20, group=None, always_reduce=False, timing=False):
        """``always_reduce``: register the hooks and issue the collectives even in a world of one rank (exercises the
        asynchronous RCCL path on a single GPU).  ``timing``: bracket finish() with events on the current stream;
        ``exposed_ms()`` is then the part of the gradient exchange that backward did not hide."""
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.reduce = dist.is_initialized() and (self.world > 1 or always_reduce)
        self.timing, self._events = timing, []
        total = sum(p.numel() for p in self.params)
        device = self.params[0].device
        self.flat = torch.zeros(total, dtype=torch.float32, device=device)
        self._views, self._bucket_of, self.buckets = [], {}, []
        # parameters in REVERSE registration order: the decoder's gradients arrive first
        offset, start, pending = 0, 0, []
        for p in reversed(self.params):
            n = p.numel()
            self._views.append((p, offset, n))
            pending.append(p)
            offset += n
            if (offset - start) * 4 >= bucket_bytes:
                self._close_bucket(start, offset, pending)
                start, pending = offset, []
        if pending:
            self._close_bucket(start, offset, pending)
        self._handles = []
        self._attach_views()
        if self.reduce:
            for p in self.params:
                p.register_post_accumulate_grad_hook(self._on_grad)

    def _close_bucket(self, start, end, members):
        index = len(self.buckets)
        self.buckets.append({"start": start, "end": end, "count": len(members), "ready": 0})
        for p in members:
            self._bucket_of[p] = index

    def _attach_views(self):
        for p, offset, n in self._views:
            p.grad = self.flat[offset:offset + n].view_as(p)

    def zero_grad(self):
        """Replaces optimizer.zero_grad(): zero the flat buffer and keep .grad pointing into it."""
        self.flat.zero_()
        for b in self.buckets:
            b["ready"] = 0
        self._handles = []
        self._attach_views()

    def _on_grad(self, param):
        b = self.buckets[self._bucket_of[param]]
        b["ready"] += 1
        if b["ready"] == b["count"]:
            view = self.flat[b["start"]:b["end"]]
            self._handles.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """Wait for the in-flight buckets and turn the sums into means."""
        if not self.reduce:
            return
        launched = len(self._handles)
        ev = None
        if self.timing and self.flat.is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        for h in self._handles:
            h.wait()
        self._handles = []
        if launched != len(self.buckets):
            # some parameter received no gradient this step: reduce the stragglers' buckets now
            for b in self.buckets:
                if b["ready"] != b["count"]:
                    dist.all_reduce(self.flat[b["start"]:b["end"]], op=dist.ReduceOp.SUM, group=self.group)
        if self.world > 1:
            self.flat.mul_(1.0 / self.world)
        if ev is not None:
            ev[1].record()
            self._events.append(ev)

    def exposed_ms(self, reset=True):
        """Mean time per step the stream spent in finish() (waiting for bucket all-reduces + the 1/world scale)."""
        if not self._events:
            return None
        torch.cuda.synchronize()
        ms = sum(a.elapsed_time(b) for a, b in self._events) / len(self._events)
        if reset:
            self._events = []
        return ms
