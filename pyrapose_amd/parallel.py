"""Data parallelism over RCCL/xGMI: one process per GPU, per-image batch sharding.

The reference has no collective path (bin/train.py:82-89 is a disabled multi_gpu_model branch); the
single-device semantics it fixes are: the loss normalisers count positives over the WHOLE batch
(losses.py:62-66, :402-405) and Adam clips by the GLOBAL gradient norm (bin/train.py:101).  Exact DP
equivalence therefore needs two exchanges per step and nothing else:
  1. a 3-integer SUM all-reduce of the positive counts before the loss backward, so every rank
     normalises by the global count and its gradient is its exact share of the global-batch gradient;
  2. a SUM all-reduce of the flat gradient buffer, cut into buckets in backward-completion order
     (heads -> FPN -> res5 -> res4 -> res3) and launched on a side stream as soon as the last
     weight-gradient kernel of a bucket has been enqueued, overlapping the remaining backward.
The global-norm clip then runs redundantly on every rank on the reduced buffer (fixed-order reduction,
so all ranks compute the same factor).  Frozen tensors (conv1, res2*) are never communicated.
"""
import torch
import torch.distributed as dist


def plan_buckets(entries, bwd_ops, bucket_bytes):
    """entries: ParamStore.entries (layout order); bwd_ops: Engine.bwd_ops.  Returns a list of
    (start, end, ready_op_index) in launch order.  Pure function -- unit-tested on CPU."""
    trainable = [(e["offset"], e["offset"] + e["count"]) for e in entries.values() if e["trainable"]]
    if not trainable:
        return []
    lo = min(a for a, _ in trainable)
    hi = max(b for _, b in trainable)
    # walk the layout backwards (== backward completion order) and cut every >= bucket_bytes
    cuts, acc, end = [], 0, hi
    for (a, b) in sorted(trainable, reverse=True):
        acc += (b - a) * 4
        if acc >= bucket_bytes:
            cuts.append((a, end))
            end, acc = a, 0
    if end > lo:
        cuts.append((lo, end))
    ready = [-1] * len(cuts)
    for i, op in enumerate(bwd_ops):
        wr = getattr(op, "wrange", None)
        if wr is None:
            continue
        for bi, (a, b) in enumerate(cuts):
            if a <= wr[0] < b:
                ready[bi] = max(ready[bi], i)
                break
    out = [(a, b, r) for (a, b), r in zip(cuts, ready)]
    out.sort(key=lambda t: t[2])
    return out


class DataParallel(object):
    def __init__(self, engine, group=None, bucket_bytes=32 << 20):
        self.eng, self.group = engine, group
        self.active = dist.is_initialized()
        self.world = dist.get_world_size(group) if self.active else 1
        self.buckets = plan_buckets(engine.params.entries, engine.bwd_ops, bucket_bytes)
        self.by_op = {}
        for (a, b, r) in self.buckets:
            self.by_op.setdefault(r, []).append((a, b))
        self.flat = engine.params.grad
        self.on_gpu = self.flat.is_cuda
        self.comm_stream = torch.cuda.Stream() if self.on_gpu else None
        self.works = []
        engine.grad_sync = self

    def reduce_counts(self, counts):
        if self.active:
            dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=self.group)

    def _launch(self, a, b, stream=None):
        t = self.flat[a:b]
        if not self.active:
            return
        if self.on_gpu:
            ev = torch.cuda.Event()
            ev.record(stream if stream is not None else torch.cuda.current_stream())
            self.comm_stream.wait_event(ev)
            with torch.cuda.stream(self.comm_stream):
                self.works.append(dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    def after_bwd_op(self, i, stream=None):
        """`stream` = the stream op i was enqueued on (weight gradients run on a side lane)."""
        for (a, b) in self.by_op.get(i, ()):
            self._launch(a, b, stream)

    def finish(self):
        for (a, b) in self.by_op.get(-1, ()):  # buckets no weight-gradient op maps to (defensive)
            self._launch(a, b)
        for w in self.works:
            w.wait()  # orders the compute stream after the collective; no host sync
        self.works = []
