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
import os

import torch
import torch.distributed as dist


def ensure_process_group(device=None):
    """WORLD_SIZE > 1 (the process was started by torch.distributed.run or bench.py's own launcher) and no process group yet:
    create the RCCL group (backend 'nccl'; PP_DIST_BACKEND=gloo for CPU / shared-card rehearsals).  Returns the world size."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        # the rank's GPU becomes the CURRENT device before anything allocates: without this the first x.cuda() of
        # train_on_batch / the DevicePrefetcher of fit_generator would stage every rank's batches on cuda:0
        # (PP_DIST_BACKEND=gloo rehearsals with several ranks on one card keep device 0)
        local = int(device if device is not None else os.environ.get("LOCAL_RANK", "0"))
        if torch.cuda.is_available() and local < torch.cuda.device_count():
            torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("PP_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dev = torch.device("cuda", int(device if device is not None else os.environ.get("LOCAL_RANK", "0")))
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    return world


_CONTROL_GROUP = None


def control_group():
    """A gloo side group for host-side, long-wait synchronisation (the epoch end of fit_generator: rank 0 snapshots and evaluates,
    possibly for longer than the RCCL watchdog's 10 minutes, while the other ranks wait).  Created on first use -- collectively,
    every rank must call this at the same point -- with a timeout of PP_EPOCH_END_TIMEOUT_S seconds (default 24 h)."""
    global _CONTROL_GROUP
    if not dist.is_initialized():
        return None
    if _CONTROL_GROUP is None:
        import datetime
        secs = float(os.environ.get("PP_EPOCH_END_TIMEOUT_S", "86400"))
        _CONTROL_GROUP = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=secs))
    return _CONTROL_GROUP


class RankZeroFailed(RuntimeError):
    """raised on every rank when the rank-0 section of epoch_end_sync raised (the original traceback text is the message)"""


def epoch_end_sync(rank0_fn, get_state, set_state):
    """Run `rank0_fn()` on rank 0 only, then hand `get_state()` of rank 0 to `set_state(state)` on the other ranks.  The wait is a
    broadcast on the gloo control group (no RCCL collective is pending meanwhile, so no watchdog can fire however long rank 0
    works), and an exception in rank0_fn travels with the state: EVERY rank raises (rank 0 the original exception, the others
    RankZeroFailed with its traceback), so the job exits non-zero together instead of N - 1 ranks hanging in a barrier."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        rank0_fn()
        return
    grp = control_group()
    rank = dist.get_rank()
    err, exc, state = None, None, None
    if rank == 0:
        try:
            rank0_fn()
            state = get_state()
        except BaseException as e:  # noqa: BLE001 -- the failure is re-raised below, after the others have been told
            import traceback
            exc, err = e, traceback.format_exc()
    box = [state, err]
    dist.broadcast_object_list(box, src=0, group=grp)
    if box[1] is not None:
        if rank == 0:
            raise exc
        raise RankZeroFailed("rank 0 failed in its epoch-end section:\n%s" % box[1])
    if rank != 0:
        set_state(box[0])


def bucket_bytes_from_env(default=32 << 20):
    """PP_BUCKET_MB: size at which the flat gradient buffer is cut into all-reduce buckets (default 32 MB: ~6 buckets for the
    169 MB of ResNet-50 gradients; xGMI is point to point, so fewer, larger collectives beat many small ones)."""
    v = os.environ.get("PP_BUCKET_MB")
    if not v:
        return default
    mb = float(v)
    if not mb > 0:
        raise ValueError("PP_BUCKET_MB must be positive, got %r" % v)
    return int(mb * (1 << 20))


def describe_buckets(buckets, bwd_ops=None):
    """One line per bucket (launch order): float offsets, bytes, index (and name) of the backward launch that releases it."""
    lines = ["gradient all-reduce plan: %d buckets, %.1f MB" % (len(buckets), sum(b - a for a, b, _ in buckets) * 4 / 2**20)]
    for i, (a, b, r) in enumerate(buckets):
        name = ""
        if bwd_ops is not None and 0 <= r < len(bwd_ops):
            name = " (%s)" % getattr(bwd_ops[r], "name", type(bwd_ops[r]).__name__)
        lines.append("  bucket %d: floats [%d, %d) = %.2f MB, released after backward launch %d%s" % (i, a, b, (b - a) * 4 / 2**20, r, name))
    return "\n".join(lines)


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
    # a bucket is complete once EVERY launch that writes into it has been enqueued: a weight-gradient launch writes the
    # layer's kernel AND bias slots (wrange = [kernel offset, bias end)), and a cut may fall between the two
    ready = [-1] * len(cuts)
    for i, op in enumerate(bwd_ops):
        wr = getattr(op, "wrange", None)
        if wr is None:
            continue
        for bi, (a, b) in enumerate(cuts):
            if wr[0] < b and a < wr[1]:
                ready[bi] = max(ready[bi], i)
    out = [(a, b, r) for (a, b), r in zip(cuts, ready)]
    out.sort(key=lambda t: t[2])
    return out


class NativeComm(object):
    """An RCCL communicator owned by the HIP library (include/pyrapose_hip.h: pp_comm_*, pp_allreduce_bucket; librccl.so through
    dlopen) on a stream of the engine's own: the all-reduces are plain launches on that stream, ordered against the lanes with
    stream waits -- no torch.distributed in the data path.  The 128-byte unique id is the only thing that needs a side channel."""

    def __init__(self, device, world, rank, unique_id):
        import ctypes as C
        from . import ops
        from ._lib import check, lib
        self.world, self.rank = int(world), int(rank)
        self.stream = torch.cuda.Stream(device)
        self.ctx = ops.Context(device, self.stream)
        self.handle = C.c_void_p()
        self._id = C.create_string_buffer(bytes(unique_id), 128)
        check(lib.pp_comm_init(self.ctx.handle, self.world, self.rank, self._id, C.byref(self.handle)), self.ctx.handle, "pp_comm_init")

    @staticmethod
    def available():
        from ._lib import MISSING, lib
        return "pp_comm_available" not in MISSING and bool(lib.pp_comm_available())

    @staticmethod
    def unique_id(device=0):
        import ctypes as C
        from . import ops
        from ._lib import check, lib
        ctx = ops.Context(device)
        buf = C.create_string_buffer(128)
        check(lib.pp_comm_unique_id(ctx.handle, buf), ctx.handle, "pp_comm_unique_id")
        ctx.close()
        return bytes(buf.raw)

    def allreduce(self, t, after=None):
        """in-place SUM all-reduce of a float32 device tensor on the communicator's stream, ordered after stream `after`"""
        import ctypes as C
        from ._lib import check, lib
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        self.stream.wait_stream(after if after is not None else torch.cuda.current_stream())
        check(lib.pp_allreduce_bucket(self.ctx.handle, self.handle, C.c_void_p(t.data_ptr()), t.numel()), self.ctx.handle, "pp_allreduce_bucket")

    def allreduce_counts(self, counts):
        """in-place SUM all-reduce of an int32 device tensor, ordered on both sides with the CURRENT stream (the losses read it next)"""
        import ctypes as C
        from ._lib import check, lib
        assert counts.is_cuda and counts.dtype == torch.int32 and counts.is_contiguous()
        cur = torch.cuda.current_stream()
        self.stream.wait_stream(cur)
        check(lib.pp_allreduce_counts(self.ctx.handle, self.handle, C.c_void_p(counts.data_ptr()), counts.numel()), self.ctx.handle,
              "pp_allreduce_counts")
        cur.wait_stream(self.stream)

    def close(self):
        from ._lib import lib
        if self.handle:
            torch.cuda.synchronize()
            lib.pp_comm_destroy(self.handle)
            self.handle = None
        self.ctx.close()


_NATIVE_COMMS = {}  # (device, id(group), world) -> NativeComm shared by the DataParallel objects of this process


class DataParallel(object):
    def __init__(self, engine, group=None, bucket_bytes=None, native=None):
        """native: True = the gradient / count all-reduces run on the library's own RCCL communicator (NativeComm: pp_allreduce_bucket
        on a stream the engine owns); False = through torch.distributed (the host-layer fallback: gloo rehearsals, CPU tests);
        None = native when the library can load RCCL, the tensors live on a GPU and the process group (if any) is an RCCL one
        (PP_DP_NATIVE=0 / 1 overrides).  Without a process group, native=True builds a one-rank communicator."""
        self.eng, self.group = engine, group
        bucket_bytes = bucket_bytes_from_env() if bucket_bytes is None else bucket_bytes
        self.active = dist.is_initialized()
        if not self.active and int(os.environ.get("WORLD_SIZE", "1")) > 1:
            raise RuntimeError("DataParallel: WORLD_SIZE=%s but torch.distributed is not initialised -- every rank would silently train "
                               "its own copy; call parallel.ensure_process_group() first" % os.environ["WORLD_SIZE"])
        self.world = dist.get_world_size(group) if self.active else 1
        self.native = None
        on_gpu = engine.params.grad.is_cuda
        env = os.environ.get("PP_DP_NATIVE")
        explicit = native is True
        if native is None:
            native = (env != "0") and on_gpu and ((self.active and dist.get_backend(group) == "nccl") or env == "1")
            want = bool(native)
            native = want and NativeComm.available()
            if want and self.active and self.world > 1:
                # every rank takes the SAME path: a rank whose library cannot load RCCL would otherwise sit in torch.distributed's
                # all-reduce while the others wait in ncclCommInitRank
                flag = torch.tensor([1 if native else 0], dtype=torch.int32, device=engine.params.grad.device)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
                native = bool(int(flag.item()))
        dev = (engine.params.grad.device.index or 0) if on_gpu else -1
        cache_key = (dev, id(group) if group is not None else 0, self.world)
        if native and cache_key in _NATIVE_COMMS:
            # one communicator per (device, group) and process: a second engine on the same ranks (bench.py builds one per measured
            # mode, one after the other) reuses it instead of paying ncclCommInitRank again -- every rank takes this branch together
            self.native = _NATIVE_COMMS[cache_key]
            native = False
        if native:
            if not (on_gpu and NativeComm.available()):
                raise RuntimeError("DataParallel(native=True): needs device tensors and a loadable librccl.so (PP_RCCL_LIB)")
            rank = dist.get_rank(group) if self.active else 0
            box = [NativeComm.unique_id(dev) if rank == 0 else None]
            if self.active and self.world > 1:
                dist.broadcast_object_list(box, src=0, group=group)  # (the id's only journey; the data path never sees torch.distributed)
            try:
                self.native = NativeComm(dev, self.world, rank, box[0])
                err = None
            except Exception as e:  # noqa: BLE001 -- (ncclCommInitRank reports a failure on every rank of the communicator)
                self.native, err = None, e
            if self.active and self.world > 1:
                flag = torch.tensor([0 if err else 1], dtype=torch.int32, device=engine.params.grad.device)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
                if not int(flag.item()) and self.native is not None:
                    self.native.close()
                    self.native = None
            if self.active:
                _NATIVE_COMMS[cache_key] = self.native  # (None after a failure: the next engine does not try again)
            if self.native is None:
                if explicit or not self.active:
                    raise RuntimeError("DataParallel: the library-owned RCCL communicator could not be created (%s)" % (err,))
                import warnings
                warnings.warn("pyrapose_amd.parallel: library-owned RCCL communicator unavailable (%s): gradient all-reduce through "
                              "torch.distributed" % (err,))
        self.buckets = plan_buckets(engine.params.entries, engine.bwd_ops, bucket_bytes)
        if os.environ.get("PP_DP_DEBUG") == "1":
            rank = dist.get_rank(group) if self.active else 0
            print("[pyrapose_amd.parallel rank %d/%d] %s\n  all-reduce path: %s" % (rank, self.world, describe_buckets(self.buckets, engine.bwd_ops),
                  "pp_allreduce_bucket (library-owned RCCL communicator)" if self.native is not None else "torch.distributed"), flush=True)
        self.by_op = {}
        for (a, b, r) in self.buckets:
            self.by_op.setdefault(r, []).append((a, b))
        self.flat = engine.params.grad
        self.on_gpu = self.flat.is_cuda
        self.works = []
        engine.grad_sync = self

    def reduce_counts(self, counts):
        if self.native is not None:
            self.native.allreduce_counts(counts)
        elif self.active:
            dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=self.group)

    def _launch(self, a, b, stream=None):
        t = self.flat[a:b]
        if self.native is not None:
            self.native.allreduce(t, after=stream)
            return
        if not self.active:
            return
        if self.on_gpu:
            # the process group enqueues the collective on ITS OWN stream, ordered after the stream that is current at the call:
            # making the producing lane current is all the ordering there is to do (no event, no communication stream of ours --
            # one HIP stream fewer per rank: streams beyond the hardware queues of a process alias and serialise)
            with torch.cuda.stream(stream if stream is not None else torch.cuda.current_stream()):
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
        st = getattr(self.eng, "streams", None)
        if self.native is not None:
            (st[0] if st else torch.cuda.current_stream()).wait_stream(self.native.stream)  # the optimizer runs on the engine's lane 0
            return
        if self.on_gpu and st:
            with torch.cuda.stream(st[0]):  # the optimizer runs on the engine's lane 0
                for w in self.works:
                    w.wait()  # orders that stream after the collective; no host sync
        else:
            for w in self.works:
                w.wait()
        self.works = []
