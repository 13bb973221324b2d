"""CPU, world_size 2 over gloo: the data-parallel exchange plan (bucketed SUM all-reduce of the flat gradient
buffer in backward-completion order + global positive counts) reproduces the single-process result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class FakeOp(object):
    def __init__(self, wrange):
        self.wrange = wrange


def make_layout(sizes, frozen):
    entries, off = {}, 0
    for i, n in enumerate(sizes):
        entries["t%d" % i] = dict(offset=off, count=n, trainable=(i not in frozen))
        off += (n + 63) // 64 * 64
    return entries, off


def test_plan_buckets_covers_trainable_range_once():
    from pyrapose_amd.parallel import plan_buckets
    sizes = [1000, 64, 5000, 128, 70000, 256, 300000, 64, 9000]
    entries, total = make_layout(sizes, frozen={0, 1})
    names = list(entries)
    # backward visits tensors in reverse layout order
    ops = [FakeOp(None)]
    for n in reversed(names[2:]):
        e = entries[n]
        ops.append(FakeOp((e["offset"], e["offset"] + e["count"])))
        ops.append(FakeOp(None))
    buckets = plan_buckets(entries, ops, bucket_bytes=256 * 1024)
    assert len(buckets) >= 2
    lo = entries["t2"]["offset"]; hi = entries["t8"]["offset"] + entries["t8"]["count"]
    spans = sorted((a, b) for a, b, _ in buckets)
    assert spans[0][0] == lo and spans[-1][1] == hi
    for (a0, b0), (a1, b1) in zip(spans, spans[1:]):
        assert b0 == a1                       # contiguous, no overlap, no gap
    ready = [r for _, _, r in buckets]
    assert ready == sorted(ready) and all(r >= 0 for r in ready)
    # a bucket is launched only after the last weight-gradient op that writes into it
    for a, b, r in buckets:
        for i, op in enumerate(ops):
            if op.wrange and a <= op.wrange[0] < b:
                assert i <= r
    # frozen tensors are never communicated
    assert spans[0][0] >= entries["t1"]["offset"] + entries["t1"]["count"]


def test_bucket_size_from_env_and_plan_description(monkeypatch):
    from pyrapose_amd import parallel
    monkeypatch.delenv("PP_BUCKET_MB", raising=False)
    assert parallel.bucket_bytes_from_env() == 32 << 20
    monkeypatch.setenv("PP_BUCKET_MB", "8")
    assert parallel.bucket_bytes_from_env() == 8 << 20
    monkeypatch.setenv("PP_BUCKET_MB", "0")
    with pytest.raises(ValueError):
        parallel.bucket_bytes_from_env()

    class Op(object):
        def __init__(self, name, wrange=None):
            self.name, self.wrange = name, wrange
    text = parallel.describe_buckets([(100, 300, 1), (0, 100, 2)], [Op("a"), Op("wgrad:reg_out", (100, 300)), Op("wgrad:P3", (0, 100))])
    lines = text.splitlines()
    assert lines[0].startswith("gradient all-reduce plan: 2 buckets")
    assert "floats [100, 300)" in lines[1] and "launch 1 (wgrad:reg_out)" in lines[1]
    assert "floats [0, 100)" in lines[2] and "launch 2 (wgrad:P3)" in lines[2]


def test_bucket_cut_between_kernel_and_bias_waits_for_the_layers_launch():
    """A weight-gradient launch writes its layer's kernel AND bias slot.  With a bucket size that puts the cut between the two,
    BOTH buckets must wait for that launch (the bias sits in the earlier bucket of the layout walk)."""
    from pyrapose_amd.parallel import plan_buckets
    # layers: (kernel, bias) pairs, 64-float aligned
    sizes = [4096, 64, 4096, 64, 4096, 64]
    entries, total = make_layout(sizes, frozen=set())
    names = list(entries)
    ops = []
    for li in (2, 1, 0):  # backward: last layer first; one launch per layer covering [kernel offset, bias end)
        k, b = entries[names[2 * li]], entries[names[2 * li + 1]]
        ops.append(FakeOp((k["offset"], b["offset"] + b["count"])))
    # walking the layout backwards, 4096 + 64 floats per layer: a bucket of exactly one kernel (16 KB) cuts at every kernel,
    # a bucket of 64 floats cuts at every entry -> bias and kernel of a layer land in different buckets
    buckets = plan_buckets(entries, ops, bucket_bytes=64 * 4)
    spans = sorted((a, b) for a, b, _ in buckets)
    assert any(a == entries[names[1]]["offset"] for a, _ in spans)  # a cut does sit at a bias entry
    for a, b, r in buckets:
        writers = [i for i, op in enumerate(ops) if op.wrange[0] < b and a < op.wrange[1]]
        assert writers and r == max(writers), (a, b, r, writers)


def _worker(rank, world, port, sizes, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pyrapose_amd.parallel import DataParallel
    entries, total = make_layout(sizes, frozen={0})
    names = list(entries)

    class Params(object):
        pass

    class Eng(object):
        pass

    eng = Eng()
    eng.params = Params()
    eng.params.entries = entries
    eng.params.grad = torch.zeros(total, dtype=torch.float32)
    eng.bwd_ops = []
    for n in reversed(names[1:]):
        e = entries[n]
        eng.bwd_ops.append(FakeOp((e["offset"], e["offset"] + e["count"])))
        eng.bwd_ops.append(FakeOp(None))
    dp = DataParallel(eng, bucket_bytes=64 * 1024)
    # "backward": rank r writes its share of the global-batch gradient
    rng = np.random.default_rng(100 + rank)
    counts = torch.tensor([3 + rank, 5 * (rank + 1), rank, 0], dtype=torch.int32)
    dp.reduce_counts(counts)
    for i, op in enumerate(eng.bwd_ops):
        if op.wrange:
            a, b = op.wrange
            eng.params.grad[a:b] = torch.from_numpy(rng.standard_normal(b - a).astype(np.float32))
        dp.after_bwd_op(i)
    dp.finish()
    q.put((rank, counts.numpy().copy(), eng.params.grad.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_allreduce_matches_single_process():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    sizes = [500, 3000, 64, 20000, 128, 40000]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, sizes, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        r, c, g = q.get(timeout=120)
        res[r] = (c, g)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # expected: element-wise sum of what each rank produced
    entries, total = make_layout(sizes, frozen={0})
    want = np.zeros(total, np.float32)
    for rank in range(2):
        rng = np.random.default_rng(100 + rank)
        part = np.zeros(total, np.float32)
        for n in reversed(list(entries)[1:]):
            e = entries[n]
            part[e["offset"]: e["offset"] + e["count"]] = rng.standard_normal(e["count"]).astype(np.float32)
        want += part
    for rank in range(2):
        c, g = res[rank]
        assert np.array_equal(c, np.array([7, 15, 1, 0], np.int32))
        np.testing.assert_allclose(g, want, rtol=1e-6, atol=1e-6)
    assert np.array_equal(res[0][1], res[1][1])   # both ranks hold the identical reduced buffer


def _sync_worker(rank, world, port, fail, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["PP_EPOCH_END_TIMEOUT_S"] = "120"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pyrapose_amd import parallel
    box = {"lr": 1e-5, "stop": False, "ran": 0}

    def rank0():
        box["ran"] += 1
        if fail:
            raise ValueError("snapshot disk full")
        box["lr"], box["stop"] = 2.5e-6, True

    try:
        parallel.epoch_end_sync(rank0, lambda: [box["lr"], box["stop"]], lambda s: box.update(lr=s[0], stop=s[1]))
        q.put((rank, "ok", box["lr"], box["stop"], box["ran"]))
    except Exception as e:  # noqa: BLE001
        q.put((rank, type(e).__name__, str(e)[-200:], None, box["ran"]))
        dist.destroy_process_group()
        raise SystemExit(3)
    dist.destroy_process_group()


@pytest.mark.parametrize("fail", [False, True])
def test_epoch_end_sync_state_and_failure_reach_every_rank(fail):
    """rank 0 alone runs the epoch-end callbacks; its learning rate / stop flag reach the other rank over the gloo
    control group, and when a rank-0 callback raises EVERY rank raises and exits non-zero (no rank is left in a barrier)."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sync_worker, args=(r, 2, port, fail, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        r = q.get(timeout=120)
        res[r[0]] = r[1:]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == (3 if fail else 0)
    assert res[0][3] == 1 and res[1][3] == 0          # the section ran on rank 0 only
    if fail:
        assert res[0][0] == "ValueError" and res[1][0] == "RankZeroFailed" and "snapshot disk full" in res[1][1]
    else:
        assert res[0][:3] == ("ok", 2.5e-6, True) and res[1][:3] == ("ok", 2.5e-6, True)


@pytest.mark.parametrize("world", [4, 8])
def test_bucket_plan_and_batch_dealing_at_world_4_and_8(world):
    """the bucket plan does not depend on the world size and covers every trainable float exactly once with
    monotone release indices; fit_generator's dealing (rank r takes batches r, r + world, ...) visits every batch of an epoch once."""
    from pyrapose_amd.parallel import plan_buckets
    rng = np.random.default_rng(world)
    sizes = [int(v) for v in rng.integers(64, 400000, size=40)]
    frozen = set(range(6))
    entries, total = make_layout(sizes, frozen)
    names = list(entries)
    ops = []
    for n in reversed(names[6:]):
        e = entries[n]
        ops += [FakeOp(None), FakeOp((e["offset"], e["offset"] + e["count"]))]
    for mb in (1, 4, 32):
        buckets = plan_buckets(entries, ops, bucket_bytes=mb << 20)
        cover = np.zeros(total, np.int32)
        for a, b, _ in buckets:
            cover[a:b] += 1
        for i, n in enumerate(names):
            e = entries[n]
            want = 0 if i in frozen else 1
            assert (cover[e["offset"]: e["offset"] + e["count"]] == want).all(), (n, mb)
        rel = [r for _, _, r in buckets]
        assert rel == sorted(rel) and rel[0] >= 0
    # dealing: `steps` global steps of `world` batches; the generator has n_batches batches
    n_batches = 8 * world + 3
    steps = -(-n_batches // world)
    seen = []
    for rank in range(world):
        seen += [(i * world + rank) % n_batches for i in range(steps)]
    assert set(seen) == set(range(n_batches))                      # every batch of the epoch is visited
    assert len(seen) - n_batches < world                           # at most one partial global step wraps around


def _path_worker(rank, world, port, case, q):
    """DataParallel's choice between the library-owned RCCL communicator and torch.distributed, with a stand-in for NativeComm
    (no GPU here): the ranks must end up on the SAME path whatever one of them finds."""
    import warnings
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["PP_DP_NATIVE"] = "1"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pyrapose_amd import parallel

    class FakeComm(object):
        made, closed = 0, 0

        def __init__(self, dev, world_, rank_, uid):
            assert uid == b"u" * 128 and world_ == world and rank_ == rank
            if case == "init_fails_on_rank_1" and rank_ == 1:
                raise RuntimeError("ncclCommInitRank: unhandled system error")
            FakeComm.made += 1
            self.world, self.rank, self.stream = world_, rank_, None

        @staticmethod
        def available():
            return not (case == "unavailable_on_rank_1" and rank == 1)

        @staticmethod
        def unique_id(dev=0):
            return b"u" * 128

        def allreduce(self, t, after=None):
            dist.all_reduce(t)

        def allreduce_counts(self, c):
            dist.all_reduce(c)

        def close(self):
            FakeComm.closed += 1

    parallel.NativeComm = FakeComm
    parallel._NATIVE_COMMS.clear()

    class Grad(object):  # a "device" gradient buffer that lives on the CPU
        is_cuda = True
        device = torch.device("cpu")

        def __init__(self, n):
            self.t = torch.zeros(n, dtype=torch.float32)

        def __getitem__(self, s):
            return self.t[s]

    class Stream(object):
        def wait_stream(self, other):
            pass

    entries, total = make_layout([100, 2000, 3000], frozen={0})

    def engine():
        class Obj(object):
            pass
        eng = Obj()
        eng.params = Obj()
        eng.params.entries, eng.params.grad = entries, Grad(total)
        eng.bwd_ops = [FakeOp((e["offset"], e["offset"] + e["count"])) for e in list(entries.values())[:0:-1]]
        eng.streams = [Stream()]
        return eng

    with warnings.catch_warnings(record=True) as wlist:
        warnings.simplefilter("always")
        dps = [parallel.DataParallel(engine(), bucket_bytes=4096), parallel.DataParallel(engine(), bucket_bytes=4096)]
    for dp in dps:  # whichever path: the sum over the ranks
        if dp.native is None:
            dp.on_gpu = False  # (the stand-in buffer really lives on the CPU: torch.distributed's path must not look for a HIP stream)
        dp.flat.t[:] = float(rank + 1)
        c = torch.tensor([rank + 1, 0, 0, 0], dtype=torch.int32)
        dp.reduce_counts(c)
        for i in range(len(dp.eng.bwd_ops)):
            dp.after_bwd_op(i)
        dp.finish()
        assert int(c[0]) == 3
        for n, e in list(entries.items())[1:]:
            assert bool((dp.flat.t[e["offset"]: e["offset"] + e["count"]] == 3.0).all()), n
    q.put((rank, [type(dp.native).__name__ for dp in dps], dps[0].native is dps[1].native, FakeComm.made, FakeComm.closed,
           [str(w.message)[:60] for w in wlist]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case", ["available", "unavailable_on_rank_1", "init_fails_on_rank_1"])
def test_ranks_agree_on_the_allreduce_path(case):
    """Every rank takes the library-owned communicator, or every rank takes torch.distributed: a rank that cannot load RCCL, or whose
    ncclCommInitRank fails, pulls the others onto the fallback with it (no rank is left inside a collective of the other path); the
    communicator is created once per process and shared by the engines built after the first."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_path_worker, args=(r, 2, port, case, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        r = q.get(timeout=120)
        res[r[0]] = r[1:]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in range(2):
        kinds, same, made, closed, warns = res[rank]
        if case == "available":
            assert kinds == ["FakeComm", "FakeComm"] and same and made == 1 and closed == 0 and not warns
        elif case == "unavailable_on_rank_1":
            assert kinds == ["NoneType", "NoneType"] and made == 0 and not warns
        else:
            assert kinds == ["NoneType", "NoneType"] and made == (1 if rank == 0 else 0) and closed == made  # (one attempt, not one per engine)
            assert len(warns) == 1 and "RCCL communicator" in warns[0]
