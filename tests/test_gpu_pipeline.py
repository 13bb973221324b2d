"""GPU: two optional restructurings of the training step that must not change its results.
(1) Look-ahead (PP_PREFETCH=1): the frozen prefix (conv1 + res2, never trained -- bin/train.py / models/resnet.py:87-110 freeze
them through their BatchNorm layers) of batch i+1 runs on its own stream beside batch i (Engine.forward(next_x=...)): not a
single bit of the forward pass may change, whatever the caller does with the look-ahead.
(2) Sparse forward of the 3D-box head (PP_SPARSE_FWD=1): in a training step that head computes only what its loss reads."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

B, H, W, C = 2, 97, 131, 5


@pytest.fixture(scope="module")
def ctx():
    from pyrapose_amd.runtime import default_context
    return default_context()


def batches(n, seed=0):
    rng = np.random.default_rng(seed)
    mean = np.array([103.939, 116.779, 123.68], np.float32)
    return [torch.from_numpy(rng.integers(0, 256, size=(B, H, W, 3)).astype(np.float32) - mean).cuda() for _ in range(n)]


def heads(eng):
    torch.cuda.synchronize()
    return [a.t[:, : a.C].clone() for a in (eng.reg_out, eng.cls_out, eng.mask_out)]  # (columns past the channels are padding: never written)


def same(a, b):
    return all(torch.equal(x, y) for x, y in zip(a, b))


def test_prefetched_prefix_is_bit_identical(ctx, monkeypatch):
    from pyrapose_amd import arch
    from pyrapose_amd.engine import Engine
    Wt = arch.init_weights(C, seed=3)
    xs = batches(4)
    monkeypatch.delenv("PP_PREFETCH", raising=False)
    ref = Engine(ctx, C, B, H, W, weights=Wt, train=True)
    assert ref.prefix_lane is None
    want = []
    for x in xs:
        ref.forward(x)
        want.append(heads(ref))
    monkeypatch.setenv("PP_PREFETCH", "1")
    eng = Engine(ctx, C, B, H, W, weights=Wt, train=True)
    assert eng.prefix_lane is not None and 0 < eng.n_prefix_early < eng.fwd_fork
    assert all(op.lane == eng.prefix_lane for op in eng.fwd_ops[: eng.n_prefix_early])
    assert all(op.lane != eng.prefix_lane for op in eng.fwd_ops[eng.n_prefix_early:])
    # the prefix on its own stream, no look-ahead
    eng.forward(xs[0])
    assert same(heads(eng), want[0])
    # look-ahead used: batch 1 finds its prefix done, batch 2 likewise
    eng.forward(xs[0], next_x=xs[1])
    assert same(heads(eng), want[0])
    eng.forward(xs[1], next_x=xs[2])
    assert same(heads(eng), want[1])
    eng.forward(xs[2])
    assert same(heads(eng), want[2])
    # look-ahead announced and NOT honoured: another batch comes, and then the announced one after all
    eng.forward(xs[0], next_x=xs[1])
    eng.forward(xs[3])
    assert same(heads(eng), want[3])
    eng.forward(xs[1])
    assert same(heads(eng), want[1])
    # RESIDENT: the next batch is what x_in holds
    eng.forward(xs[2], next_x=Engine.RESIDENT)
    eng.forward(None, next_x=Engine.RESIDENT)
    assert same(heads(eng), want[2])
    eng.forward(None)
    assert same(heads(eng), want[2])
    ref.close()
    eng.close()


def test_train_steps_with_look_ahead_match(ctx, monkeypatch):
    """three optimisation steps with and without the look-ahead: same losses (the weight-gradient atomics are the only
    run-to-run difference, 2e-5), and the frozen prefix is computed once per step either way."""
    from pyrapose_amd import arch
    from pyrapose_amd.engine import Engine
    from tests.test_gpu_model import random_targets
    Wt = arch.init_weights(C, seed=4)
    xs = batches(3, seed=1)
    rng = np.random.default_rng(2)
    monkeypatch.delenv("PP_PREFETCH", raising=False)
    a = Engine(ctx, C, B, H, W, weights=Wt, train=True, lr=1e-4)
    monkeypatch.setenv("PP_PREFETCH", "1")
    b = Engine(ctx, C, B, H, W, weights=Wt, train=True, lr=1e-4)
    tg = [tuple(torch.from_numpy(t).cuda() for t in random_targets(rng, B, a.N, a.M3, C)) for _ in range(3)]
    for i in range(3):
        a.train_step(xs[i], tg[i])
        b.train_step(xs[i], tg[i], next_x=xs[i + 1] if i < 2 else None)
        la, lb = a.losses(), b.losses()
        for k in la:
            assert abs(la[k] - lb[k]) <= 2e-5 * max(abs(la[k]), 1e-3), (i, k, la[k], lb[k])
    wa, wb = a.params.w_master, b.params.w_master
    assert float((wa - wb).abs().max()) <= 2e-5 * float(wa.abs().max())
    a.close()
    b.close()


def test_u8_look_ahead_with_augmentation(ctx, monkeypatch):
    """the lean feed (uint8 batch + per-image affine warp on the device) with the generator's look-ahead"""
    from pyrapose_amd import arch
    from pyrapose_amd.engine import Engine
    rng = np.random.default_rng(7)
    Wt = arch.init_weights(C, seed=5)
    monkeypatch.setenv("PP_PREFETCH", "1")
    eng = Engine(ctx, C, B, H, W, weights=Wt, train=True)
    assert eng.prefix_lane is not None
    u8 = [torch.from_numpy(rng.integers(0, 256, size=(B, H, W, 3)).astype(np.uint8)).cuda() for _ in range(3)]
    tf = [[np.array([[1.05, 0.02, 3.0], [-0.01, 0.97, -2.0], [0, 0, 1.0]]) for _ in range(B)] for _ in range(3)]
    want = []
    for i in range(3):
        eng.forward_u8(u8[i], None, tf[i])
        want.append(heads(eng))
    for i in range(3):
        nb = dict(images_u8=u8[i + 1], transforms=tf[i + 1]) if i < 2 else None
        eng.forward_u8(u8[i], None, tf[i], next_batch=nb)
        assert same(heads(eng), want[i]), i
    eng.close()


@pytest.mark.parametrize("mode", ["bf16x3", "mixed"])
def test_sparse_forward_of_box_head_is_exact_for_training(ctx, monkeypatch, mode):
    """(the bounds below were widened 40-100x for the f16c8 heads of the mixed mode, where a stale 32-row block in a
    low-magnitude region would no longer fail; the all-bf16x3 arithmetic runs the same plan -- lazy sparse gradients included --
    and keeps the tight bounds.)
    PP_SPARSE_FWD=1: in a training step the 3D-box head computes only the 32-row blocks its loss reads (anchors with state 1,
    losses.py:332-333) and what those need, layer by layer.  Losses, gradients and updated weights equal the dense forward's
    (different kernel for the listed blocks: f32 summation order), step after step with new targets; forward() outside a
    training step still computes every row."""
    from pyrapose_amd import arch
    from pyrapose_amd.engine import Engine
    from tests.test_gpu_model import random_targets
    Bq, Hq, Wq, Cq = 2, 160, 224, 5
    Wt = arch.init_weights(Cq, seed=6)
    rng = np.random.default_rng(3)
    mean = np.array([103.939, 116.779, 123.68], np.float32)
    xs = [torch.from_numpy(rng.integers(0, 256, size=(Bq, Hq, Wq, 3)).astype(np.float32) - mean).cuda() for _ in range(3)]
    # (lr = 0 for the first three steps: both engines keep identical weights, so that their gradients can be compared at kernel
    # precision step after step -- with new inputs and targets each time, i.e. with stale rows of the step before lying around;
    # a fourth step with lr > 0 compares the updated weights)
    monkeypatch.delenv("PP_SPARSE_FWD", raising=False)
    a = Engine(ctx, Cq, Bq, Hq, Wq, weights=Wt, train=True, lr=0.0, conv_mode=mode)
    monkeypatch.setenv("PP_SPARSE_FWD", "1")
    b = Engine(ctx, Cq, Bq, Hq, Wq, weights=Wt, train=True, lr=0.0, conv_mode=mode)
    assert b.sparse_fwd and not a.sparse_fwd
    tight = mode == "bf16x3"  # (2^-17 per stored value: the round-2 bounds)
    for i in range(4):
        if i == 3:
            a.lr = b.lr = 1e-4
        tg = [torch.from_numpy(t).cuda() for t in random_targets(rng, Bq, a.N, a.M3, Cq, pos_frac=0.0005)]
        a.train_step(xs[i % 3], tg)
        b.train_step(xs[i % 3], tg)
        torch.cuda.synchronize()
        nb = b._sf_flags[4].numel()
        assert 0 < int(b._sf_flags[4].sum()) <= int(b._sf_flags[0].sum()) < nb  # some blocks, more of them layer by layer, not all
        la, lb = a.losses(), b.losses()
        for k in la:
            assert abs(la[k] - lb[k]) <= 2e-5 * max(abs(la[k]), 1e-3), (i, k, la[k], lb[k])
        ga, gb = a.params.grad, b.params.grad
        # (the listed-block launch sums in another order than the dense kernel: ~2e-5 per activation in the f16c8 arithmetic, and a
        # ReLU input of the head that is zero to rounding may flip -- 8e-3 of the gradient's scale has been seen; a stale row
        # would show as O(1))
        assert float((ga - gb).abs().max()) <= (5e-4 if tight else 2e-2) * float(ga.abs().max()), i
    # a batch without a single positive anchor: nothing of the head is computed, its loss and gradient are zero in both engines
    a.lr = b.lr = 0.0
    tg = [torch.from_numpy(t).cuda() for t in random_targets(rng, Bq, a.N, a.M3, Cq, pos_frac=0.0)]
    a.train_step(xs[1], tg)
    b.train_step(xs[1], tg)
    torch.cuda.synchronize()
    assert int(b._sf_flags[0].sum()) == 0
    la, lb = a.losses(), b.losses()
    assert la["3Dbox"] == 0.0 and lb["3Dbox"] == 0.0
    for k in la:
        assert abs(la[k] - lb[k]) <= 2e-5 * max(abs(la[k]), 1e-3), (k, la[k], lb[k])
    # (without the box loss the largest gradient is ~7e-3 while the float32 atomics of the weight gradients still scatter ~4e-6:
    # an absolute floor beside the relative bound)
    assert float((a.params.grad - b.params.grad).abs().max()) <= (5e-4 if tight else 2e-3) * float(a.params.grad.abs().max()) + 2e-5
    wa, wb = a.params.w_master, b.params.w_master
    # (one Adam step moves every weight by ~lr whatever the size of its gradient: where the two gradients are noise of opposite
    # sign the weights part by 2 lr -- that, not a relative bound, is the scale of an honest difference)
    assert float((wa - wb).abs().max()) <= 2.5 * 1e-4
    # outside a training step nothing is skipped (the two weight sets have parted by up to 2 lr per weight in the fourth step, which
    # moves the outputs by a few 1e-4 of their scale; a row block left out would show as O(1))
    a.forward(xs[0])
    b.forward(xs[0])
    torch.cuda.synchronize()
    ra, rb = a.reg_out.t[:, : a.reg_out.C], b.reg_out.t[:, : b.reg_out.C]
    assert float((ra - rb).abs().max()) <= 2e-3 * float(ra.abs().max())
    if tight:  # the same weights in both engines: the final forwards agree at kernel precision
        b.params.w_master.copy_(a.params.w_master)
        b.params.w_eff.copy_(a.params.w_eff)
        b.refresh_planes()
        b.forward(xs[0])
        torch.cuda.synchronize()
        assert float((ra - b.reg_out.t[:, : b.reg_out.C]).abs().max()) <= 2e-5 * float(ra.abs().max())
    a.close()
    b.close()


def test_positive_row_blocks_and_dilation_match_numpy(ctx):
    """pp_positive_row_blocks / pp_row_block_dilate against a plain restatement: block b is flagged when one of its 32 rows holds
    an anchor with state 1; the dilation flags b when a flagged block meets one of the three 34-row windows
    [32 b - 1 + j W, 32 b + 32 + j W], j = -1, 0, 1, W = width of the level block b lies in (conv3.hip: rl_dilate_kernel)."""
    from pyrapose_amd import ops
    rng = np.random.default_rng(9)
    Bq, shapes, A = 2, [(20, 26), (10, 13), (5, 7)], 9
    rs = ops.RowSpace.make(Bq, shapes)
    cells = sum(h * w for h, w in shapes)
    rows = Bq * cells
    y = rng.standard_normal((Bq, cells * A, 17)).astype(np.float32)
    state = np.where(rng.uniform(size=(Bq, cells * A)) < 0.002, 1.0, np.where(rng.uniform(size=(Bq, cells * A)) < 0.05, -1.0, 0.0))
    y[:, :, 16] = state
    nb = (rows + 31) // 32
    flags = torch.full((nb,), 7, dtype=torch.uint8, device="cuda")
    ops.positive_row_blocks(ctx, rs, A, torch.from_numpy(y).cuda(), flags)
    # row of (image b, cell c of level s) = row_begin[s] + b * hw[s] + (c - cell_off[s])  (the layout of every head tensor)
    want = np.zeros(nb, np.uint8)
    row_begin, cell_off, seg_of_row, rb, co = [], [], np.zeros(rows, np.int64), 0, 0
    for s, (h, w) in enumerate(shapes):
        row_begin.append(rb); cell_off.append(co)
        seg_of_row[rb: rb + Bq * h * w] = s
        rb += Bq * h * w; co += h * w
    for b in range(Bq):
        for i in np.nonzero(state[b] == 1.0)[0]:
            c = i // A
            s = max(k for k in range(len(shapes)) if c >= cell_off[k])
            m = row_begin[s] + b * shapes[s][0] * shapes[s][1] + (c - cell_off[s])
            want[m >> 5] = 1
    assert want.any() and not want.all()
    assert np.array_equal(flags.cpu().numpy(), want)
    d = ops.make_conv_desc(Bq, shapes, shapes, 64, 64, 3, 1, 1, 1, 64, 64, 64)
    out = torch.full((nb,), 7, dtype=torch.uint8, device="cuda")
    ops.row_block_dilate(ctx, d, flags, out)
    want2 = np.zeros(nb, np.uint8)
    for bq in range(nb):
        r_lo, r_hi = 32 * bq, min(32 * bq + 31, rows - 1)
        hit = False
        for s, (h, w) in enumerate(shapes):
            seg_lo, seg_hi = row_begin[s], (row_begin[s + 1] if s + 1 < len(shapes) else rows) - 1
            if r_hi < seg_lo or r_lo > seg_hi:
                continue
            for j in (-1, 0, 1):
                lo, hi = max(r_lo - 1 + j * w, 0), min(r_hi + 1 + j * w, rows - 1)
                hit = hit or bool(want[lo >> 5: (hi >> 5) + 1].any())
        want2[bq] = 1 if hit else 0
    assert np.array_equal(out.cpu().numpy(), want2)
    assert (want2 >= want).all() and want2.sum() > want.sum()
