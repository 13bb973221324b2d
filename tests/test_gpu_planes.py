"""GPU: the planes-only storage format of the bf16x3 path (activations / gradients as bf16 (hi, lo) planes, value = hi + lo):
every kernel variant that reads or writes planes -- gathered operand, output tile, residual / addend / ReLU source of the
epilogues, split-K finish, stride-2 parity classes, row-list sparse backward, weight gradient over a block list, the
pointwise ops on tensor views -- against the SAME launch on float32 tensors.  The products are identical (same hi, lo);
a value read back from planes is within 2^-15 of the float32 it was split from, so results agree to ~3e-5 of the tensor's
magnitude; where no operand is read back from planes they agree to f32 summation order."""
import numpy as np
import pytest
import torch

from tests.test_gpu_conv import BF3_CASES, WG3_CASES, _cat_rows, _device_weight, _setup, rel_err, tf_same

pytestmark = pytest.mark.gpu


FMT = [0]  # plane format of the context the running test got (merged() / split() decode with it)


@pytest.fixture(scope="module", params=[0, 1], ids=["bf16x3", "f16c8"])
def ctx(request):
    """every test of this module once per plane format / arithmetic (csrc/planes_fmt.h): bf16 pairs, and P16"""
    from pyrapose_amd.runtime import default_context
    FMT[0] = request.param
    return default_context().twin(request.param)


@pytest.fixture(scope="module")
def ectx():
    from pyrapose_amd.runtime import default_context
    return default_context()


def split(ctx, t):
    from pyrapose_amd import ops
    hi, lo = ops.new_planes(t.shape[0], t.shape[1])
    ops.split_planes3(ctx, t, hi, lo)
    return hi, lo


def merged(pl):
    from pyrapose_amd import ops
    return ops.planes_to_f32(pl, FMT[0])


def nan_planes(like):
    from pyrapose_amd import ops
    return ops.new_planes(like.shape[0], like.shape[1], fill=0x7fc0)


def raw(pl):
    """the two planes as plain [rows, ld] int16 matrices (copies)"""
    return pl[0].reshape(pl[0].shape[0], -1), pl[1].reshape(pl[1].shape[0], -1)


def _geometry(case):
    from pyrapose_amd import ops
    name, B, shapes, cin, cout, k, stride, pad, ld_y, xs, w, bias = _setup(case, seed=3)
    out_shapes = []
    for (h, ww) in shapes:
        if pad == "same":
            out_shapes.append((-(-h // stride), -(-ww // stride)))
        else:
            out_shapes.append(((h + 2 * pad - k) // stride + 1, (ww + 2 * pad - k) // stride + 1))
    if pad == "same":
        pt, pl = tf_same(shapes[0][0], k, stride)[0], tf_same(shapes[0][1], k, stride)[0]
    else:
        pt = pl = pad
    ld_y = (ld_y or cout)
    ld_y = (ld_y + 31) // 32 * 32  # what the engine uses in bf16x3 mode (planes need ld % 8 == 0)
    wd, ld_w = _device_weight(w, cout)
    d = ops.make_conv_desc(B, shapes, out_shapes, cin, cout, k, stride, pt, pl, cin, ld_y, ld_w)
    taps = k * k
    u16 = dict(dtype=torch.int16, device="cuda")
    fh, fl = torch.zeros((taps, cout, cin), **u16), torch.zeros((taps, cout, cin), **u16)
    cred = (cout + 31) // 32 * 32
    dh, dl = torch.zeros((taps, cin, cred), **u16), torch.zeros((taps, cin, cred), **u16)
    return d, wd, ld_w, ld_y, cred, xs, bias, out_shapes, B, cin, cout


@pytest.mark.parametrize("case", BF3_CASES, ids=[c[0] for c in BF3_CASES])
def test_planes_only_fwd_and_bwd_data(ctx, case):
    from pyrapose_amd import ops
    d, wd, ld_w, ld_y, cred, xs, bias, out_shapes, B, cin, cout = _geometry(case)
    rng = np.random.default_rng(2)
    taps = d.kh * d.kw
    u16 = dict(dtype=torch.int16, device="cuda")
    fh, fl = torch.zeros((taps, cout, cin), **u16), torch.zeros((taps, cout, cin), **u16)
    dh, dl = torch.zeros((taps, cin, cred), **u16), torch.zeros((taps, cin, cred), **u16)
    ops.conv_split_weights3(ctx, d, wd, fh, fl, dh, dl)
    x = _cat_rows(xs)
    rows_out = sum(B * h * w for h, w in out_shapes)
    res = torch.zeros((rows_out, ld_y), dtype=torch.float32, device="cuda")
    res[:, :cout] = torch.as_tensor(rng.standard_normal((rows_out, cout)), dtype=torch.float32).cuda()
    bd = torch.zeros((ld_w,), dtype=torch.float32)
    bd[:cout] = torch.as_tensor(bias, dtype=torch.float32)
    bd = bd.cuda()
    # ---- forward: f32 everywhere vs planes everywhere
    y = torch.full((rows_out, ld_y), float("nan"), dtype=torch.float32, device="cuda")
    ops.conv_fwd3(ctx, d, x, fh, fl, bd, res, True, y)
    xp, rp = split(ctx, x), split(ctx, res)
    yp = nan_planes(y)
    ops.conv_fwd3(ctx, d, None, fh, fl, bd, None, True, None, x_planes=xp, y_planes=yp, res_planes=rp)
    got = merged(yp)[:, :cout]
    scale = float(y[:, :cout].abs().max())
    assert float((got - y[:, :cout]).abs().max()) <= 4e-5 * scale
    # the residual given as f32, output as planes only: exactly the split of the f32 output of the same kernel
    yp2 = nan_planes(y)
    y2 = torch.full_like(y, float("nan"))
    ops.conv_fwd3(ctx, d, None, fh, fl, bd, res, True, y2, x_planes=xp, y_planes=yp2)
    yp3 = nan_planes(y)
    ops.conv_fwd3(ctx, d, None, fh, fl, bd, res, True, None, x_planes=xp, y_planes=yp3)
    (a_h, a_l), (b_h, b_l) = raw(yp2), raw(yp3)
    assert torch.equal(a_h[:, :cout], b_h[:, :cout]) and torch.equal(a_l[:, :cout], b_l[:, :cout])
    wh, wl = raw(split(ctx, torch.nan_to_num(y2)))
    assert torch.equal(b_h[:, :cout], wh[:, :cout]) and torch.equal(b_l[:, :cout], wl[:, :cout])
    # ---- bwd-data: dy, addend, ReLU source as planes, dx as planes only
    gy = torch.zeros((rows_out, ld_y), dtype=torch.float32, device="cuda")
    gy[:, :cout] = torch.as_tensor(rng.standard_normal((rows_out, cout)), dtype=torch.float32).cuda()
    add = torch.as_tensor(rng.standard_normal(tuple(x.shape)), dtype=torch.float32).cuda()
    rsrc = torch.relu(torch.as_tensor(rng.standard_normal(tuple(x.shape)), dtype=torch.float32)).cuda()  # a post-ReLU activation
    dx = torch.full_like(x, float("nan"))
    ops.conv_bwd_data3(ctx, d, gy, dh, dl, add, rsrc, dx)
    gp, ap, mp = split(ctx, gy), split(ctx, add), split(ctx, rsrc)
    dxp = nan_planes(dx)
    ops.conv_bwd_data3(ctx, d, None, dh, dl, None, None, None, dy_planes=gp, dx_planes=dxp, addend_planes=ap, relu_src_hi=mp[0])
    got = merged(dxp)
    scale = float(dx.abs().max())
    assert float((got - dx).abs().max()) <= 4e-5 * scale
    assert torch.equal(got == 0, dx == 0) or float(((got == 0) != (dx == 0)).float().mean()) < 1e-4  # the same ReLU mask
    # without addend / mask
    dx0 = torch.full_like(x, float("nan"))
    ops.conv_bwd_data3(ctx, d, gy, dh, dl, None, None, dx0)
    dxp0 = nan_planes(dx)
    ops.conv_bwd_data3(ctx, d, None, dh, dl, None, None, None, dy_planes=gp, dx_planes=dxp0)
    assert float((merged(dxp0) - dx0).abs().max()) <= 4e-5 * max(float(dx0.abs().max()), 1e-30)


@pytest.mark.parametrize("splits", [3])
def test_planes_only_split_k(ctx, splits, monkeypatch):
    from pyrapose_amd import ops
    rng = np.random.default_rng(11)
    B, H, W, cin, cout, k = 2, 9, 13, 128, 96, 3
    ld_y = 96
    d = ops.make_conv_desc(B, [(H, W)], [(H, W)], cin, cout, k, 1, 1, 1, cin, ld_y, 96)
    rows = B * H * W
    x = torch.as_tensor(rng.standard_normal((rows, cin)), dtype=torch.float32).cuda()
    w = torch.as_tensor(rng.standard_normal((k * k * cin, 96)) * 0.05, dtype=torch.float32).cuda()
    bias = torch.as_tensor(rng.standard_normal((96,)), dtype=torch.float32).cuda()
    res = torch.as_tensor(rng.standard_normal((rows, ld_y)), dtype=torch.float32).cuda()
    i16 = dict(dtype=torch.int16, device="cuda")
    fh, fl = torch.zeros((k * k, cout, cin), **i16), torch.zeros((k * k, cout, cin), **i16)
    dh, dl = torch.zeros((k * k, cin, ld_y), **i16), torch.zeros((k * k, cin, ld_y), **i16)
    ops.conv_split_weights3(ctx, d, w, fh, fl, dh, dl)
    gy = torch.as_tensor(rng.standard_normal((rows, ld_y)), dtype=torch.float32).cuda()
    add = torch.as_tensor(rng.standard_normal((rows, cin)), dtype=torch.float32).cuda()
    msk = torch.relu(torch.as_tensor(rng.standard_normal((rows, cin)), dtype=torch.float32)).cuda()
    xp, rp, gp, ap, mp = (split(ctx, t) for t in (x, res, gy, add, msk))

    def run():
        yp, dxp = nan_planes(res), nan_planes(x)
        ops.conv_fwd3(ctx, d, None, fh, fl, bias, None, True, None, x_planes=xp, y_planes=yp, res_planes=rp)
        ops.conv_bwd_data3(ctx, d, None, dh, dl, None, None, None, dy_planes=gp, dx_planes=dxp, addend_planes=ap, relu_src_hi=mp[0])
        torch.cuda.synchronize()
        return merged(yp), merged(dxp)

    monkeypatch.setenv("PP_CONV3_SPLITS", "1")
    y1, dx1 = run()
    ctx.set_workspace(splits * rows * max(ld_y, cin) * 4)
    try:
        monkeypatch.setenv("PP_CONV3_SPLITS", str(splits))
        ctx.workspace.fill_(float("nan"))
        y2, dx2 = run()
        y3, dx3 = run()
        assert torch.equal(y2, y3) and torch.equal(dx2, dx3)  # deterministic
    finally:
        ctx.set_workspace(0)
    assert float((y2 - y1).abs().max()) <= 4e-5 * float(y1.abs().max())
    assert float((dx2 - dx1).abs().max()) <= 4e-5 * float(dx1.abs().max())
    assert bool(y1.isfinite().all()) and bool(dx1.isfinite().all()) and float(y1.abs().max()) > 0


@pytest.mark.parametrize("frac", [0.0, 0.03])
def test_planes_only_row_block_skip(ctx, frac):
    """sparse backward (3D-box head) on planes: block list from planes, row-list bwd-data and block-list bwd-weight"""
    from pyrapose_amd import ops
    rng = np.random.default_rng(17)
    B, shapes, cin, cout, k = 2, [(20, 26), (10, 13), (5, 7)], 128, 64, 3
    rows = sum(B * h * w for h, w in shapes)
    d = ops.make_conv_desc(B, shapes, shapes, cin, cout, k, 1, 1, 1, cin, cout, cout)
    x = torch.as_tensor(rng.standard_normal((rows, cin)), dtype=torch.float32).cuda()
    w = torch.as_tensor(rng.standard_normal((k * k * cin, cout)) * 0.05, dtype=torch.float32).cuda()
    dy = torch.zeros((rows, cout), dtype=torch.float32, device="cuda")
    live = torch.as_tensor(rng.uniform(size=rows) < frac).cuda()
    dy[live] = torch.as_tensor(rng.standard_normal((int(live.sum()), cout)), dtype=torch.float32).cuda()
    i16 = dict(dtype=torch.int16, device="cuda")
    fh, fl = torch.zeros((k * k, cout, cin), **i16), torch.zeros((k * k, cout, cin), **i16)
    dh, dl = torch.zeros((k * k, cin, cout), **i16), torch.zeros((k * k, cin, cout), **i16)
    ops.conv_split_weights3(ctx, d, w, fh, fl, dh, dl)
    flags, blocks = ops.row_block_list(ctx, dy, cout)
    gp, xp = split(ctx, dy), split(ctx, x)
    f2, b2 = torch.zeros_like(flags), torch.zeros_like(blocks)
    ops.row_block_list_planes(ctx, gp, cout, f2, b2)
    nb = (rows + 31) // 32
    assert torch.equal(f2[:nb], flags[:nb]) and torch.equal(b2[: nb + 1], blocks[: nb + 1])
    add = torch.as_tensor(rng.standard_normal((rows, cin)), dtype=torch.float32).cuda()
    msk = torch.relu(torch.as_tensor(rng.standard_normal((rows, cin)), dtype=torch.float32)).cuda()
    ap, mp = split(ctx, add), split(ctx, msk)
    dx0 = torch.full((rows, cin), float("nan"), device="cuda")
    ops.conv_bwd_data3(ctx, d, dy, dh, dl, add, msk, dx0)
    dxp = nan_planes(dx0)
    ops.conv_bwd_data3(ctx, d, None, dh, dl, None, None, None, dy_planes=gp, dx_planes=dxp, dy_skip=(f2, b2), addend_planes=ap,
                       relu_src_hi=mp[0])
    assert float((merged(dxp) - dx0).abs().max()) <= 4e-5 * float(dx0.abs().max())
    assert not bool(f2[nb: 2 * nb].all())  # some blocks were filled, not computed
    # in place on the addend (no ReLU mask): the rows no non-zero reaches are not touched at all, the others get the same sum
    dx1 = torch.full((rows, cin), float("nan"), device="cuda")
    ops.conv_bwd_data3(ctx, d, dy, dh, dl, add, None, dx1)
    acc = split(ctx, add)
    ops.conv_bwd_data3(ctx, d, None, dh, dl, None, None, None, dy_planes=gp, dx_planes=acc, dy_skip=(f2, b2), addend_planes=acc)
    assert float((merged(acc) - dx1).abs().max()) <= 7e-5 * float(dx1.abs().max())  # (the sum is rounded to the plane format twice)
    # without an addend the result is zero outside the blocks the launch flagged in the second half of its scratch: a scan
    # restricted to those blocks (pp_row_block_list_planes_within) finds what the full scan finds
    dxz = nan_planes(dx0)
    ops.conv_bwd_data3(ctx, d, None, dh, dl, None, None, None, dy_planes=gp, dx_planes=dxz, dy_skip=(f2, b2), relu_src_hi=mp[0])
    within = f2[nb: 2 * nb].clone()
    fa, ba = torch.zeros_like(flags), torch.zeros_like(blocks)
    fb, bb = torch.zeros_like(flags), torch.zeros_like(blocks)
    ops.row_block_list_planes(ctx, dxz, cin, fa, ba)
    ops.row_block_list_planes(ctx, dxz, cin, fb, bb, within=within)
    assert torch.equal(fa[:nb], fb[:nb]) and torch.equal(ba[: nb + 1], bb[: nb + 1])
    assert not bool((fa[:nb].bool() & ~within.bool()).any())
    # a launch that takes the hint but cannot run over the listed blocks (here: 1x1) flags every block as possibly non-zero
    d1 = ops.make_conv_desc(B, shapes, shapes, cin, cout, 1, 1, 0, 0, cin, cout, cout)
    w1h, w1l = torch.zeros((1, cout, cin), **i16), torch.zeros((1, cout, cin), **i16)
    d1h, d1l = torch.zeros((1, cin, cout), **i16), torch.zeros((1, cin, cout), **i16)
    ops.conv_split_weights3(ctx, d1, w[:cin].contiguous(), w1h, w1l, d1h, d1l)
    f3, b3 = f2.clone(), b2.clone()
    f3[nb:] = 0
    ops.conv_bwd_data3(ctx, d1, None, d1h, d1l, None, None, None, dy_planes=gp, dx_planes=nan_planes(dx0), dy_skip=(f3, b3))
    assert bool(f3[nb: 2 * nb].all())
    dw0, dw1 = torch.zeros_like(w), torch.zeros_like(w)
    db0, db1 = torch.zeros((cout,), device="cuda"), torch.zeros((cout,), device="cuda")
    ops.conv_bwd_weight3(ctx, d, x, dy, dw0, db0)
    ops.conv_bwd_weight3(ctx, d, None, None, dw1, db1, x_planes=xp, dy_planes=gp, dy_skip=(f2, b2))
    scale = max(float(dw0.abs().max()), 1e-30)
    assert float((dw0 - dw1).abs().max()) <= 2e-6 * scale
    assert float((db0 - db1).abs().max()) <= 4e-5 * max(float(db0.abs().max()), 1e-30)
    if frac == 0.0:
        assert not dw1.any() and not db1.any()


@pytest.mark.parametrize("in_place", [False, True])
def test_row_list_bwd_data_split_k(ctx, in_place):
    """the listed-block data gradient with a scratch buffer: the reduction is split two ways (few listed tiles, long k-loops)
    and the slices are added per listed block; same result as the dense launch, with addend + ReLU mask or in place"""
    from pyrapose_amd import ops
    rng = np.random.default_rng(23)
    B, shapes, cin, cout, k = 2, [(20, 26), (10, 13), (5, 7)], 128, 256, 3
    rows = sum(B * h * w for h, w in shapes)
    d = ops.make_conv_desc(B, shapes, shapes, cin, cout, k, 1, 1, 1, cin, cout, cout)
    w = torch.as_tensor(rng.standard_normal((k * k * cin, cout)) * 0.05, dtype=torch.float32).cuda()
    dy = torch.zeros((rows, cout), dtype=torch.float32, device="cuda")
    live = torch.as_tensor(rng.uniform(size=rows) < 0.01).cuda()
    dy[live] = torch.as_tensor(rng.standard_normal((int(live.sum()), cout)), dtype=torch.float32).cuda()
    i16 = dict(dtype=torch.int16, device="cuda")
    fh, fl = torch.zeros((k * k, cout, cin), **i16), torch.zeros((k * k, cout, cin), **i16)
    dh, dl = torch.zeros((k * k, cin, cout), **i16), torch.zeros((k * k, cin, cout), **i16)
    ops.conv_split_weights3(ctx, d, w, fh, fl, dh, dl)
    gp = split(ctx, dy)
    flags, blocks = ops.row_block_list(ctx, dy, cout)
    nb = (rows + 31) // 32
    add = torch.as_tensor(rng.standard_normal((rows, cin)), dtype=torch.float32).cuda()
    msk = None if in_place else torch.relu(torch.as_tensor(rng.standard_normal((rows, cin)), dtype=torch.float32)).cuda()
    want = torch.full((rows, cin), float("nan"), device="cuda")
    ops.conv_bwd_data3(ctx, d, dy, dh, dl, add, msk, want)  # dense, float32 operands
    ap = split(ctx, add)
    outs = []
    for ws_mb in (0, 16):
        ctx.set_workspace(ws_mb << 20)
        try:
            if ws_mb:
                ctx.workspace.fill_(float("nan"))
            f2, b2 = flags.clone(), blocks.clone()
            if in_place:
                acc = split(ctx, add)
                ops.conv_bwd_data3(ctx, d, None, dh, dl, None, None, None, dy_planes=gp, dx_planes=acc, dy_skip=(f2, b2), addend_planes=acc)
                got = merged(acc)
            else:
                dxp = nan_planes(want)
                ops.conv_bwd_data3(ctx, d, None, dh, dl, None, None, None, dy_planes=gp, dx_planes=dxp, dy_skip=(f2, b2), addend_planes=ap,
                                   relu_src_hi=split(ctx, msk)[0])
                got = merged(dxp)
            torch.cuda.synchronize()
            assert 0 < int(f2[nb: 2 * nb].sum()) < nb  # the listed launch ran, and not over everything
            assert float((got - want).abs().max()) <= 4e-5 * float(want.abs().max()), ws_mb
            outs.append(got)
        finally:
            ctx.set_workspace(0)
    assert float((outs[0] - outs[1]).abs().max()) <= 4e-5 * float(want.abs().max())  # (two partial sums instead of one; planes hold 2^-15)


def test_pointwise_ops_on_views(ctx):
    from pyrapose_amd import ops
    rng = np.random.default_rng(5)
    B, C = 2, 64
    for (sh, sw, th, tw) in ((3, 4, 6, 8), (17, 23, 34, 45)):
        src = torch.as_tensor(rng.standard_normal((B * sh * sw, C)), dtype=torch.float32).cuda()
        oth = torch.as_tensor(rng.standard_normal((B * th * tw, C)), dtype=torch.float32).cuda()
        want = torch.empty_like(oth)
        ops.upsample_add_fwd(ctx, B, sh, sw, th, tw, C, src, oth, want)
        sp, op_ = split(ctx, src), split(ctx, oth)
        outp = nan_planes(oth)
        outf = torch.empty_like(oth)
        ops.upsample_add_fwd_v(ctx, B, sh, sw, th, tw, C, ops.tview(None, sp), ops.tview(None, op_), ops.tview(outf, outp))
        assert float((outf - want).abs().max()) <= 4e-5 * float(want.abs().max())
        wh, wl = split(ctx, outf)
        assert torch.equal(outp[0], wh) and torch.equal(outp[1], wl)  # both output formats hold the same values
        g = torch.as_tensor(rng.standard_normal((B * th * tw, C)), dtype=torch.float32).cuda()
        gs_want = torch.empty_like(src)
        ops.upsample_add_bwd(ctx, B, sh, sw, th, tw, C, g, None, gs_want)
        gsp = nan_planes(src)
        ops.upsample_add_bwd_v(ctx, B, sh, sw, th, tw, C, ops.tview(None, split(ctx, g)), None, ops.tview(None, gsp))
        assert float((merged(gsp) - gs_want).abs().max()) <= 4e-5 * float(gs_want.abs().max())
    a = torch.as_tensor(rng.standard_normal((1000, 8)), dtype=torch.float32).cuda()
    b = torch.as_tensor(rng.standard_normal((1000, 8)), dtype=torch.float32).cuda()
    outp = nan_planes(a)
    ops.add_n_v(ctx, ops.tview(a), ops.tview(None, split(ctx, b)), None, ops.tview(None, outp))
    assert float((merged(outp) - (a + b)).abs().max()) <= 4e-5 * float((a + b).abs().max())
    out = torch.empty_like(a)
    ops.add_n_v(ctx, ops.tview(a), ops.tview(b), ops.tview(a), ops.tview(out))
    assert torch.equal(out, (a + b) + a)
    yp = nan_planes(a)
    ops.relu_fwd_v(ctx, ops.tview(None, split(ctx, a)), ops.tview(None, yp))
    assert torch.equal(merged(yp) > 0, a > 0)
    back = torch.empty_like(a)
    ops.merge_planes3(ctx, split(ctx, a), back)
    assert float((back - a).abs().max()) <= 2 ** -16 * float(a.abs().max())
    with pytest.raises(ValueError):
        ops.add_n_v(ctx, ops.tview(a, split(ctx, a)), None, None, ops.tview(out))  # an input view is f32 OR planes
    with pytest.raises(ValueError):  # two separate planes are not the packed layout
        sep = (torch.zeros_like(a, dtype=torch.int16), torch.zeros_like(a, dtype=torch.int16))
        ops.split_planes3(ctx, a, sep[0], sep[1])


@pytest.mark.parametrize("mode", ["1", "0"])
def test_engine_planes_mode_matches_f32_storage(ectx, mode, monkeypatch):
    """The whole training step with planes-only storage (default) vs float32 storage with in-loop splits (PP_PLANES=0): same
    products; residual / addend reads differ by 2^-15 -> every output agrees to ~1e-4."""
    from pyrapose_amd import arch
    from pyrapose_amd.engine import Engine
    from tests.test_gpu_model import random_targets, synth_input
    B, H, W, C = 2, 96, 128, 5
    rng = np.random.default_rng(71)
    Wt = arch.init_weights(C, seed=72)
    x = torch.from_numpy(synth_input(rng, B, H, W)).cuda()
    monkeypatch.setenv("PP_PLANES", mode)
    eng = Engine(ectx, C, B, H, W, weights=Wt, train=True, conv_mode="bf16x3")
    assert eng.po == (mode == "1")
    if mode == "1":
        assert eng.C3.t is None and eng.C3.pl is not None and eng.pyr.t is None and eng.reg_out.t is not None
    tg = [torch.from_numpy(a).cuda() for a in random_targets(rng, B, eng.N, eng.M3, C, pos_frac=0.02)]
    eng.set_targets(*tg)
    eng.forward(x)
    eng.loss_and_backward()
    torch.cuda.synchronize()
    key = (eng.reg_out.t[:, : eng.reg_out.C].clone(), eng.cls_out.t[:, : eng.cls_out.C].clone(), eng.C5.f32(eng.ctx).clone(),
           eng.params.grad.clone(), eng.losses())
    store = test_engine_planes_mode_matches_f32_storage.__dict__.setdefault("runs", {})
    store[mode] = key
    if len(store) == 2:
        a, b = store["1"], store["0"]
        for i in range(3):
            assert float((a[i] - b[i]).abs().max()) <= 2e-4 * float(b[i].abs().max()), i  # (2^-15 per stored tensor, ~60 layers)
        ga, gb = a[3].double(), b[3].double()
        assert float((ga - gb).norm() / gb.norm()) < 5e-2  # (a ReLU input that is zero to rounding may flip between the modes)
        for k in ("3Dbox", "cls", "mask"):
            assert abs(a[4][k] - b[4][k]) <= 4e-5 * abs(b[4][k])


@pytest.mark.parametrize("case", [c for c in WG3_CASES if c[0] in ("head3x3_multilevel", "head_out_cout117", "lat1x1", "down3x3s2_odd", "x_five_levels")],
                         ids=lambda c: c[0])
def test_weight_gradient_slices_are_deterministic_and_match_atomics(ctx, case, monkeypatch):
    """PP_WGRAD3_DETERMINISTIC=1 + a scratch buffer: the row splits of the weight gradient write slices that a finishing pass
    adds in a fixed order (no float atomics): bit-identical from run to run, equal to the atomic path up to summation order,
    dw += semantics kept.  (Off by default: 1-2 % slower in the training step.)"""
    from pyrapose_amd import ops
    d, wd, ld_w, ld_y, cred, xs, bias, out_shapes, B, cin, cout = _geometry(case)
    rng = np.random.default_rng(4)
    x = _cat_rows(xs)
    rows_out = sum(B * h * w for h, w in out_shapes)
    gy = torch.zeros((rows_out, ld_y), dtype=torch.float32, device="cuda")
    gy[:, :cout] = torch.as_tensor(rng.standard_normal((rows_out, cout)), dtype=torch.float32).cuda()
    xp, gp = split(ctx, x), split(ctx, gy)
    base = torch.as_tensor(rng.standard_normal(tuple(wd.shape)), dtype=torch.float32).cuda()  # dw += : start from something

    def run(planes):
        dw, db = base.clone(), torch.ones((ld_w,), dtype=torch.float32, device="cuda")
        if planes:
            ops.conv_bwd_weight3(ctx, d, None, None, dw, db, x_planes=xp, dy_planes=gp)
        else:
            ops.conv_bwd_weight3(ctx, d, x, gy, dw, db)
        torch.cuda.synchronize()
        return dw, db
    monkeypatch.setenv("PP_WGRAD3_SPLITS", "3")
    monkeypatch.setenv("PP_WGRAD3_DETERMINISTIC", "1")
    a_dw, a_db = run(True)                       # no scratch buffer: atomics
    ctx.set_workspace(64 << 20)
    try:
        ctx.workspace.fill_(float("nan"))        # any contents: every word the finishing pass reads was written by the launch
        s_dw, s_db = run(True)
        s_dw2, s_db2 = run(True)
        f_dw, f_db = run(False)                  # float32 operands, same reduction
    finally:
        ctx.set_workspace(0)
    assert torch.equal(s_dw, s_dw2) and torch.equal(s_db, s_db2)
    scale = float((a_dw - base).abs().max())
    assert float((s_dw - a_dw).abs().max()) <= 2e-6 * scale and float((f_dw - a_dw).abs().max()) <= 2e-6 * scale
    assert float((s_db - a_db).abs().max()) <= 4e-5 * max(float((a_db - 1).abs().max()), 1e-30)
    assert bool(s_dw.isfinite().all()) and float((s_dw - base).abs().max()) > 0
    assert torch.equal(s_dw[:, cout:], base[:, cout:])   # padding columns: + 0


def test_lazy_sparse_gradients(ctx):
    """pp_ctx_set_row_block_lazy: a listed-block data gradient that leaves the rows outside its blocks UNWRITTEN, and consumers that
    never look at them -- the restricted scan, the listed-block data gradient (its gather skips rows outside the flagged blocks of
    dy) and the listed-block weight gradient give what they give on the zero-filled tensor; a launch that would read the unwritten
    rows refuses the call."""
    from pyrapose_amd import ops
    rng = np.random.default_rng(23)
    B, shapes, c1, c2, k = 2, [(20, 26), (10, 13), (6, 7)], 128, 64, 3
    rows = sum(B * h * w for h, w in shapes)
    nb = (rows + 31) // 32
    i16 = dict(dtype=torch.int16, device="cuda")
    d_b = ops.make_conv_desc(B, shapes, shapes, c1, c2, k, 1, 1, 1, c1, c2, c2)  # layer b: c1 -> c2 (its dy is sparse, its dx lazy)
    d_a = ops.make_conv_desc(B, shapes, shapes, c1, c1, k, 1, 1, 1, c1, c1, c1)  # layer a in front of it: c1 -> c1 (reads the lazy tensor)
    wb = torch.as_tensor(rng.standard_normal((k * k * c1, c2)) * 0.05, dtype=torch.float32).cuda()
    wa = torch.as_tensor(rng.standard_normal((k * k * c1, c1)) * 0.05, dtype=torch.float32).cuda()
    bh, bl = torch.zeros((k * k, c1, c2), **i16), torch.zeros((k * k, c1, c2), **i16)
    ah, al = torch.zeros((k * k, c1, c1), **i16), torch.zeros((k * k, c1, c1), **i16)
    ops.conv_split_weights3(ctx, d_b, wb, torch.zeros((k * k, c2, c1), **i16), torch.zeros((k * k, c2, c1), **i16), bh, bl)
    ops.conv_split_weights3(ctx, d_a, wa, torch.zeros((k * k, c1, c1), **i16), torch.zeros((k * k, c1, c1), **i16), ah, al)
    dy = torch.zeros((rows, c2), dtype=torch.float32, device="cuda")
    live = torch.as_tensor(rng.uniform(size=rows) < 0.01).cuda()
    dy[live] = torch.as_tensor(rng.standard_normal((int(live.sum()), c2)), dtype=torch.float32).cuda()
    gp = split(ctx, dy)
    msk = split(ctx, torch.relu(torch.as_tensor(rng.standard_normal((rows, c1)), dtype=torch.float32)).cuda())
    xa = split(ctx, torch.as_tensor(rng.standard_normal((rows, c1)), dtype=torch.float32).cuda())
    res = {}
    for lazy in (False, True):
        f, b = ops.row_block_list(ctx, dy, c2)
        g1 = nan_planes(torch.empty((rows, c1)))
        ops.conv_bwd_data3(ctx, d_b, None, bh, bl, None, None, None, dy_planes=gp, dx_planes=g1, dy_skip=(f, b), relu_src_hi=msk[0], lazy_out=lazy)
        within = f[nb: 2 * nb].clone()
        assert 0 < int(within.sum()) < nb
        f1, b1 = torch.zeros_like(f), torch.zeros_like(b)
        ops.row_block_list_planes(ctx, g1, c1, f1, b1, within=within)
        g0 = nan_planes(torch.empty((rows, c1)))
        ops.conv_bwd_data3(ctx, d_a, None, ah, al, None, None, None, dy_planes=g1, dx_planes=g0, dy_skip=(f1, b1), lazy_in=lazy)
        dw, db = torch.zeros_like(wa), torch.zeros((c1,), device="cuda")
        ops.conv_bwd_weight3(ctx, d_a, None, None, dw, db, x_planes=xa, dy_planes=g1, dy_skip=(f1, b1), lazy_in=lazy)
        torch.cuda.synchronize()
        res[lazy] = (raw(g1), within, f1[:nb].clone(), b1[: nb + 1].clone(), raw(g0), dw, db)
    (z1, wz, fz, bz, z0, dwz, dbz), (l1, wl, fl_, bl_, l0, dwl, dbl) = res[False], res[True]
    assert torch.equal(wz, wl) and torch.equal(fz, fl_) and torch.equal(bz, bl_)
    rows_in = wz.bool().repeat_interleave(32)[:rows]
    for a, b_ in zip(z1, l1):  # the computed blocks are the same bits; the others: zeros when filled, the NaN pattern when lazy
        assert torch.equal(a[rows_in], b_[rows_in])
        assert not a[~rows_in].any()
    assert bool((l1[0][~rows_in] == 0x7fc0).all())
    for a, b_ in zip(z0, l0):  # the next layer's data gradient never fetched an unwritten row
        assert torch.equal(a, b_)
    assert float((dwz - dwl).abs().max()) <= 2e-6 * float(dwz.abs().max()) and not torch.isnan(dwl).any()
    assert float((dbz - dbl).abs().max()) <= 1e-5 * max(float(dbz.abs().max()), 1e-30)
    # refusals: a 1x1 data gradient and a weight gradient without a block list cannot honour a lazy dy
    d1 = ops.make_conv_desc(B, shapes, shapes, c1, c1, 1, 1, 0, 0, c1, c1, c1)
    w1h, w1l = torch.zeros((1, c1, c1), **i16), torch.zeros((1, c1, c1), **i16)
    f1, b1 = ops.row_block_list(ctx, dy, c2)
    with pytest.raises(ValueError):
        ops.conv_bwd_data3(ctx, d1, None, w1h, w1l, None, None, None, dy_planes=xa, dx_planes=nan_planes(torch.empty((rows, c1))), dy_skip=(f1, b1),
                           lazy_in=True)
    with pytest.raises(ValueError):
        ops.conv_bwd_weight3(ctx, d_a, None, None, torch.zeros_like(wa), None, x_planes=xa, dy_planes=xa, lazy_in=True)


def test_lds_dma_kernel_matches_register_staged(ctx, monkeypatch):
    """igemm4x (csrc/conv3.hip): the head-conv launch on the multi-stage LDS-DMA pipeline -- taken for planes in / planes out, 3x3
    stride 1, cout % 128 == 0 and at least two rounds of 256 x 128 tiles, with the rows of the last partly filled round going to a
    split-K launch of the 128 x 128 kernel.  Same products in the same order as the register-staged igemm3x (PP_CONV3_DMA=0, read at
    every launch): the whole-round rows are bit-identical, the split-K rows agree to f32 summation order; forward with bias +
    residual + ReLU and data gradient with addend + ReLU mask; within the per-launch bound of the arithmetic against float64."""
    import torch.nn.functional as F
    from pyrapose_amd import ops
    rng = np.random.default_rng(31)
    B, shapes, cin, cout, k = 1, [(262, 128)], 128, 512, 3    # 33 536 rows: 133 x 4 = 532 tiles = 2 rounds + 20 (tail path)
    rows = sum(B * h * w for h, w in shapes)
    d = ops.make_conv_desc(B, shapes, shapes, cin, cout, k, 1, 1, 1, cin, cout, cout)
    dT = ops.make_conv_desc(B, shapes, shapes, cout, cout, k, 1, 1, 1, cout, cout, cout)  # square layer for the data gradient
    x = torch.as_tensor(rng.standard_normal((rows, cin)), dtype=torch.float32).cuda()
    w = torch.as_tensor(rng.standard_normal((k * k * cin, cout)) * 0.05, dtype=torch.float32).cuda()
    w2 = torch.as_tensor(rng.standard_normal((k * k * cout, cout)) * 0.02, dtype=torch.float32).cuda()
    bias = torch.as_tensor(rng.standard_normal((cout,)), dtype=torch.float32).cuda()
    res = torch.as_tensor(rng.standard_normal((rows, cout)), dtype=torch.float32).cuda()
    dy = torch.as_tensor(rng.standard_normal((rows, cout)), dtype=torch.float32).cuda()
    i16 = dict(dtype=torch.int16, device="cuda")
    fh, fl = torch.zeros((k * k, cout, cin), **i16), torch.zeros((k * k, cout, cin), **i16)
    dh, dl = torch.zeros((k * k, cin, cout), **i16), torch.zeros((k * k, cin, cout), **i16)
    ops.conv_split_weights3(ctx, d, w, fh, fl, dh, dl)
    f2h, f2l = torch.zeros((k * k, cout, cout), **i16), torch.zeros((k * k, cout, cout), **i16)
    d2h, d2l = torch.zeros((k * k, cout, cout), **i16), torch.zeros((k * k, cout, cout), **i16)
    ops.conv_split_weights3(ctx, dT, w2, f2h, f2l, d2h, d2l)
    xp, rp, gp = split(ctx, x), split(ctx, res), split(ctx, dy)
    out = {}
    ctx.set_workspace(64 << 20)
    try:
        for dma in ("0", "1"):
            monkeypatch.setenv("PP_CONV3_DMA", dma)
            yp = nan_planes(res)
            ops.conv_fwd3(ctx, d, None, fh, fl, bias, None, True, None, x_planes=xp, y_planes=yp, res_planes=rp)
            dxp = nan_planes(res)
            ops.conv_bwd_data3(ctx, dT, None, d2h, d2l, None, None, None, dy_planes=gp, dx_planes=dxp, addend_planes=rp, relu_src_hi=rp[0])
            torch.cuda.synchronize()
            out[dma] = (raw(yp), raw(dxp), merged(yp), merged(dxp))
    finally:
        ctx.set_workspace(0)
        monkeypatch.delenv("PP_CONV3_DMA", raising=False)
    full = 2 * 256 // 4 * 254            # rows of the two whole rounds: 128 row tiles of 254 rows
    for i in (0, 1):
        (ah, al), (bh, bl) = out["0"][i], out["1"][i]
        assert torch.equal(ah[:full], bh[:full]) and torch.equal(al[:full], bl[:full]), i
        va, vb = out["0"][2 + i], out["1"][2 + i]
        assert not torch.isnan(vb).any()
        # (split-K rows: another summation order, then re-encoded: one step of the plane format's remainder, 2^-15 in P16)
        assert float((va - vb).abs().max()) <= (8e-5 if FMT[0] == 1 else 4e-5) * float(va.abs().max()), i
    # float64 reference of the forward
    xv = merged(xp)
    xi = xv.double().reshape(1, 262, 128, cin).permute(0, 3, 1, 2)
    wt = w.double().reshape(k, k, cin, cout).permute(3, 2, 0, 1)
    ref = torch.relu(F.conv2d(xi, wt, bias.double(), padding=1).permute(0, 2, 3, 1).reshape(-1, cout) + merged(rp).double())
    err = (out["1"][2].double() - ref).abs().max() / ref.abs().max()
    assert float(err) <= (1e-4 if FMT[0] == 1 else 3e-5), float(err)


def test_lds_dma_kernel_two_workgroups_per_cu_matches_register_staged(ctx, monkeypatch):
    """Round 4, igemm4x<NWM = 2> (csrc/conv3.hip): the 128 x 128 form of the LDS-DMA pipeline, two workgroups per CU, gathered ring
    2 x 16 KB + weight ring 2 x 16 KB -- taken by launches between one round of 128-row tiles and two rounds of 256-row tiles (the
    256-channel heads, the FPN 3x3).  Same products in the same order as igemm3x: EVERY row bit-identical (PP_CONV3_DMA2=0 / 1,
    read per launch); forward with bias + residual + ReLU over a three-level row space, data gradient with addend + mask."""
    from pyrapose_amd import ops
    rng = np.random.default_rng(41)
    B, shapes, cin, cout, k = 2, [(60, 80), (30, 40), (15, 20)], 64, 256, 3     # 12 600 rows -> 100 row tiles x 2 = 200 tiles
    rows = sum(B * h * w for h, w in shapes)
    d = ops.make_conv_desc(B, shapes, shapes, cin, cout, k, 1, 1, 1, cin, cout, cout)
    dT = ops.make_conv_desc(B, shapes, shapes, cout, cout, k, 1, 1, 1, cout, cout, cout)
    x = torch.as_tensor(rng.standard_normal((rows, cin)), dtype=torch.float32).cuda()
    w = torch.as_tensor(rng.standard_normal((k * k * cin, cout)) * 0.05, dtype=torch.float32).cuda()
    w2 = torch.as_tensor(rng.standard_normal((k * k * cout, cout)) * 0.02, dtype=torch.float32).cuda()
    bias = torch.as_tensor(rng.standard_normal((cout,)), dtype=torch.float32).cuda()
    res = torch.as_tensor(rng.standard_normal((rows, cout)), dtype=torch.float32).cuda()
    dy = torch.as_tensor(rng.standard_normal((rows, cout)), dtype=torch.float32).cuda()
    i16 = dict(dtype=torch.int16, device="cuda")
    fh, fl = torch.zeros((k * k, cout, cin), **i16), torch.zeros((k * k, cout, cin), **i16)
    dh, dl = torch.zeros((k * k, cin, cout), **i16), torch.zeros((k * k, cin, cout), **i16)
    ops.conv_split_weights3(ctx, d, w, fh, fl, dh, dl)
    f2h, f2l = torch.zeros((k * k, cout, cout), **i16), torch.zeros((k * k, cout, cout), **i16)
    d2h, d2l = torch.zeros((k * k, cout, cout), **i16), torch.zeros((k * k, cout, cout), **i16)
    ops.conv_split_weights3(ctx, dT, w2, f2h, f2l, d2h, d2l)
    xp, rp, gp = split(ctx, x), split(ctx, res), split(ctx, dy)
    out = {}
    monkeypatch.setenv("PP_CONV3_DMA2_MIN", "1")      # (the default asks for a full round of tiles: this shape has 200)
    monkeypatch.setenv("PP_CONV3_TILE", "2,2")
    try:
        for dma in ("0", "1"):
            monkeypatch.setenv("PP_CONV3_DMA2", dma)
            yp = nan_planes(res)
            ops.conv_fwd3(ctx, d, None, fh, fl, bias, None, True, None, x_planes=xp, y_planes=yp, res_planes=rp)
            dxp = nan_planes(res)
            ops.conv_bwd_data3(ctx, dT, None, d2h, d2l, None, None, None, dy_planes=gp, dx_planes=dxp, addend_planes=rp, relu_src_hi=rp[0])
            torch.cuda.synchronize()
            out[dma] = (raw(yp), raw(dxp), merged(yp))
    finally:
        monkeypatch.delenv("PP_CONV3_DMA2", raising=False)
    for i in (0, 1):
        (ah, al), (bh, bl) = out["0"][i], out["1"][i]
        assert torch.equal(ah, bh) and torch.equal(al, bl), i
    assert not torch.isnan(out["1"][2]).any() and float(out["1"][2].abs().max()) > 0


@pytest.mark.parametrize("case", ["res5c_2c", "res4a_1_s2", "res3_2a_dgrad", "small_split"])
def test_persistent_1x1_gemm_matches_register_staged(ctx, monkeypatch, case):
    """Round 4, igemm4p_kernel (csrc/conv4.hip): the 1x1 convolutions as a persistent LDS-DMA GEMM (ring of 3 / 4 stages, items =
    tiles x reduction splits walked by one workgroup per CU).  Unsplit: same products in the same order as igemm3f -> bit-identical
    planes (PP_CONV4P=0 / 1, read per launch); with the reduction split: f32 summation order.  Cases: branch2c with residual + ReLU
    (2 400 rows, 512 -> 2 048: more tiles than CUs -> several items per workgroup), a stride-2 projection, a data gradient with
    addend + ReLU mask, and a launch small enough to be split."""
    from pyrapose_amd import ops
    rng = np.random.default_rng(43)
    cfg = {"res5c_2c": (8, [(15, 20)], 512, 2048, 1, "fwd"), "res4a_1_s2": (2, [(60, 80)], 512, 1024, 2, "fwd"),
           "res3_2a_dgrad": (2, [(60, 80)], 512, 128, 1, "dgrad"), "small_split": (1, [(15, 20)], 2048, 512, 1, "fwd")}[case]
    B, shapes, cin, cout, stride, kind = cfg
    out_shapes = [(-(-h // stride), -(-w // stride)) for h, w in shapes]
    rows_in, rows = sum(B * h * w for h, w in shapes), sum(B * h * w for h, w in out_shapes)
    d = ops.make_conv_desc(B, shapes, out_shapes, cin, cout, 1, stride, 0, 0, cin, cout, cout)
    x = torch.as_tensor(rng.standard_normal((rows_in, cin)), dtype=torch.float32).cuda()
    w = torch.as_tensor(rng.standard_normal((cin, cout)) * 0.05, dtype=torch.float32).cuda()
    bias = torch.as_tensor(rng.standard_normal((cout,)), dtype=torch.float32).cuda()
    res = torch.as_tensor(rng.standard_normal((rows, cout)), dtype=torch.float32).cuda()
    dy = torch.as_tensor(rng.standard_normal((rows, cout)), dtype=torch.float32).cuda()
    addend = torch.as_tensor(rng.standard_normal((rows_in, cin)), dtype=torch.float32).cuda()
    i16 = dict(dtype=torch.int16, device="cuda")
    fh, fl = torch.zeros((1, cout, cin), **i16), torch.zeros((1, cout, cin), **i16)
    dh, dl = torch.zeros((1, cin, cout), **i16), torch.zeros((1, cin, cout), **i16)
    ops.conv_split_weights3(ctx, d, w, fh, fl, dh, dl)
    xp, rp, gp, ap = split(ctx, x), split(ctx, res), split(ctx, dy), split(ctx, addend)
    out = {}
    ctx.set_workspace(64 << 20)
    try:
        for variant in ("0", "1", "3", "s"):
            monkeypatch.setenv("PP_CONV4P", "0" if variant == "0" else "1")
            monkeypatch.setenv("PP_CONV4P_NST", "3" if variant == "3" else "4")
            if variant == "s":
                monkeypatch.delenv("PP_CONV4P_SPLITS", raising=False)   # the launcher's own choice (splits the small case)
            else:
                monkeypatch.setenv("PP_CONV4P_SPLITS", "1")
                monkeypatch.setenv("PP_CONV3_SPLITS", "1")
            if kind == "fwd":
                yp = nan_planes(res)
                ops.conv_fwd3(ctx, d, None, fh, fl, bias, None, True, None, x_planes=xp, y_planes=yp, res_planes=rp)
            else:
                yp = nan_planes(addend)
                ops.conv_bwd_data3(ctx, d, None, dh, dl, None, None, None, dy_planes=gp, dx_planes=yp, addend_planes=ap, relu_src_hi=xp[0])
            torch.cuda.synchronize()
            out[variant] = (raw(yp), merged(yp))
    finally:
        ctx.set_workspace(0)
        for k_ in ("PP_CONV4P", "PP_CONV4P_NST", "PP_CONV4P_SPLITS", "PP_CONV3_SPLITS"):
            monkeypatch.delenv(k_, raising=False)
    (ah, al), va = out["0"]
    assert not torch.isnan(va).any() and float(va.abs().max()) > 0
    for variant in ("1", "3"):
        (bh, bl), _ = out[variant]
        assert torch.equal(ah, bh) and torch.equal(al, bl), variant
    vs = out["s"][1]
    assert not torch.isnan(vs).any()
    assert float((va - vs).abs().max()) <= (8e-5 if FMT[0] == 1 else 4e-5) * float(va.abs().max())
    # float64 reference
    xv, wv = merged(xp).double(), w.double()
    if kind == "fwd":
        h0, w0 = shapes[0]
        xs = xv.reshape(B, h0, w0, cin)[:, ::stride, ::stride, :].reshape(-1, cin)
        ref = torch.relu(xs @ wv + bias.double() + merged(rp).double())
    else:
        ref = (merged(gp).double() @ wv.t() + merged(ap).double()) * (xv > 0)
    err = (va.double() - ref).abs().max() / ref.abs().max()
    assert float(err) <= (1e-4 if FMT[0] == 1 else 3e-5), float(err)


@pytest.mark.parametrize("case", ["heads3", "one_level_narrow", "cout144"])
def test_tap_row_reuse_weight_gradient_matches_wgrad3f(ctx, monkeypatch, case):
    """Round 4, wgrad3r_kernel (csrc/conv4.hip): the weight gradient of a 3x3 stride-1 'same' layer with one staged (x, dy) tile
    pair per kernel ROW -- the three taps read the x tile at pixel offsets -1 / 0 / +1, edge pixels masked out of the fragments --
    against wgrad3f (PP_WGRAD3R=0, read per launch): same products, another f32 summation order.  Dense and over a list of
    32-row blocks; bias gradient; a three-level row space (levels change inside a split), a narrow single level (two image-row
    wraps per step), a cout that leaves the second column tile mostly empty; and against float64."""
    import torch.nn.functional as F
    from pyrapose_amd import ops
    rng = np.random.default_rng(47)
    B, shapes, cin, cout = {"heads3": (2, [(32, 48), (16, 24), (8, 12)], 128, 256),
                            "one_level_narrow": (3, [(20, 16)], 256, 128),
                            "cout144": (1, [(32, 40)], 128, 144)}[case]
    k = 3
    rows = sum(B * h * w for h, w in shapes)
    ld_y = (cout + 31) // 32 * 32
    d = ops.make_conv_desc(B, shapes, shapes, cin, cout, k, 1, 1, 1, cin, ld_y, (cout + 15) // 16 * 16)
    x = torch.as_tensor(rng.standard_normal((rows, cin)), dtype=torch.float32).cuda()
    dy = torch.zeros((rows, ld_y), dtype=torch.float32, device="cuda")
    dy[:, :cout] = torch.as_tensor(rng.standard_normal((rows, cout)), dtype=torch.float32).cuda()
    # a sparse gradient: non-zero in a few runs of rows only
    dys = torch.zeros_like(dy)
    for a in rng.integers(0, rows - 40, size=6):
        n = int(rng.integers(3, 40))
        dys[a: a + n] = dy[a: a + n]
    xp, gp, sp = split(ctx, x), split(ctx, dy), split(ctx, dys)
    skip = ops.row_block_list(ctx, dys, cout)
    ld_w = (cout + 15) // 16 * 16
    out = {}
    for r in ("0", "1", "2"):  # wgrad3f / wgrad3r / wgrad3w (producer + consumer waves)
        monkeypatch.setenv("PP_WGRAD3R", r)
        dw, db = torch.zeros((k * k * cin, ld_w), device="cuda"), torch.zeros((ld_w,), device="cuda")
        ops.conv_bwd_weight3(ctx, d, None, None, dw, db, x_planes=xp, dy_planes=gp)
        dws, dbs = torch.zeros((k * k * cin, ld_w), device="cuda"), torch.zeros((ld_w,), device="cuda")
        ops.conv_bwd_weight3(ctx, d, None, None, dws, dbs, x_planes=xp, dy_planes=sp, dy_skip=skip)
        torch.cuda.synchronize()
        out[r] = (dw, db, dws, dbs)
    monkeypatch.delenv("PP_WGRAD3R", raising=False)
    tol = 3e-5 if FMT[0] == 1 else 1e-5  # (two evaluations of one arithmetic: f32 summation order)
    for r in ("1", "2"):
        for i in range(4):
            a, b_ = out["0"][i], out[r][i]
            assert float(b_.abs().max()) > 0
            assert float((a - b_).abs().max()) <= tol * float(a.abs().max()), (r, i, float((a - b_).abs().max()), float(a.abs().max()))
        assert float(out[r][0][:, cout:].abs().max()) == 0 if ld_w > cout else True
    # float64: dW[ty][tx] = sum over pixels of x[pixel + offset]^T dy[pixel], image by image and level by level
    xv, gv = merged(xp).double(), merged(gp).double()[:, :cout]
    ref = torch.zeros((k, k, cin, cout), dtype=torch.float64, device="cuda")
    r0 = 0
    for (h, w) in shapes:
        n = B * h * w
        xi = xv[r0: r0 + n].reshape(B, h, w, cin)
        gi = gv[r0: r0 + n].reshape(B, h, w, cout)
        xpad = F.pad(xi, (0, 0, 1, 1, 1, 1))
        for ty in range(3):
            for tx in range(3):
                ref[ty, tx] += torch.einsum("bhwi,bhwo->io", xpad[:, ty: ty + h, tx: tx + w, :], gi)
        r0 += n
    got = out["1"][0][:, :cout].double().reshape(k, k, cin, cout)
    err = float((got - ref).abs().max() / ref.abs().max())
    assert err <= (1e-4 if FMT[0] == 1 else 2e-5), err
    assert float((out["1"][1][:cout].double() - gv.sum(0)).abs().max()) <= 1e-4 * float(gv.sum(0).abs().max())
