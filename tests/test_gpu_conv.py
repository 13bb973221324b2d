"""GPU: the implicit-GEMM MFMA convolution family (fwd / bwd-data / bwd-weight) through the C ABI
against a plain PyTorch-CPU float64 reference of the same op.  Tolerance: 2e-5 relative to the
output's max magnitude (the kernels are exact-f32 fma chains; the north star allows 1e-3)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

RTOL = 2e-5


@pytest.fixture(scope="module")
def ctx():
    from pyrapose_amd import ops
    return ops.Context(0)


def tf_same(n, k, s):
    out = -(-n // s)
    tot = max((out - 1) * s + k - n, 0)
    return tot // 2, tot - tot // 2, out


def ref_conv(x_list, w_hwio, bias, stride, pad):
    """x_list: per level NHWC float64 tensors.  Returns per-level NHWC outputs."""
    w = torch.as_tensor(w_hwio, dtype=torch.float64).permute(3, 2, 0, 1)
    outs = []
    for x in x_list:
        xt = x.permute(0, 3, 1, 2)
        k = w.shape[2]
        if pad == "same":
            pt, pb, _ = tf_same(xt.shape[2], k, stride)
            pl, pr, _ = tf_same(xt.shape[3], k, stride)
        else:
            pt = pb = pl = pr = pad
        y = F.conv2d(F.pad(xt, (pl, pr, pt, pb)), w, None if bias is None else torch.as_tensor(bias, dtype=torch.float64), stride=stride)
        outs.append(y.permute(0, 2, 3, 1))
    return outs


def rel_err(got, want):
    want = np.asarray(want, np.float64)
    return float(np.abs(np.asarray(got, np.float64) - want).max() / max(np.abs(want).max(), 1e-30))


CASES = [
    # name, B, shapes, cin, cout, k, stride, pad, ld_y
    ("head3x3_multilevel", 2, [(12, 16), (6, 8), (3, 4)], 256, 512, 3, 1, "same", None),
    ("head_out_cout117", 2, [(12, 16), (6, 8), (3, 4)], 256, 117, 3, 1, "same", 128),
    ("mask_out_cout13", 2, [(12, 16)], 256, 13, 3, 1, "same", 16),
    ("lat1x1", 2, [(7, 9)], 512, 256, 1, 1, "same", None),
    ("down3x3s2_even", 2, [(12, 16)], 256, 256, 3, 2, "same", None),
    ("down3x3s2_odd", 1, [(17, 23)], 256, 256, 3, 2, "same", None),
    ("bneck1x1s2", 2, [(12, 16)], 256, 128, 1, 2, 0, None),
    ("bneck1x1s2_odd", 1, [(9, 13)], 128, 64, 1, 2, 0, None),
    ("bneck3x3_zp1_c64", 2, [(9, 11)], 64, 64, 3, 1, 1, None),
    ("bneck1x1_c64_c256", 3, [(10, 6)], 64, 256, 1, 1, 0, None),
    ("ragged_rows", 1, [(5, 7)], 128, 192, 3, 1, "same", None),
    # tap-row reuse (igemm3x) edges: fewer rows than one tile; image boundaries every 25 rows inside overlapping tiles;
    # rows wider than a tile (the +-1 neighbours of a row live in the next tile); many tiny levels
    ("x_tiny", 1, [(3, 4)], 64, 64, 3, 1, "same", None),
    ("x_many_images", 7, [(5, 5)], 64, 96, 3, 1, "same", None),
    ("x_wide_rows", 1, [(3, 150)], 32, 64, 3, 1, "same", None),
    ("x_five_levels", 2, [(9, 12), (5, 6), (3, 3), (2, 2), (1, 1)], 64, 128, 3, 1, "same", None),
]


def _setup(case, seed=0):
    name, B, shapes, cin, cout, k, stride, pad, ld_y = case
    rng = np.random.default_rng(seed)
    xs = [torch.as_tensor(rng.standard_normal((B, h, w, cin)), dtype=torch.float64) for h, w in shapes]
    w = rng.standard_normal((k, k, cin, cout)) / np.sqrt(k * k * cin)
    bias = rng.standard_normal((cout,))
    return name, B, shapes, cin, cout, k, stride, pad, ld_y, xs, w, bias


def _device_weight(w, cout):
    from pyrapose_amd.engine import _ru
    k2 = w.reshape(-1, cout)
    ld_w = _ru(cout, 16)
    buf = np.zeros((k2.shape[0], ld_w), np.float32)
    buf[:, :cout] = k2
    return torch.from_numpy(buf).cuda(), ld_w


def _cat_rows(ts, ld=None):
    m = torch.cat([t.reshape(-1, t.shape[-1]) for t in ts], dim=0).to(torch.float32)
    if ld is not None and ld != m.shape[1]:
        p = torch.zeros((m.shape[0], ld), dtype=torch.float32)
        p[:, : m.shape[1]] = m
        m = p
    return m.contiguous().cuda()


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_conv_fwd_bwd(ctx, case):
    from pyrapose_amd import ops
    name, B, shapes, cin, cout, k, stride, pad, ld_y, xs, w, bias = _setup(case)
    ref = ref_conv(xs, w, bias, stride, pad)
    out_shapes = [(r.shape[1], r.shape[2]) for r in ref]
    if pad == "same":
        pt, pl = tf_same(shapes[0][0], k, stride)[0], tf_same(shapes[0][1], k, stride)[0]
    else:
        pt = pl = pad
    ld_y = ld_y or ((cout + 15) // 16 * 16)
    wd, ld_w = _device_weight(w, cout)
    x = _cat_rows(xs)
    d = ops.make_conv_desc(B, shapes, out_shapes, cin, cout, k, stride, pt, pl, cin, ld_y, ld_w)
    rows_out = sum(B * h * ww for h, ww in out_shapes)
    # ---- forward: bias + residual + relu
    rng = np.random.default_rng(1)
    res = [torch.as_tensor(rng.standard_normal(tuple(r.shape)), dtype=torch.float64) for r in ref]
    y = torch.full((rows_out, ld_y), float("nan"), dtype=torch.float32, device="cuda")
    bd = torch.zeros((ld_w,), dtype=torch.float32)
    bd[:cout] = torch.as_tensor(bias, dtype=torch.float32)
    ops.conv_fwd(ctx, d, x, wd, bd.cuda(), _cat_rows(res, ld_y), True, y)
    want = torch.cat([torch.relu(r + q).reshape(-1, cout) for r, q in zip(ref, res)], dim=0).numpy()
    got = y.cpu().numpy()[:, :cout]
    assert np.isfinite(got).all()
    assert rel_err(got, want) < RTOL
    # plain conv, no epilogue
    y2 = torch.zeros_like(y)
    ops.conv_fwd(ctx, d, x, wd, None, None, False, y2)
    want2 = torch.cat([(r - torch.as_tensor(bias)).reshape(-1, cout) for r in ref], dim=0).numpy()
    assert rel_err(y2.cpu().numpy()[:, :cout], want2) < RTOL

    if cin % 64 != 0:
        return
    # ---- backward: reference by autograd in float64
    xg = [t.clone().requires_grad_(True) for t in xs]
    wt = torch.as_tensor(w, dtype=torch.float64).requires_grad_(True)
    outs = []
    for t in xg:
        xt = t.permute(0, 3, 1, 2)
        if pad == "same":
            a, b_, _ = tf_same(xt.shape[2], k, stride)
            c_, e_, _ = tf_same(xt.shape[3], k, stride)
        else:
            a = b_ = c_ = e_ = pad
        outs.append(F.conv2d(F.pad(xt, (c_, e_, a, b_)), wt.permute(3, 2, 0, 1), None, stride=stride).permute(0, 2, 3, 1))
    gys = [torch.as_tensor(rng.standard_normal(tuple(o.shape)), dtype=torch.float64) for o in outs]
    loss = sum((o * g).sum() for o, g in zip(outs, gys))
    grads = torch.autograd.grad(loss, xg + [wt])
    gx_ref, gw_ref = grads[:-1], grads[-1]
    gy = _cat_rows(gys, ld_y)
    # bwd-data with addend and relu mask
    addend = [torch.as_tensor(rng.standard_normal(tuple(t.shape)), dtype=torch.float64) for t in xs]
    rsrc = [torch.as_tensor(rng.standard_normal(tuple(t.shape)), dtype=torch.float64) for t in xs]
    dx = torch.full((x.shape[0], cin), float("nan"), dtype=torch.float32, device="cuda")
    ops.conv_bwd_data(ctx, d, gy, wd, _cat_rows(addend), _cat_rows(rsrc), dx)
    want = torch.cat([((g + a) * (r > 0)).reshape(-1, cin) for g, a, r in zip(gx_ref, addend, rsrc)], dim=0).numpy()
    assert rel_err(dx.cpu().numpy(), want) < RTOL
    dx2 = torch.zeros_like(dx)
    ops.conv_bwd_data(ctx, d, gy, wd, None, None, dx2)
    want = torch.cat([g.reshape(-1, cin) for g in gx_ref], dim=0).numpy()
    assert rel_err(dx2.cpu().numpy(), want) < RTOL
    # bwd-weight (+ bias gradient)
    dw = torch.zeros((k * k * cin, ld_w), dtype=torch.float32, device="cuda")
    db = torch.zeros((ld_w,), dtype=torch.float32, device="cuda")
    ops.conv_bwd_weight(ctx, d, x, gy, dw, db)
    assert rel_err(dw.cpu().numpy()[:, :cout], gw_ref.reshape(-1, cout).numpy()) < 5e-5
    assert np.all(dw.cpu().numpy()[:, cout:] == 0)
    db_ref = sum(g.reshape(-1, cout).sum(0) for g in gys).numpy()
    assert rel_err(db.cpu().numpy()[:cout], db_ref) < 5e-5
    # operands from pre-split planes
    xd, gd = _cat_rows(xs), _cat_rows(gys, ld_y)
    xh, xl = ops.new_planes(xd.shape[0], xd.shape[1])
    gh, gl = ops.new_planes(gd.shape[0], gd.shape[1])
    ops.split_planes3(ctx, xd, xh, xl)
    ops.split_planes3(ctx, gd, gh, gl)
    dw2 = torch.zeros_like(dw); db2 = torch.zeros_like(db)
    ops.conv_bwd_weight3(ctx, d, None, None, dw2, db2, x_planes=(xh, xl), dy_planes=(gh, gl))
    assert rel_err(dw2.cpu().numpy()[:, :cout], gw_ref.reshape(-1, cout).numpy()) < 1e-4
    assert rel_err(db2.cpu().numpy()[:cout], db_ref) < 5e-5


def test_stem_7x7_rgb(ctx):
    """conv1: ZeroPadding2D(3) + 7x7/2 on the packed RGB(+0) input (cin == 4 path)."""
    from pyrapose_amd import ops
    rng = np.random.default_rng(3)
    B, H, W = 2, 38, 50
    x = torch.as_tensor(rng.standard_normal((B, H, W, 3)), dtype=torch.float64)
    w = rng.standard_normal((7, 7, 3, 64)) / 12.0
    ref = ref_conv([x], w, None, 2, 3)[0]
    oh, ow = ref.shape[1], ref.shape[2]
    w4 = np.concatenate([w, np.zeros((7, 7, 1, 64))], axis=2).reshape(-1, 64)
    buf = np.zeros(((w4.shape[0] + 15) // 16 * 16, 64), np.float32)
    buf[: w4.shape[0]] = w4
    x3 = x.to(torch.float32).contiguous().cuda()
    x4 = torch.empty((B * H * W, 4), dtype=torch.float32, device="cuda")
    ops.pack_rgb_to_4(ctx, x3, x4)
    d = ops.make_conv_desc(B, [(H, W)], [(oh, ow)], 4, 64, 7, 2, 3, 3, 4, 64, 64)
    y = torch.empty((B * oh * ow, 64), dtype=torch.float32, device="cuda")
    ops.conv_fwd(ctx, d, x4, torch.from_numpy(buf).cuda(), None, None, False, y)
    assert rel_err(y.cpu().numpy(), ref.reshape(-1, 64).numpy()) < RTOL


def test_bad_arguments_raise(ctx):
    from pyrapose_amd import ops
    d = ops.make_conv_desc(1, [(4, 4)], [(4, 4)], 24, 32, 3, 1, 1, 1, 24, 32, 32)   # cin not a multiple of 16
    t = torch.zeros((16, 32), dtype=torch.float32, device="cuda")
    with pytest.raises(ValueError):
        ops.conv_fwd(ctx, d, t, t, None, None, False, t)


def test_pointwise_ops(ctx):
    from pyrapose_amd import ops
    rng = np.random.default_rng(5)
    B, C = 2, 64
    # maxpool 3x3/2 'same' on even and odd extents
    for (h, w) in ((12, 16), (13, 15)):
        x = torch.as_tensor(rng.standard_normal((B, h, w, C)), dtype=torch.float32)
        oh, ow = (h + 1) // 2, (w + 1) // 2
        pt, pb, _ = tf_same(h, 3, 2)
        pl, pr, _ = tf_same(w, 3, 2)
        ref = F.max_pool2d(F.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb), value=float("-inf")), 3, 2).permute(0, 2, 3, 1)
        y = torch.empty((B * oh * ow, C), dtype=torch.float32, device="cuda")
        ops.maxpool3x3s2(ctx, B, h, w, C, x.cuda(), oh, ow, y)
        assert torch.equal(y.cpu().view(B, oh, ow, C), ref)
    # nearest upsample + add, x2 and the non-integer 17x23 -> 34x45 case (TF half-pixel rule)
    from oracle import model_torch as MT
    for (sh, sw, th, tw) in ((3, 4, 6, 8), (17, 23, 34, 45)):
        src = torch.as_tensor(rng.standard_normal((B, sh, sw, C)), dtype=torch.float32)
        oth = torch.as_tensor(rng.standard_normal((B, th, tw, C)), dtype=torch.float32)
        up = MT.upsample_like(src.permute(0, 3, 1, 2), oth.permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
        out = torch.empty((B * th * tw, C), dtype=torch.float32, device="cuda")
        ops.upsample_add_fwd(ctx, B, sh, sw, th, tw, C, src.cuda(), oth.cuda(), out)
        assert torch.equal(out.cpu().view(B, th, tw, C), up + oth)
        # backward = adjoint of the gather
        g = torch.as_tensor(rng.standard_normal((B, th, tw, C)), dtype=torch.float64)
        s64 = src.double().requires_grad_(True)
        (MT.upsample_like(s64.permute(0, 3, 1, 2), oth.permute(0, 3, 1, 2)).permute(0, 2, 3, 1) * g).sum().backward()
        gs = torch.empty((B * sh * sw, C), dtype=torch.float32, device="cuda")
        ops.upsample_add_bwd(ctx, B, sh, sw, th, tw, C, g.float().cuda(), None, gs)
        assert rel_err(gs.cpu().numpy(), s64.grad.reshape(-1, C).numpy()) < 1e-6
    a = torch.as_tensor(rng.standard_normal((1000, 8)), dtype=torch.float32)
    out = torch.empty_like(a).cuda()
    ops.add_n(ctx, a.cuda(), a.cuda(), a.cuda(), out)
    assert torch.equal(out.cpu(), (a + a) + a)


BF3_CASES = [c for c in CASES if c[3] % 32 == 0 and c[4] % 32 == 0]


@pytest.fixture(scope="module", params=[0, 1], ids=["bf16x3", "f16c8"])
def pctx(ctx, request):
    """the context once per plane format / arithmetic of the *_bf16x3 entry points (csrc/planes_fmt.h): bf16 pairs (three bf16
    MFMAs per product), and P16 (f16 MFMA + block-scaled e5m2 cross terms)"""
    return ctx.twin(request.param)


@pytest.mark.parametrize("case", BF3_CASES, ids=[c[0] for c in BF3_CASES])
def test_conv_bf16x3_fwd_bwd_data(pctx, case):
    """3 x bf16 MFMA path: float32-class accuracy (<= 1e-4 of the output magnitude, ~2^-16 per product)."""
    ctx = pctx
    from pyrapose_amd import ops
    name, B, shapes, cin, cout, k, stride, pad, ld_y, xs, w, bias = _setup(case, seed=3)
    ref = ref_conv(xs, w, bias, stride, pad)
    out_shapes = [(r.shape[1], r.shape[2]) for r in ref]
    if pad == "same":
        pt, pl = tf_same(shapes[0][0], k, stride)[0], tf_same(shapes[0][1], k, stride)[0]
    else:
        pt = pl = pad
    ld_y = ld_y or ((cout + 15) // 16 * 16)
    wd, ld_w = _device_weight(w, cout)
    x = _cat_rows(xs)
    d = ops.make_conv_desc(B, shapes, out_shapes, cin, cout, k, stride, pt, pl, cin, ld_y, ld_w)
    taps = k * k
    u16 = dict(dtype=torch.int16, device="cuda")
    fh, fl = torch.zeros((taps, cout, cin), **u16), torch.zeros((taps, cout, cin), **u16)
    dh, dl = torch.zeros((taps, cin, cout), **u16), torch.zeros((taps, cin, cout), **u16)
    ops.conv_split_weights3(ctx, d, wd, fh, fl, dh, dl)
    # the planes reproduce the weights to 2^-15 (half + e5m2 remainder)
    back = ops.weight_planes_to_f32(fh, fl, ctx.planes_fmt).permute(0, 2, 1).reshape(taps * cin, cout).cpu().numpy()
    assert rel_err(back, w.reshape(-1, cout)) < 3.1e-5
    back = ops.weight_planes_to_f32(dh, dl, ctx.planes_fmt).reshape(taps * cin, cout).cpu().numpy()
    assert rel_err(back, w.reshape(-1, cout)) < 3.1e-5
    rows_out = sum(B * h * ww for h, ww in out_shapes)
    rng = np.random.default_rng(1)
    res = [torch.as_tensor(rng.standard_normal(tuple(r.shape)), dtype=torch.float64) for r in ref]
    y = torch.full((rows_out, ld_y), float("nan"), dtype=torch.float32, device="cuda")
    bd = torch.zeros((ld_w,), dtype=torch.float32)
    bd[:cout] = torch.as_tensor(bias, dtype=torch.float32)
    ops.conv_fwd3(ctx, d, x, fh, fl, bd.cuda(), _cat_rows(res, ld_y), True, y)
    want = torch.cat([torch.relu(r + q).reshape(-1, cout) for r, q in zip(ref, res)], dim=0).numpy()
    e = rel_err(y.cpu().numpy()[:, :cout], want)
    assert e < 1e-4, e
    # same conv with the gathered operand pre-split into (hi, lo) planes: same products; the launch may take a different
    # kernel (tap order), so equal up to f32 summation order
    xh, xl = ops.new_planes(x.shape[0], x.shape[1])
    ops.split_planes3(ctx, x, xh, xl)
    assert rel_err(ops.planes_to_f32((xh, xl), ctx.planes_fmt).cpu().numpy(), x.cpu().numpy()) < 3.1e-5
    y2 = torch.full_like(y, float("nan"))
    yh, yl = ops.new_planes(y.shape[0], y.shape[1])
    ops.conv_fwd3(ctx, d, None, fh, fl, bd.cuda(), _cat_rows(res, ld_y), True, y2, x_planes=(xh, xl), y_planes=(yh, yl))
    assert rel_err(y2[:, :cout].cpu().numpy(), y[:, :cout].cpu().numpy()) < 2e-6
    # split capture: the launch also emits the bf16 split of its f32 operand (from the loop's registers where the tap-row
    # reuse kernel runs, by a separate pass elsewhere); the result itself does not change
    ch, cl = ops.new_planes(xh.shape[0], xh.shape[1] * xh.shape[2], fill=0x7fc0)
    y3 = torch.full_like(y, float("nan"))
    ops.conv_fwd3(ctx, d, x, fh, fl, bd.cuda(), _cat_rows(res, ld_y), True, y3, x_capture=(ch, cl))
    assert torch.equal(torch.nan_to_num(y3), torch.nan_to_num(y))
    assert torch.equal(ch, xh) and torch.equal(cl, xl)
    y3.fill_(float("nan"))
    ops.conv_fwd3(ctx, d, x, fh, fl, bd.cuda(), _cat_rows(res, ld_y), True, y3)  # the capture is one-shot
    assert torch.equal(torch.nan_to_num(y3), torch.nan_to_num(y))
    # ... and the epilogue's pre-split copy of the output is exactly what the split kernel makes of it
    wh, wl = ops.new_planes(y2.shape[0], y2.shape[1])
    ops.split_planes3(ctx, torch.nan_to_num(y2), wh, wl)
    flat = lambda t: t.reshape(t.shape[0], -1)
    assert torch.equal(flat(yh)[:, :cout], flat(wh)[:, :cout]) and torch.equal(flat(yl)[:, :cout], flat(wl)[:, :cout])
    # bwd-data against float64 autograd
    xg = [t.clone().requires_grad_(True) for t in xs]
    wt = torch.as_tensor(w, dtype=torch.float64)
    outs = []
    for t in xg:
        xt = t.permute(0, 3, 1, 2)
        if pad == "same":
            a, b_, _ = tf_same(xt.shape[2], k, stride)
            c_, e_, _ = tf_same(xt.shape[3], k, stride)
        else:
            a = b_ = c_ = e_ = pad
        outs.append(F.conv2d(F.pad(xt, (c_, e_, a, b_)), wt.permute(3, 2, 0, 1), None, stride=stride).permute(0, 2, 3, 1))
    gys = [torch.as_tensor(rng.standard_normal(tuple(o.shape)), dtype=torch.float64) for o in outs]
    gx_ref = torch.autograd.grad(sum((o * g).sum() for o, g in zip(outs, gys)), xg)
    gy = _cat_rows(gys, ld_y)
    addend = [torch.as_tensor(rng.standard_normal(tuple(t.shape)), dtype=torch.float64) for t in xs]
    rsrc = [torch.as_tensor(rng.standard_normal(tuple(t.shape)), dtype=torch.float64) for t in xs]
    dx = torch.full((x.shape[0], cin), float("nan"), dtype=torch.float32, device="cuda")
    ops.conv_bwd_data3(ctx, d, gy, dh, dl, _cat_rows(addend), _cat_rows(rsrc), dx)
    want = torch.cat([((g + a) * (r > 0)).reshape(-1, cin) for g, a, r in zip(gx_ref, addend, rsrc)], dim=0).numpy()
    e = rel_err(dx.cpu().numpy(), want)
    assert e < 1e-4, e
    gh, gl = ops.new_planes(gy.shape[0], gy.shape[1])
    ops.split_planes3(ctx, gy, gh, gl)
    cred = (cout + 31) // 32 * 32
    ch, cl = ops.new_planes(gh.shape[0], gh.shape[1] * gh.shape[2], fill=0x7fc0)
    dx3 = torch.full_like(dx, float("nan"))
    ops.conv_bwd_data3(ctx, d, gy, dh, dl, _cat_rows(addend), _cat_rows(rsrc), dx3, dy_capture=(ch, cl))
    assert torch.equal(dx3, dx)
    assert torch.equal(flat(ch)[:, :cred], flat(gh)[:, :cred]) and torch.equal(flat(cl)[:, :cred], flat(gl)[:, :cred])
    dx2 = torch.full_like(dx, float("nan"))
    xh2, xl2 = ops.new_planes(dx.shape[0], dx.shape[1])
    ops.conv_bwd_data3(ctx, d, None, dh, dl, _cat_rows(addend), _cat_rows(rsrc), dx2, dy_planes=(gh, gl), dx_planes=(xh2, xl2))
    assert rel_err(dx2.cpu().numpy(), dx.cpu().numpy()) < 2e-6
    wh, wl = ops.new_planes(dx2.shape[0], dx2.shape[1])
    ops.split_planes3(ctx, dx2, wh, wl)
    assert torch.equal(xh2, wh) and torch.equal(xl2, wl)


WG3_CASES = [c for c in CASES if c[3] % 64 == 0]


@pytest.mark.parametrize("case", WG3_CASES, ids=[c[0] for c in WG3_CASES])
def test_conv_bf16x3_bwd_weight(pctx, case):
    """Weight gradient on the bf16 matrix cores (transposed LDS reads) vs float64 autograd."""
    ctx = pctx
    from pyrapose_amd import ops
    name, B, shapes, cin, cout, k, stride, pad, ld_y, xs, w, bias = _setup(case, seed=5)
    xg = [t.clone() for t in xs]
    wt = torch.as_tensor(w, dtype=torch.float64).requires_grad_(True)
    outs = []
    for t in xg:
        xt = t.permute(0, 3, 1, 2)
        if pad == "same":
            a, b_, _ = tf_same(xt.shape[2], k, stride)
            c_, e_, _ = tf_same(xt.shape[3], k, stride)
        else:
            a = b_ = c_ = e_ = pad
        outs.append(F.conv2d(F.pad(xt, (c_, e_, a, b_)), wt.permute(3, 2, 0, 1), None, stride=stride).permute(0, 2, 3, 1))
    out_shapes = [(o.shape[1], o.shape[2]) for o in outs]
    rng = np.random.default_rng(2)
    gys = [torch.as_tensor(rng.standard_normal(tuple(o.shape)), dtype=torch.float64) for o in outs]
    gw_ref, = torch.autograd.grad(sum((o * g).sum() for o, g in zip(outs, gys)), [wt])
    if pad == "same":
        pt, pl = tf_same(shapes[0][0], k, stride)[0], tf_same(shapes[0][1], k, stride)[0]
    else:
        pt = pl = pad
    ld_y = ld_y or ((cout + 15) // 16 * 16)
    ld_w = (cout + 15) // 16 * 16
    d = ops.make_conv_desc(B, shapes, out_shapes, cin, cout, k, stride, pt, pl, cin, ld_y, ld_w)
    dw = torch.zeros((k * k * cin, ld_w), dtype=torch.float32, device="cuda")
    db = torch.zeros((ld_w,), dtype=torch.float32, device="cuda")
    ops.conv_bwd_weight3(ctx, d, _cat_rows(xs), _cat_rows(gys, ld_y), dw, db)
    e = rel_err(dw.cpu().numpy()[:, :cout], gw_ref.reshape(-1, cout).numpy())
    assert e < 1e-4, e
    assert np.all(dw.cpu().numpy()[:, cout:] == 0)
    db_ref = sum(g.reshape(-1, cout).sum(0) for g in gys).numpy()
    assert rel_err(db.cpu().numpy()[:cout], db_ref) < 5e-5
    # operands from pre-split planes
    xd, gd = _cat_rows(xs), _cat_rows(gys, ld_y)
    xh, xl = ops.new_planes(xd.shape[0], xd.shape[1])
    gh, gl = ops.new_planes(gd.shape[0], gd.shape[1])
    ops.split_planes3(ctx, xd, xh, xl)
    ops.split_planes3(ctx, gd, gh, gl)
    dw2 = torch.zeros_like(dw); db2 = torch.zeros_like(db)
    ops.conv_bwd_weight3(ctx, d, None, None, dw2, db2, x_planes=(xh, xl), dy_planes=(gh, gl))
    assert rel_err(dw2.cpu().numpy()[:, :cout], gw_ref.reshape(-1, cout).numpy()) < 1e-4
    assert rel_err(db2.cpu().numpy()[:cout], db_ref) < 5e-5


def test_split_weights_batch_matches_per_tensor(pctx):
    """pp_conv_split_weights_bf16x3_batch: one launch over several tensors == the per-tensor launches, bit for bit."""
    ctx = pctx
    from pyrapose_amd import ops
    rng = np.random.default_rng(5)
    jobs, want = [], []
    i16 = dict(dtype=torch.int16, device="cuda")
    for (cin, cout, k, need_dg) in [(64, 144, 3, True), (256, 13, 3, True), (128, 512, 1, False), (32, 32, 1, True)]:
        ld_w = (cout + 15) // 16 * 16
        d = ops.make_conv_desc(1, [(8, 8)], [(8, 8)], cin, cout, k, 1, k // 2, k // 2, cin, (cout + 31) // 32 * 32, ld_w)
        w = torch.zeros((k * k * cin, ld_w), dtype=torch.float32, device="cuda")
        w[:, :cout] = torch.as_tensor(rng.standard_normal((k * k * cin, cout)), dtype=torch.float32).cuda()
        mk = lambda shape: (torch.full(shape, -1, **i16), torch.full(shape, -1, **i16))
        dg_shape = (k * k, cin, (cout + 31) // 32 * 32)
        a = mk((k * k, cout, cin)) + (mk(dg_shape) if need_dg else (None, None))
        b = mk((k * k, cout, cin)) + (mk(dg_shape) if need_dg else (None, None))
        ops.conv_split_weights3(ctx, d, w, *a)
        jobs.append((d, w) + b)
        want.append((a, b))
    batch = ops.SplitWeightsBatch(jobs)
    batch.run(ctx)
    torch.cuda.synchronize()
    for a, b in want:
        for ta, tb in zip(a, b):
            assert (ta is None) == (tb is None)
            if ta is not None:
                assert torch.equal(ta, tb)


@pytest.mark.parametrize("splits", [2, 5])
def test_conv_bf16x3_split_k(pctx, splits, monkeypatch):
    """Split-K (pp_ctx_set_workspace): same result as the single-pass launch up to f32 summation order, both directions,
    fused bias / residual / mask / ReLU applied by the finishing pass; deterministic (two runs are bit-identical)."""
    ctx = pctx
    from pyrapose_amd import ops
    rng = np.random.default_rng(11)
    B, H, W, cin, cout, k = 2, 9, 13, 128, 80, 3
    ld_y = 96
    d = ops.make_conv_desc(B, [(H, W)], [(H, W)], cin, cout, k, 1, 1, 1, cin, ld_y, 80)
    rows = B * H * W
    x = torch.as_tensor(rng.standard_normal((rows, cin)), dtype=torch.float32).cuda()
    w = torch.as_tensor(rng.standard_normal((k * k * cin, 80)) * 0.05, dtype=torch.float32).cuda()
    bias = torch.as_tensor(rng.standard_normal((80,)), dtype=torch.float32).cuda()
    res = torch.as_tensor(rng.standard_normal((rows, ld_y)), dtype=torch.float32).cuda()
    i16 = dict(dtype=torch.int16, device="cuda")
    fh, fl = torch.zeros((k * k, cout, cin), **i16), torch.zeros((k * k, cout, cin), **i16)
    dh, dl = torch.zeros((k * k, cin, ld_y), **i16), torch.zeros((k * k, cin, ld_y), **i16)
    ops.conv_split_weights3(ctx, d, w, fh, fl, dh, dl)
    gy = torch.zeros((rows, ld_y), dtype=torch.float32, device="cuda")
    gy[:, :cout] = torch.as_tensor(rng.standard_normal((rows, cout)), dtype=torch.float32).cuda()
    add = torch.as_tensor(rng.standard_normal((rows, cin)), dtype=torch.float32).cuda()
    msk = torch.as_tensor(rng.standard_normal((rows, cin)), dtype=torch.float32).cuda()

    def run():
        y = torch.full((rows, ld_y), float("nan"), dtype=torch.float32, device="cuda")
        dx = torch.full((rows, cin), float("nan"), dtype=torch.float32, device="cuda")
        ops.conv_fwd3(ctx, d, x, fh, fl, bias, res, True, y)
        ops.conv_bwd_data3(ctx, d, gy, dh, dl, add, msk, dx)
        torch.cuda.synchronize()
        return y[:, :cout].clone(), dx.clone()

    monkeypatch.setenv("PP_CONV3_SPLITS", "1")
    y1, dx1 = run()
    ctx.set_workspace(splits * rows * max(ld_y, cin) * 4)
    try:
        monkeypatch.setenv("PP_CONV3_SPLITS", str(splits))
        ctx.workspace.fill_(float("nan"))  # any contents: every word a launch reads it has written first
        y2, dx2 = run()
        y3, dx3 = run()
        assert torch.equal(y2, y3) and torch.equal(dx2, dx3)
        # split capture under split-K: every (kernel row, channel chunk) group belongs to exactly one split
        xh, xl = ops.new_planes(x.shape[0], x.shape[1])
        ops.split_planes3(ctx, x, xh, xl)
        ch, cl = ops.new_planes(xh.shape[0], xh.shape[1] * xh.shape[2], fill=0x7fc0)
        y4 = torch.full((rows, ld_y), float("nan"), dtype=torch.float32, device="cuda")
        ops.conv_fwd3(ctx, d, x, fh, fl, bias, res, True, y4, x_capture=(ch, cl))
        assert torch.equal(y4[:, :cout], y2) and torch.equal(ch, xh) and torch.equal(cl, xl)
    finally:
        ctx.set_workspace(0)
    assert rel_err(y2.cpu().numpy(), y1.cpu().numpy()) < 2e-6
    assert rel_err(dx2.cpu().numpy(), dx1.cpu().numpy()) < 2e-6
    assert not torch.equal(y1, torch.zeros_like(y1))


@pytest.mark.parametrize("frac", [0.0, 0.03, 1.0])
def test_row_block_skip_matches_dense(pctx, frac):
    """Sparse gradients (3D-box head: non-zero only around positive anchors): bwd-weight over the listed 32-row blocks of dy
    and bwd-data with tile skipping give the dense results -- a block of zero rows adds exactly 0.0."""
    ctx = pctx
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
    nb = (rows + 31) // 32
    want = np.array([bool(live[32 * b: 32 * b + 32].any()) for b in range(nb)])
    assert np.array_equal(flags.cpu().numpy()[:nb].astype(bool), want)
    bl = blocks.cpu().numpy()
    assert bl[0] == want.sum() and np.array_equal(bl[1: 1 + bl[0]], np.nonzero(want)[0])
    add = torch.as_tensor(rng.standard_normal((rows, cin)), dtype=torch.float32).cuda()
    msk = torch.as_tensor(rng.standard_normal((rows, cin)), dtype=torch.float32).cuda()
    dx0, dx1 = torch.full((rows, cin), float("nan"), device="cuda"), torch.full((rows, cin), float("nan"), device="cuda")
    ops.conv_bwd_data3(ctx, d, dy, dh, dl, add, msk, dx0)
    ops.conv_bwd_data3(ctx, d, dy, dh, dl, add, msk, dx1, dy_skip=(flags, blocks))
    assert rel_err(dx1.cpu().numpy(), dx0.cpu().numpy()) < 2e-6  # (another kernel: equal up to f32 summation order)
    dead = ~flags[nb: 2 * nb].bool().cpu().numpy()  # scratch half: the dx blocks no non-zero reaches
    if dead.any():
        rows_dead = np.repeat(dead, 32)[:rows]
        assert torch.equal(dx1[torch.as_tensor(rows_dead).cuda()], torch.where(msk > 0, add, torch.zeros_like(add))[torch.as_tensor(rows_dead).cuda()])
    assert frac > 0.5 or dead.any()
    dw0, dw1 = torch.zeros_like(w), torch.zeros_like(w)
    db0, db1 = torch.zeros((cout,), device="cuda"), torch.zeros((cout,), device="cuda")
    ops.conv_bwd_weight3(ctx, d, x, dy, dw0, db0)
    ops.conv_bwd_weight3(ctx, d, x, dy, dw1, db1, dy_skip=(flags, blocks))
    scale = max(float(dw0.abs().max()), 1e-30)
    assert float((dw0 - dw1).abs().max()) <= 2e-6 * scale and float((db0 - db1).abs().max()) <= 2e-6 * max(float(db0.abs().max()), 1e-30)
    if frac == 0.0:
        assert not dw1.any() and not db1.any() and torch.equal(dx1, torch.where(msk > 0, add, torch.zeros_like(add)))
    # the hint is one-shot: the next call is dense again
    dw2 = torch.zeros_like(w)
    ops.conv_bwd_weight3(ctx, d, x, dy, dw2, None)
    assert float((dw0 - dw2).abs().max()) <= 2e-6 * scale
