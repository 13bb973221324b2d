"""GPU: the two plane formats side by side (csrc/planes_fmt.h) -- the converter between them (scale, ReLU mask), the gradient scale,
and the engine's arithmetics (bf16x3 / f16c8 / mixed) against each other on the same weights and batch.  The parity of the default
(mixed) step against the float64 oracle is tests/test_gpu_model.py and tests/test_gpu_parity.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from pyrapose_amd.runtime import default_context
    return default_context()


def test_convert_planes_round_trip_scale_and_mask(ctx):
    from pyrapose_amd import ops
    g = torch.Generator(device="cuda").manual_seed(4)
    rows, ld = 777, 64
    # magnitudes over twelve binades, exact zeros, negatives
    x = torch.randn((rows, ld), device="cuda", generator=g) * torch.exp2(torch.randint(-10, 3, (rows, ld), device="cuda", generator=g).float())
    x[::7, ::3] = 0.0
    p0, p1, back = ops.new_planes(rows, ld), ops.new_planes(rows, ld), ops.new_planes(rows, ld)
    ops.split_planes3(ctx, x, p0[0], p0[1])
    v0 = ops.planes_to_f32(p0, 0)
    assert float((v0 - x).abs().max()) <= 2.0 ** -16 * float(x.abs().max())
    # 0 -> 1: the value P16 holds is within 2^-15 of the element (half + e5m2 remainder) down to the smallest normal half, within
    # 2^-28 absolute below it (the remainder's own subnormals); zeros stay exact zeros
    ops.convert_planes(ctx, p0, 0, p1, 1)
    v1 = ops.planes_to_f32(p1, 1)
    assert bool(((v1 - v0).abs() <= torch.maximum(v0.abs() * 2.0 ** -14, torch.full_like(v0, 2.0 ** -27))).all())
    assert torch.equal(v1 == 0, x == 0)
    # the same encoding as a direct split under the P16 context
    q1 = ops.new_planes(rows, ld)
    ops.split_planes3(ctx.twin(1), v0, q1[0], q1[1])
    assert torch.equal(q1[0], p1[0]) and torch.equal(q1[1], p1[1])
    # 1 -> 0 is exact (a half plus an e5m2 * 2^-12 fits a bf16 pair's 16 bits of mantissa)
    ops.convert_planes(ctx, p1, 1, back, 0)
    assert float(((ops.planes_to_f32(back, 0) - v1).abs() / v1.abs().clamp_min(1e-30)).max()) <= 2.0 ** -16
    # scale: {2^G, 2^-G}, either index; the ReLU mask of a third tensor (either format: the sign test of the hi half)
    sc = torch.tensor([2.0 ** 11, 2.0 ** -11], device="cuda")
    m = torch.randn((rows, ld), device="cuda", generator=g)
    m[::5] = 0.0
    for mfmt in (0, 1):
        mp = ops.new_planes(rows, ld)
        ops.split_planes3(ctx.twin(mfmt), m, mp[0], mp[1])
        out = ops.new_planes(rows, ld, fill=0x7fc0)
        ops.convert_planes(ctx, p1, 1, out, 0, sc, 1, mp[0])
        got = ops.planes_to_f32(out, 0)
        want = torch.where(m > 0, v1 * 2.0 ** -11, torch.zeros_like(v1))
        assert float((got - want).abs().max()) <= 2.0 ** -16 * float(want.abs().max())
        assert torch.equal(got == 0, (m <= 0) | (v1 == 0))
    out = ops.new_planes(rows, ld)
    ops.convert_planes(ctx, p0, 0, out, 1, sc, 0)
    w1, big = ops.planes_to_f32(out, 1), v0.abs() * 2048 >= 28672.0  # (what leaves the range of the format is clamped)
    assert bool((((w1 - v0 * 2048).abs() <= torch.maximum(v0.abs() * 2048 * 2.0 ** -14, torch.full_like(v0, 2.0 ** -27))) | big).all())
    assert bool((w1[big].abs() == 28672.0).all()) and int(big.sum()) > 0
    with pytest.raises(Exception):
        ops.convert_planes(ctx, p0, 0, out, 2)


def test_use_stream_on_twinned_context_rebinds_both(ctx):
    """Context.use_stream used to forward to the twin, which forwarded back (RecursionError in the default mixed mode,
    where every engine context has a twin).  Both handles must move, and launches of either format must run on the new stream."""
    from pyrapose_amd import ops
    c = ops.Context(0)
    t = c.twin(1)
    assert t.twin(0) is c
    side = torch.cuda.Stream()
    c.use_stream(side)
    assert c.stream is side and t.stream is side
    x = torch.randn((64, 32), device="cuda")
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for cc, fmt in ((c, 0), (t, 1)):
            p = ops.new_planes(64, 32)
            ops.split_planes3(cc, x, p[0], p[1])
            assert float((ops.planes_to_f32(p, fmt) - x).abs().max()) <= 2.0 ** -14 * float(x.abs().max())
    t.use_stream(torch.cuda.current_stream())  # from the twin's side too
    assert c.stream is t.stream
    torch.cuda.synchronize()
    c.close()


def test_p16_range(ctx):
    """|x| is clamped to 28672 (the remainder * 2^12 must fit e5m2); what a half cannot hold at all (< 2^-24) becomes zero -- the
    reason gradients travel scaled."""
    from pyrapose_amd import ops
    c1 = ctx.twin(1)
    x = torch.tensor([[1e5, -1e5, 28672.0, 3e-8, 1e-9, 6.1e-5, -2.5, 0.0]], device="cuda").repeat(4, 1)
    p = ops.new_planes(4, 8)
    ops.split_planes3(c1, x, p[0], p[1])
    v = ops.planes_to_f32(p, 1)[0].cpu().numpy()
    assert v[0] == 28672.0 and v[1] == -28672.0 and v[2] == 28672.0
    assert v[4] == 0.0 and v[7] == 0.0 and v[6] == -2.5
    assert abs(v[5] - 6.1e-5) <= 2.0 ** -11 * 6.1e-5  # a subnormal half + remainder


@pytest.mark.parametrize("counts,want", [([5, 900, 41], 2), ([1, 1, 1], 0), ([0, 3, 3], 0), ([64, 64, 127], 6), ([100000, 70000, 65536], 16)])
def test_grad_scale_from_counts(ctx, counts, want, monkeypatch):
    """2^G = 2^(base + floor(log2(max(1, min counts)))) and its inverse, as exact powers of two (base: 8, PP_GSCALE_LOG2)"""
    from pyrapose_amd import ops
    monkeypatch.delenv("PP_GSCALE_LOG2", raising=False)
    c = torch.tensor(counts, dtype=torch.int32, device="cuda")
    s = torch.zeros((2,), device="cuda")
    ops.grad_scale_from_counts(ctx, c, s)
    a, b = s.cpu().tolist()
    assert a == 2.0 ** (8 + want) and b == 2.0 ** -(8 + want)
    # loss weights above the reference's defaults take their factor out of the scale (Engine.gscale_adjust)
    ops.grad_scale_from_counts(ctx, c, s, log2_adjust=-3)
    assert s.cpu().tolist() == [2.0 ** (5 + want), 2.0 ** -(5 + want)]


def _run(ctx, mode, Wt, x, tg, C, B, H, W):
    from pyrapose_amd.engine import Engine
    eng = Engine(ctx, C, B, H, W, weights=Wt, train=True, conv_mode=mode)
    eng.set_targets(*tg(eng))
    eng.forward(x)
    eng.loss_and_backward()
    torch.cuda.synchronize()
    out = dict(reg=eng.reg_out.t[:, : eng.reg_out.C].clone(), cls=eng.cls_out.t[:, : eng.cls_out.C].clone(),
               mask=eng.mask_out.t[:, : eng.mask_out.C].clone(), C3=eng.C3.f32(eng.ctx).clone(), C5=eng.C5.f32(eng.ctx).clone(),
               pyr=eng.pyr.f32(eng.ctx).clone(), grad=eng.params.grad.clone(), losses=eng.losses())
    info = dict(fmts=dict({n: a.fmt for n, a in eng.acts.items()}, C3=eng.C3.fmt, C5=eng.C5.fmt, pyramid=eng.pyr.fmt, reg_out=eng.reg_out.fmt), converts=[o["x"].name for o in eng.graph_ops if o["kind"] == "convert"],
                entries=dict(eng.params.entries), arith=eng.arith, gscale=(eng.gscale.cpu().tolist() if eng.gscale is not None else None))
    eng.close()
    return out, info


def test_engine_arithmetics_agree(ctx):
    """bf16x3 everywhere / f16c8 everywhere / mixed (default): same weights, same batch.  The mixed engine keeps the backbone in
    bf16 pairs -- bit-identical to the bf16x3 engine up to C3 / C4 / C5 -- re-encodes those three tensors once, and runs FPN + heads
    on P16 planes; forward results, losses and gradients agree at the scale of the coarser arithmetic (2^-15 per stored tensor)."""
    from pyrapose_amd import arch
    from tests.test_gpu_model import random_targets, synth_input
    B, H, W, C = 2, 128, 160, 5
    rng = np.random.default_rng(11)
    Wt = arch.init_weights(C, seed=12)
    x = torch.from_numpy(synth_input(rng, B, H, W)).cuda()
    tgs = {}

    def tg(eng):
        if "t" not in tgs:
            tgs["t"] = [torch.from_numpy(a).cuda() for a in random_targets(rng, B, eng.N, eng.M3, C, pos_frac=0.02)]
        return tgs["t"]
    ref, iref = _run(ctx, "bf16x3", Wt, x, tg, C, B, H, W)
    assert iref["converts"] == [] and set(iref["fmts"].values()) == {0} and iref["gscale"] is None
    for mode in ("mixed", "f16c8"):
        got, info = _run(ctx, mode, Wt, x, tg, C, B, H, W)
        assert info["arith"] == mode
        if mode == "mixed":
            assert len(info["converts"]) == 3, info["converts"]  # C3, C4, C5 (by the names of the backbone's last blocks)
            assert info["fmts"]["C3"] == 0 and info["fmts"]["C5"] == 0 and info["fmts"]["pyramid"] == 1 and info["fmts"]["reg_out"] == 1
            assert torch.equal(got["C3"], ref["C3"]) and torch.equal(got["C5"], ref["C5"])  # the backbone forward is the same launches
        else:
            assert info["converts"] == [] and info["fmts"]["C3"] == 1
        assert info["gscale"] is not None and info["gscale"][0] >= 256.0 and info["gscale"][0] * info["gscale"][1] == 1.0
        for k in ("reg", "cls", "mask", "pyr", "C5"):
            assert float((got[k] - ref[k]).abs().max()) <= 3e-4 * float(ref[k].abs().max()), (mode, k)
        for k in ("3Dbox", "cls", "mask"):
            assert abs(got["losses"][k] - ref["losses"][k]) <= 1e-4 * abs(ref["losses"][k]), (mode, k)
        # gradients, tensor by tensor (a ReLU input that is zero to rounding may flip between two arithmetics: norm-wise bounds)
        worst = 0.0
        for name, e in info["entries"].items():
            a = got["grad"][e["offset"]: e["offset"] + e["count"]].double()
            b = ref["grad"][e["offset"]: e["offset"] + e["count"]].double()
            if float(b.norm()) == 0.0:
                assert float(a.norm()) == 0.0, name
                continue
            worst = max(worst, float((a - b).norm() / b.norm()))
        assert worst <= 2e-2, (mode, worst)
        ga, gb = got["grad"].double(), ref["grad"].double()
        assert float((ga - gb).norm() / gb.norm()) <= 5e-3, mode
