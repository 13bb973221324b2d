"""GPU: the experimental "f16c8" convolution kernel (csrc/conv2.hip: f16 main term + block-scaled e5m2 cross terms, H16L8
storage) through the C ABI against a PyTorch-CPU float64 convolution.  Not wired into the engine;
the test pins what the go / no-go measurement relied on: the kernel computes the scheme it claims, to its error level
(~3e-5; the bar is today's per-kernel 1e-4), on the tap-row-reuse edge cases of tests/test_gpu_conv.py."""
import numpy as np
import pytest
import torch

from test_gpu_conv import _cat_rows, _device_weight, ref_conv, rel_err

pytestmark = pytest.mark.gpu

CASES = [
    # name, B, shapes, cin, cout
    ("head_multilevel", 2, [(12, 16), (6, 8), (3, 4)], 256, 512),
    ("head_out_cout144", 2, [(12, 16), (6, 8), (3, 4)], 512, 144),
    ("mask_out_cout13", 2, [(12, 16)], 256, 13),
    ("bneck_c64", 2, [(9, 11)], 64, 64),
    ("tiny", 1, [(3, 4)], 64, 64),
    ("many_images", 7, [(5, 5)], 64, 96),
    ("wide_rows", 1, [(3, 150)], 64, 64),
    ("five_levels", 2, [(9, 12), (5, 6), (3, 3), (2, 2), (1, 1)], 64, 128),
]


@pytest.fixture(scope="module")
def ctx():
    from pyrapose_amd import ops
    return ops.Context(0)


def test_h16l8_round_trip_and_layout(ctx):
    from pyrapose_amd import ops
    rng = np.random.default_rng(0)
    x = torch.as_tensor(rng.standard_normal((37, 128)) * np.exp(rng.standard_normal((37, 128))), dtype=torch.float32).cuda()
    x[0, :4] = torch.tensor([0.0, 1.0, -65504.0, 1e6])            # exact values, the largest half, saturation
    hl = ops.split_hl(ctx, x, ops.new_hl(37, 128))
    back = ops.merge_hl(ctx, hl, torch.empty_like(x)).cpu().numpy()
    xs = x.cpu().numpy().copy()
    xs[0, 3] = 65504.0                                               # values beyond the half range saturate
    err = np.abs(back - xs) / np.maximum(np.abs(xs), 1e-30)
    assert err[xs != 0].max() < 2.0 ** -13 and back[0, 0] == 0.0 and back[0, 1] == 1.0 and back[0, 2] == -65504.0
    # the documented byte layout: group g of row r at (r * ld / 64 + g) * 192; 64 f16 hi values, then the lo bytes of the
    # octets 0, 2, 4, 6 followed by those of the octets 1, 3, 5, 7
    raw = hl.cpu().numpy()
    hi = raw[:, :, :128].copy().view(np.float16).reshape(37, 128)
    assert np.array_equal(hi, np.clip(xs, -65504, 65504).astype(np.float16))
    lo = raw[5, 1, 128:].astype(np.uint16) << 8
    lo = lo.view(np.float16).astype(np.float64) / 4096
    order = [8 * o + j for o in (0, 2, 4, 6, 1, 3, 5, 7) for j in range(8)]
    rem = (xs[5, 64:128].astype(np.float64) - hi[5, 64:128].astype(np.float64))[order]
    assert np.all(np.abs(lo - rem) <= np.abs(rem) * 0.126 + 1e-12)   # e5m2: 2 mantissa bits, round to nearest


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_conv_fwd_f16c8(ctx, case):
    from pyrapose_amd import ops
    name, B, shapes, cin, cout = case
    rng = np.random.default_rng(3)
    xs = [torch.as_tensor(np.maximum(rng.standard_normal((B, h, w, cin)), 0), dtype=torch.float64) for h, w in shapes]  # post-ReLU
    w = rng.standard_normal((3, 3, cin, cout)) / np.sqrt(9 * cin)
    bias = rng.standard_normal((cout,))
    ref = ref_conv(xs, w, bias, 1, "same")
    ld_y = (cout + 15) // 16 * 16
    wd, ld_w = _device_weight(w, cout)
    x = _cat_rows(xs)
    d = ops.make_conv_desc(B, shapes, shapes, cin, cout, 3, 1, 1, 1, cin, ld_y, ld_w)
    x_hl = ops.split_hl(ctx, x, ops.new_hl(x.shape[0], cin))
    w_hl = torch.zeros((9, cout, cin // 64, 192), dtype=torch.uint8, device="cuda")
    ops.conv_split_weights2(ctx, d, wd, w_hl, None)
    bd = torch.zeros((ld_w,), dtype=torch.float32)
    bd[:cout] = torch.as_tensor(bias, dtype=torch.float32)
    y = torch.full((x.shape[0], ld_y), float("nan"), dtype=torch.float32, device="cuda")
    ops.conv_fwd2(ctx, d, x_hl, w_hl, bd.cuda(), True, y)
    want = torch.cat([torch.relu(r).reshape(-1, cout) for r in ref], dim=0).numpy()
    got = y.cpu().numpy()[:, :cout]
    assert np.isfinite(got).all()
    assert rel_err(got, want) < 1e-4
    y2 = torch.full_like(y, float("nan"))
    ops.conv_fwd2(ctx, d, x_hl, w_hl, None, False, y2)
    want2 = torch.cat([(r - torch.as_tensor(bias)).reshape(-1, cout) for r in ref], dim=0).numpy()
    assert rel_err(y2.cpu().numpy()[:, :cout], want2) < 1e-4


def test_f16c8_refuses_what_the_prototype_does_not_cover(ctx):
    from pyrapose_amd import ops
    d = ops.make_conv_desc(1, [(8, 8)], [(8, 8)], 64, 64, 1, 1, 0, 0, 64, 64, 64)       # 1x1: not a 3-wide 'same' conv
    x_hl, w_hl = ops.new_hl(64, 64), torch.zeros((1, 64, 1, 192), dtype=torch.uint8, device="cuda")
    with pytest.raises((ValueError, RuntimeError)):
        ops.conv_fwd2(ctx, d, x_hl, w_hl, None, False, torch.zeros((64, 64), device="cuda"))
