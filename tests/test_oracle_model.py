"""CPU: analytic known-answer tests that pin the model/loss oracle, since no Keras
reference tensors exist."""
import math

import numpy as np
import torch

from oracle import model_torch as MT
from oracle import anchors_np as OA


def test_prior_probability_bias():
    from pyrapose_amd import arch
    W = arch.init_weights(13, 0)
    assert np.allclose(W["cls_out/bias"], -4.59511985, atol=1e-6)  # -log(99), initializers.py:23-39
    assert np.all(W["reg_out/bias"] == 0)


def test_focal_all_background_known_answer():
    # every element contributes 0.75 * p^2 * (-log(1-p)) with p = 0.01, divided by max(1, 0 positives)
    B, N, C = 2, 50, 4
    y_true = torch.zeros((B, N, C + 1), dtype=torch.float64)
    y_pred = torch.full((B, N, C), 0.01, dtype=torch.float64)
    want = B * N * C * 0.75 * 0.01 ** 2 * (-math.log(0.99))
    assert abs(float(MT.focal(y_true, y_pred)) - want) < 1e-12
    # ignored anchors drop out
    y_true[0, :10, -1] = -1
    want = (B * N - 10) * C * 0.75 * 0.01 ** 2 * (-math.log(0.99))
    assert abs(float(MT.focal(y_true, y_pred)) - want) < 1e-12


def test_focal_positive_normaliser():
    y_true = torch.zeros((1, 4, 3), dtype=torch.float64)
    y_true[0, 0, 1] = 1; y_true[0, 0, 2] = 1
    y_true[0, 1, 0] = 1; y_true[0, 1, 2] = 1
    p = torch.full((1, 4, 2), 0.3, dtype=torch.float64)
    pos = 0.25 * 0.7 ** 2 * (-math.log(0.3))
    neg = 0.75 * 0.3 ** 2 * (-math.log(0.7))
    want = (2 * pos + 6 * neg) / 2.0
    assert abs(float(MT.focal(y_true, p)) - want) < 1e-12


def test_orthogonal_l1_zero_and_knee():
    rng = np.random.default_rng(0)
    t = torch.as_tensor(rng.standard_normal((1, 6, 17)))
    t[..., 16] = 1
    assert float(MT.orthogonal_l1(t, t[..., :16].clone())) == 0.0
    # smooth-L1 knee at |x| = 1/9: quadratic below, linear above; orth term vanishes for a uniform shift
    for d, want_elem in ((0.05, 0.5 * 9 * 0.05 ** 2), (0.5, 0.5 - 0.5 / 9)):
        pred = t[..., :16] + d
        got = float(MT.orthogonal_l1(t, pred))
        assert abs(got - 0.125 * 0.8 * 16 * want_elem) < 1e-12


def test_orth_features_vanish_for_affine_cuboid_projection():
    """Each of the 24 features is (edge - parallel edge): zero for any affine image of a cuboid with the
    corner order of preprocessing/linemod.py:78-85."""
    rng = np.random.default_rng(1)
    sx, sy, sz = 1.0, 2.0, 3.0
    box = np.array([[sx, sy, sz], [sx, sy, -sz], [sx, -sy, -sz], [sx, -sy, sz],
                    [-sx, sy, sz], [-sx, sy, -sz], [-sx, -sy, -sz], [-sx, -sy, sz]])
    A = rng.standard_normal((2, 3)); b = rng.standard_normal(2)
    pts = (box @ A.T + b).reshape(1, 16)
    f = MT._orth_features(torch.as_tensor(pts))
    assert float(f.abs().max()) < 1e-12


def test_decode_round_trip_and_first_anchor():
    a = OA.anchors_for_shape((480, 640))
    assert a.shape == (56700, 4)
    rng = np.random.default_rng(2)
    gt = rng.uniform(0, 640, size=(a.shape[0], 16))
    back = OA.box3d_transform_inv_f32(a.astype(np.float32), OA.box3d_transform(a, gt).astype(np.float32))
    np.testing.assert_allclose(back, gt, atol=5e-3)


def test_initial_cls_output_is_prior():
    from pyrapose_amd import arch
    W = arch.init_weights(5, 1)
    W["cls_out/kernel"][:] = 0
    W["mask_out/kernel"][:] = 0
    x = np.random.default_rng(0).standard_normal((1, 64, 96, 3)).astype(np.float32) * 50
    with torch.no_grad():
        out = MT.forward(W, x, 5, torch.float32)
    assert out["cls"].shape == (1, (8 * 12 + 4 * 6 + 2 * 3) * 9, 5)
    np.testing.assert_allclose(out["cls"].numpy(), 0.01, rtol=1e-5)
    np.testing.assert_allclose(out["mask"].numpy(), 0.01, rtol=1e-5)


def test_upsample_like_rule():
    src = torch.arange(23, dtype=torch.float32).view(1, 1, 1, 23).repeat(1, 1, 17, 1)
    tgt = torch.zeros((1, 1, 34, 45))
    up = MT.upsample_like(src, tgt)
    want = np.minimum(np.floor((np.arange(45, dtype=np.float32) + 0.5) * (np.float32(23) / np.float32(45))), 22)
    assert np.array_equal(up[0, 0, 0].numpy(), want)
    # x2 reduces to dst >> 1
    src = torch.arange(20, dtype=torch.float32).view(1, 1, 1, 20)
    assert np.array_equal(MT.upsample_like(src, torch.zeros((1, 1, 1, 40)))[0, 0, 0].numpy(), np.arange(40) // 2)


def test_adam_clipnorm_semantics():
    g = {"a/kernel": torch.tensor([3.0, 4.0], dtype=torch.float64)}
    W = {"a/kernel": np.zeros(2)}
    m = {"a/kernel": torch.zeros(2, dtype=torch.float64)}; v = {"a/kernel": torch.zeros(2, dtype=torch.float64)}
    new_w, norm = MT.adam_clipnorm_step(W, g, m, v, 1, lr=1e-3, clipnorm=1.0)
    assert abs(norm - 5.0) < 1e-12
    gc = np.array([0.6, 0.8])
    lr_t = 1e-3 * math.sqrt(1 - 0.999) / (1 - 0.9)
    want = -lr_t * (0.1 * gc) / (np.sqrt(0.001 * gc ** 2) + 1e-7)
    np.testing.assert_allclose(new_w["a/kernel"].numpy(), want, rtol=1e-12)


def test_pyramid_variants_shapes_and_relu_gate():
    """__create_pyramid_features (retinanet.py:134-157): five levels, N = sum(ceil(H/2^l) * ceil(W/2^l)) * A, and P7 sees
    ReLU(P6): with a P6 conv that outputs only negative values every P7 feature equals the P7 bias."""
    import numpy as np
    import torch
    from oracle import model_torch as MT
    from pyrapose_amd import arch
    C, H, W = 3, 136, 200
    Wt = arch.init_weights(C, seed=2, pyramid="p3p7")
    x = np.random.default_rng(0).standard_normal((1, H, W, 3)).astype(np.float32) * 50
    out = MT.forward(Wt, x, C, torch.float32, pyramid="p3p7")
    n = sum(-(-H // 2 ** l) * -(-W // 2 ** l) for l in (3, 4, 5, 6, 7)) * 9
    assert out["3Dbox"].shape == (1, n, 16) and out["cls"].shape == (1, n, C)
    assert out["mask"].shape == (1, -(-H // 8) * -(-W // 8), C)
    Wt2 = dict(Wt)
    Wt2["P6_con/kernel"] = np.zeros_like(Wt["P6_con/kernel"])
    Wt2["P6_con/bias"] = np.full_like(Wt["P6_con/bias"], -1.0)
    Wt2["P7_con/bias"] = np.linspace(-1, 1, 256).astype(np.float32)
    C2, C3, C4, C5 = MT.resnet50(torch.from_numpy(x).permute(0, 3, 1, 2), Wt2, torch.float32)
    feats = MT.pyramid_features(C3, C4, C5, Wt2, torch.float32)
    assert len(feats) == 5 and torch.all(feats[3] == -1.0)
    assert torch.allclose(feats[4], torch.from_numpy(Wt2["P7_con/bias"]).view(1, -1, 1, 1).expand_as(feats[4]))
    assert len(MT.pyramid_features(C3, C4, C5, Wt2, torch.float32, with_p6p7=False)) == 3
