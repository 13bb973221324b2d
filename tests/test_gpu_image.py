"""GPU: the augmentation warp / resize kernels (pp_warp_affine_u8, pp_resize_linear_u8) against the numpy restatement of
OpenCV's fixed-point scheme (oracle/image_np.py): integer arithmetic on both sides -> bit-exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from pyrapose_amd.runtime import default_context
    return default_context()


def random_transform(rng, H, W):
    """the kind of matrix utils/transform.py:random_transform_generator produces: scaling 0.9..1.1 and a translation about the
    image centre (bin/train.py:189-199), plus a small rotation / shear to exercise every matrix entry"""
    s = rng.uniform(0.8, 1.2)
    a = rng.uniform(-0.2, 0.2)
    sh = rng.uniform(-0.1, 0.1)
    A = np.array([[s * np.cos(a), -s * np.sin(a + sh)], [s * np.sin(a), s * np.cos(a + sh)]])
    c = np.array([0.5 * W, 0.5 * H])
    t = c - A @ c + rng.uniform(-0.2, 0.2, 2) * np.array([W, H])
    return np.concatenate([A, t[:, None]], axis=1)


def test_warp_affine_bit_exact_vs_oracle(ctx):
    from oracle import image_np as IM
    from pyrapose_amd import ops
    rng = np.random.default_rng(0)
    B, H, W = 5, 97, 131
    img = rng.integers(0, 256, size=(B, H, W, 3)).astype(np.uint8)
    msk = rng.integers(0, 4, size=(B, H, W)).astype(np.uint8)
    mats = [random_transform(rng, H, W) for _ in range(B)]
    mats[0] = np.array([[1.0, 0, 0], [0, 1.0, 0]])           # identity
    mats[1] = np.array([[1.0, 0, 7], [0, 1.0, -4]])          # integer shift
    d_img, d_msk = torch.from_numpy(img).cuda(), torch.from_numpy(msk).cuda()
    for border, cval in (("replicate", 0), ("constant", 0), ("constant", 17)):
        got = ops.warp_affine_u8(ctx, d_img, mats, "linear", border, cval).cpu().numpy()
        for b in range(B):
            want = IM.warp_affine_u8(img[b], mats[b], "linear", border, cval)
            assert np.array_equal(got[b], want), (border, cval, b, int(np.abs(got[b].astype(int) - want).max()))
    assert np.array_equal(ops.warp_affine_u8(ctx, d_img, mats, "linear", "replicate").cpu().numpy()[0], img[0])
    got = ops.warp_affine_u8(ctx, d_msk, mats, "nearest", "constant", 0).cpu().numpy()   # apply_transform2mask
    for b in range(B):
        assert np.array_equal(got[b], IM.warp_affine_u8(msk[b], mats[b], "nearest", "constant", 0)), b
    # 3x3 matrices (the reference carries homogeneous transforms and slices [:2]) are accepted as they are
    m33 = [np.vstack([m, [0, 0, 1]]) for m in mats]
    assert np.array_equal(ops.warp_affine_u8(ctx, d_msk, m33, "nearest", "constant", 0).cpu().numpy(), got)
    with pytest.raises(ValueError):
        ops.warp_affine_u8(ctx, d_img, mats, "nearest")      # nearest is the 1-channel mask path


def test_warp_at_640x480_batch8(ctx):
    """the bench batch shape: a quick whole-batch equality and a timing line"""
    from oracle import image_np as IM
    from pyrapose_amd import ops
    rng = np.random.default_rng(1)
    B, H, W = 8, 480, 640
    img = rng.integers(0, 256, size=(B, H, W, 3)).astype(np.uint8)
    mats = [random_transform(rng, H, W) for _ in range(B)]
    d = torch.from_numpy(img).cuda()
    out = torch.empty_like(d)
    ops.warp_affine_u8(ctx, d, mats, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.warp_affine_u8(ctx, d, mats, out=out)
    e1.record()
    torch.cuda.synchronize()
    print("warpAffine 8 x 640x480x3 uint8: %.1f us per batch" % (e0.elapsed_time(e1) * 100))
    got = out.cpu().numpy()
    for b in (0, 7):
        assert np.array_equal(got[b], IM.warp_affine_u8(img[b], mats[b]))


def test_resize_bit_exact_vs_oracle(ctx):
    from oracle import image_np as IM
    from pyrapose_amd import ops
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, size=(3, 60, 84, 3)).astype(np.uint8)
    d = torch.from_numpy(img).cuda()
    for scale in (1.0, 0.5, 2.0, 480 / 540, 640 / 84, 0.37):
        got = ops.resize_linear_u8(ctx, d, scale).cpu().numpy()
        for b in range(3):
            want = IM.resize_linear_u8(img[b], scale)
            assert got[b].shape == want.shape and np.array_equal(got[b], want), (scale, b)
    assert ops.resize_scale(480, 640) == 1.0 and ops.resize_scale(540, 720, 540, 720) == 1.0
    assert ops.resize_scale(400, 1200) == IM.compute_resize_scale((400, 1200, 3))
    g1 = ops.resize_linear_u8(ctx, torch.from_numpy(img[:, :, :, 0].copy()).cuda(), 0.5).cpu().numpy()
    assert np.array_equal(g1[1], IM.resize_linear_u8(img[1, :, :, 0], 0.5))


def test_lean_feed_with_device_augmentation(ctx):
    """Engine.train_step_from_annotations(transforms=...): identity transforms change nothing; a real transform gives the
    step that the same feed takes on the host-warped image and mask (oracle warp) -- losses equal."""
    import bench
    from oracle import image_np as IM
    from pyrapose_amd import arch
    from pyrapose_amd.engine import Engine
    B, H, W, C = 2, 96, 128, 5
    rng = np.random.default_rng(5)
    _, images, anns = bench.synth_batch(B, H, W, C, seed=3, side=(20, 50))
    u8 = rng.integers(0, 256, (B, H, W, 3)).astype(np.uint8)
    Wt = arch.init_weights(C, seed=4)
    ident = [np.eye(3) for _ in range(B)]
    mats = [random_transform(rng, H, W) for _ in range(B)]

    def step(img, ann, tf):
        eng = Engine(ctx, C, B, H, W, weights=Wt, train=True)
        eng.train_step_from_annotations(torch.from_numpy(img).cuda(), ann, transforms=tf)
        torch.cuda.synchronize()
        out = eng.losses()
        eng.close()
        return out
    base = step(u8, anns, None)
    same = step(u8, anns, ident)
    assert all(abs(base[k] - same[k]) <= 1e-6 * max(abs(base[k]), 1e-6) for k in base)
    dev = step(u8, anns, mats)
    warped = np.stack([IM.warp_affine_u8(u8[b], mats[b], "linear", "replicate") for b in range(B)])
    anns_w = []
    for b, a in enumerate(anns):
        a2 = dict(a)
        a2["mask"] = [IM.warp_affine_u8(np.asarray(a["mask"][0], np.uint8), mats[b], "nearest", "constant", 0)]
        anns_w.append(a2)
    host = step(warped, anns_w, None)
    assert all(abs(dev[k] - host[k]) <= 1e-6 * max(abs(host[k]), 1e-6) for k in host), (dev, host)
    assert abs(dev["total"] - base["total"]) > 1e-4   # (the transform did change the step)
