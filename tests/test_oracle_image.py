"""CPU: analytic known answers for the OpenCV restatement of the augmentation warp / resize (oracle/image_np.py).
OpenCV itself is not installed ("parity unpinned"): these pin what can be derived from the published algorithm alone."""
import numpy as np

from oracle import image_np as IM


def _img(rng, h=37, w=53, c=3):
    return rng.integers(0, 256, size=(h, w, c)).astype(np.uint8)


def test_identity_transform_returns_the_image():
    rng = np.random.default_rng(0)
    img = _img(rng)
    I = np.array([[1.0, 0, 0], [0, 1.0, 0]])
    for border in ("replicate", "constant"):
        assert np.array_equal(IM.warp_affine_u8(img, I, "linear", border), img)
    assert np.array_equal(IM.warp_affine_u8(img[:, :, 0], I, "nearest", "constant"), img[:, :, 0])


def test_integer_translation_shifts_exactly():
    rng = np.random.default_rng(1)
    img = _img(rng)
    M = np.array([[1.0, 0, 5], [0, 1.0, -3]])   # dst(x, y) = src(x - 5, y + 3)
    out = IM.warp_affine_u8(img, M, "linear", "constant", cval=7)
    assert np.array_equal(out[:-3, 5:], img[3:, :-5])
    assert np.all(out[:, :4] == 7) and np.all(out[-2:, :] == 7)       # fully outside -> the border value
    rep = IM.warp_affine_u8(img, M, "linear", "replicate")
    assert np.array_equal(rep[:-3, 5:], img[3:, :-5]) and np.array_equal(rep[0, :5], np.repeat(img[3:4, 0], 5, axis=0))
    msk = IM.warp_affine_u8(img[:, :, 0], M, "nearest", "constant")
    assert np.array_equal(msk[:-3, 5:], img[3:, :-5, 0]) and not msk[:, :5].any()


def test_half_pixel_shift_blends_neighbours():
    img = np.zeros((4, 6, 1), np.uint8)
    img[:, 2] = 100
    img[:, 3] = 201
    out = IM.warp_affine_u8(img, np.array([[1.0, 0, 0.5], [0, 1.0, 0]]), "linear", "replicate")
    # dst(3) = (src(2) + src(3)) / 2 = 150.5 -> 151 with the +2^14 rounding; dst(2) = (0 + 100) / 2
    assert out[1, 3, 0] == 151 and out[1, 2, 0] == 50 and out[1, 4, 0] == 101


def test_scale_two_nearest_replicates_pixels():
    rng = np.random.default_rng(2)
    m = rng.integers(0, 5, size=(8, 10)).astype(np.uint8)
    out = IM.warp_affine_u8(m, np.array([[2.0, 0, 0], [0, 2.0, 0]]), "nearest", "constant")
    # source position = dst / 2 rounded half up at 1/1024 resolution: (x * 512 + 512) >> 10
    xs = (np.arange(10) * 512 + 512) >> 10
    ys = (np.arange(8) * 512 + 512) >> 10
    assert np.array_equal(out, m[ys][:, xs])


def test_resize_scale_and_identity():
    assert IM.compute_resize_scale((480, 640, 3)) == 1.0
    assert IM.compute_resize_scale((540, 720, 3), 540, 720) == 1.0
    assert abs(IM.compute_resize_scale((960, 1280, 3)) - 0.5) < 1e-15
    assert abs(IM.compute_resize_scale((400, 1200, 3)) - 640 / 1200) < 1e-15     # the max side limits
    rng = np.random.default_rng(3)
    img = _img(rng)
    assert np.array_equal(IM.resize_linear_u8(img, 1.0), img)
    # x0.5 on a constant image stays constant; on a 2x2 block pattern it gives the block means
    assert np.all(IM.resize_linear_u8(np.full((8, 8, 3), 93, np.uint8), 0.5) == 93)
    blk = np.kron(np.array([[10, 50], [90, 130]], np.uint8), np.ones((2, 2), np.uint8))
    assert np.array_equal(IM.resize_linear_u8(blk, 0.5), np.array([[10, 50], [90, 130]], np.uint8))
    up = IM.resize_linear_u8(np.array([[0, 100]], np.uint8), 2.0)   # positions -0.25, 0.25, 0.75, 1.25 -> 0, 25, 75, 100
    assert up.shape == (2, 4) and list(up[0]) == [0, 25, 75, 100]
