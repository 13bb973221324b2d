"""CPU restatement (numpy, integer arithmetic) of the OpenCV calls behind the reference's geometric augmentation --
TEST INFRASTRUCTURE (see oracle/__init__.py).

  utils/image.py:207-214  cv2.warpAffine(image, M[:2], dsize, flags=INTER_LINEAR, borderMode=..., borderValue=cval)
  utils/image.py:222-229  cv2.warpAffine(mask,  M[:2], dsize, flags=INTER_NEAREST, borderMode=BORDER_CONSTANT, borderValue=0)
  utils/image.py:281-323  compute_resize_scale, cv2.resize(img, None, fx=scale, fy=scale)

"PARITY UNPINNED": opencv-python is a third-party dependency (setup.py of the reference, unpinned) that is neither in the
reference tree nor installed here, and the reference holds no fixtures for these calls.  This file restates OpenCV 4's
published algorithm for 8-bit images (modules/imgproc/src/imgwarp.cpp: warpAffine -> remap with 1/32-pixel positions and
15-bit bilinear weights; resize.cpp: 11-bit coefficients, two passes), checked by analytic cases (tests/test_oracle_image.py:
identity, integer shifts, half-pixel blends, borders)."""
import numpy as np


def invert_affine(M):
    """cv::warpAffine without WARP_INVERSE_MAP inverts the 2x3 matrix in double"""
    m = np.asarray(M, np.float64).reshape(-1)[:6]
    D = m[0] * m[4] - m[1] * m[3]
    D = 1.0 / D if D != 0 else 0.0
    o = np.zeros(6)
    o[0], o[4] = m[4] * D, m[0] * D
    o[1], o[3] = m[1] * (-D), m[3] * (-D)
    o[2] = -o[0] * m[2] - o[1] * m[5]
    o[5] = -o[3] * m[2] - o[4] * m[5]
    return o


def _sat_i32(v):
    return np.clip(np.rint(v), -2147483648, 2147483647).astype(np.int64)


def _positions(Mi, H, W, delta, shift):
    x = np.arange(W, dtype=np.float64)[None, :]
    y = np.arange(H, dtype=np.float64)[:, None]
    X0 = _sat_i32((Mi[1] * y + Mi[2]) * 1024.0) + delta
    Y0 = _sat_i32((Mi[4] * y + Mi[5]) * 1024.0) + delta
    X = (X0 + _sat_i32(Mi[0] * x * 1024.0)) >> shift
    Y = (Y0 + _sat_i32(Mi[3] * x * 1024.0)) >> shift
    return X, Y


def _clip(v, a, b):
    return np.where(v >= a, np.where(v < b, v, b - 1), a)


def warp_affine_u8(img, M, interpolation="linear", border="replicate", cval=0):
    """img uint8 [H,W] or [H,W,C]; M the forward 2x3 matrix the reference hands to cv2.warpAffine"""
    img = np.asarray(img)
    H, W = img.shape[:2]
    S = img.reshape(H, W, -1).astype(np.int64)
    Mi = invert_affine(M)
    if interpolation == "nearest":
        X, Y = _positions(Mi, H, W, 512, 10)
        sx, sy = np.clip(X, -32768, 32767), np.clip(Y, -32768, 32767)
        inside = (sx >= 0) & (sx < W) & (sy >= 0) & (sy < H)
        v = S[_clip(sy, 0, H), _clip(sx, 0, W)]
        if border == "constant":
            v = np.where(inside[..., None], v, cval)
        return v.astype(np.uint8).reshape(img.shape)
    X, Y = _positions(Mi, H, W, 16, 5)
    sx, sy = np.clip(X >> 5, -32768, 32767), np.clip(Y >> 5, -32768, 32767)
    ax, ay = (X & 31)[..., None], (Y & 31)[..., None]
    w = [(32 - ay) * (32 - ax) * 32, (32 - ay) * ax * 32, ay * (32 - ax) * 32, ay * ax * 32]
    out = np.zeros_like(S)
    acc = np.zeros_like(S)
    for k, (oy, ox) in enumerate(((0, 0), (0, 1), (1, 0), (1, 1))):
        yy, xx = sy + oy, sx + ox
        ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
        v = S[_clip(yy, 0, H), _clip(xx, 0, W)]
        if border == "constant":
            v = np.where(ok[..., None], v, cval)
        acc = acc + v * w[k]
    out = (acc + (1 << 14)) >> 15
    if border == "constant":
        gone = (sx >= W) | (sx + 1 < 0) | (sy >= H) | (sy + 1 < 0)
        out = np.where(gone[..., None], cval, out)
    return out.astype(np.uint8).reshape(img.shape)


def compute_resize_scale(image_shape, min_side=480, max_side=640):
    """utils/image.py:281-304"""
    rows, cols = image_shape[0], image_shape[1]
    scale = min_side / min(rows, cols)
    if max(rows, cols) * scale > max_side:
        scale = max_side / max(rows, cols)
    return scale


def resize_linear_u8(img, scale):
    """cv2.resize(img, None, fx=scale, fy=scale) for uint8 (INTER_LINEAR)"""
    img = np.asarray(img)
    SH, SW = img.shape[:2]
    DH, DW = int(np.rint(SH * scale)), int(np.rint(SW * scale))
    S = img.reshape(SH, SW, -1).astype(np.int64)
    inv = 1.0 / scale

    def axis(n_dst, n_src):
        f = ((np.arange(n_dst, dtype=np.float64) + 0.5) * inv - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = (f - s.astype(np.float32)).astype(np.float32)
        lo = s < 0
        f[lo], s[lo] = 0, 0
        hi = s >= n_src - 1
        f[hi], s[hi] = 0, n_src - 1
        c0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
        c1 = np.rint(f * np.float32(2048)).astype(np.int64)
        return s, np.minimum(s + 1, n_src - 1), c0, c1
    sx, sx1, a0, a1 = axis(DW, SW)
    sy, sy1, b0, b1 = axis(DH, SH)
    r0 = S[sy][:, sx] * a0[None, :, None] + S[sy][:, sx1] * a1[None, :, None]
    r1 = S[sy1][:, sx] * a0[None, :, None] + S[sy1][:, sx1] * a1[None, :, None]
    v = (((b0[:, None, None] * (r0 >> 4)) >> 16) + ((b1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    out_shape = (DH, DW) + img.shape[2:]
    return np.clip(v, 0, 255).astype(np.uint8).reshape(out_shape)
