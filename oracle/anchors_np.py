"""numpy restatement of the reference's anchor generation, IoU and target assignment.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PINNED: bit-exact against
tests/golden/anchors_targets.npz (generated from the reference's own code).

Follows /root/reference/PyraPose/utils/anchors.py and utils/compute_overlap.pyx;
line numbers are cited per function.  Vectorised, own code.
"""
import numpy as np


# ----------------------------------------------------------------------------- T1
def default_anchor_parameters():
    """anchors.py:48-53 -- ratios/scales are float32-rounded (keras floatx)."""
    return dict(
        sizes=[32, 64, 128],
        strides=[8, 16, 32],
        ratios=np.array([0.5, 1, 2], np.float32),
        scales=np.array([2 ** 0, 2 ** (1.0 / 3.0), 2 ** (2.0 / 3.0)], np.float32),
    )


def generate_anchors(base_size, ratios, scales):
    """anchors.py:447-478.  Returns float64 [R*S, 4], anchor a = ratio_idx*S + scale_idx.

    dtype walk of the reference: ``base_size * np.tile(scales, ...)`` is a float32 product
    (python int x float32 array), widened to float64 on assignment; everything after is
    float64 with the float32 ratios promoted.
    """
    ratios = np.asarray(ratios)
    scales = np.asarray(scales)
    R, S = len(ratios), len(scales)
    side = (base_size * scales).astype(np.float64)        # float32 multiply, then widen  (:465)
    side = np.tile(side, R)                               # a = r*S + s
    rr = np.repeat(ratios, S)
    areas = side * side                                   # :468
    w = np.sqrt(areas / rr)                               # :471
    h = w * rr                                            # :472
    out = np.zeros((R * S, 4))
    out[:, 0] = 0.0 - w * 0.5                             # :475
    out[:, 2] = w - w * 0.5
    out[:, 1] = 0.0 - h * 0.5                             # :476
    out[:, 3] = h - h * 0.5
    return out


# ----------------------------------------------------------------------------- T2
def guess_shapes(image_shape, pyramid_levels):
    """anchors.py:357-369 -- integer ceil-div by 2**level."""
    return [((int(image_shape[0]) + 2 ** l - 1) // 2 ** l, (int(image_shape[1]) + 2 ** l - 1) // 2 ** l)
            for l in pyramid_levels]


def shift(shape, stride, anchors):
    """anchors.py:415-444 -- cell-major (y, x), anchor-minor."""
    sx = (np.arange(0, shape[1]) + 0.5) * stride
    sy = (np.arange(0, shape[0]) + 0.5) * stride
    gx, gy = np.meshgrid(sx, sy)
    shifts = np.stack([gx.ravel(), gy.ravel(), gx.ravel(), gy.ravel()], axis=1)   # [K,4]
    return (shifts[:, None, :] + anchors[None, :, :]).reshape(-1, 4)


def anchors_for_shape(image_shape, pyramid_levels=None, params=None):
    """anchors.py:372-412."""
    if pyramid_levels is None:
        pyramid_levels = [3, 4, 5]
    if params is None:
        params = default_anchor_parameters()
    shapes = guess_shapes(image_shape, pyramid_levels)
    parts = []
    for i, _ in enumerate(pyramid_levels):
        base = generate_anchors(params["sizes"][i], params["ratios"], params["scales"])
        parts.append(shift(shapes[i], params["strides"][i], base))
    return np.concatenate(parts, axis=0) if parts else np.zeros((0, 4))


def anchors_for_shape_f32(image_shape, pyramid_levels=None, params=None):
    """Device-side float32 anchors: layers/_misc.py:60-71 -> backend/common.py:93-116.

    Base anchors are cast to floatx (float32) in ``Anchors.__init__`` (keras.backend.variable),
    the shifts ``(arange + 0.5) * stride`` are float32, and the broadcast add is float32.
    """
    if pyramid_levels is None:
        pyramid_levels = [3, 4, 5]
    if params is None:
        params = default_anchor_parameters()
    shapes = guess_shapes(image_shape, pyramid_levels)
    parts = []
    for i, _ in enumerate(pyramid_levels):
        base = generate_anchors(params["sizes"][i], params["ratios"], params["scales"]).astype(np.float32)
        st = np.float32(params["strides"][i])
        sx = (np.arange(0, shapes[i][1], dtype=np.float32) + np.float32(0.5)) * st
        sy = (np.arange(0, shapes[i][0], dtype=np.float32) + np.float32(0.5)) * st
        gx, gy = np.meshgrid(sx, sy)
        shifts = np.stack([gx.ravel(), gy.ravel(), gx.ravel(), gy.ravel()], axis=1)
        parts.append((base[None, :, :] + shifts[:, None, :]).reshape(-1, 4))
    return np.concatenate(parts, axis=0)


# ----------------------------------------------------------------------------- T3
def compute_overlap(boxes, query):
    """compute_overlap.pyx:13-53 -- float64 IoU with the '+1' pixel convention.

    An entry stays 0 unless iw > 0 and ih > 0.  Raises ValueError on wrong dtype/ndim like the
    Cython buffer-typed signature does.
    """
    boxes = np.asarray(boxes)
    query = np.asarray(query)
    if boxes.dtype != np.float64 or query.dtype != np.float64:
        raise ValueError("Buffer dtype mismatch, expected 'double'")
    if boxes.ndim != 2 or query.ndim != 2:
        raise ValueError("Buffer has wrong number of dimensions (expected 2)")
    N, K = boxes.shape[0], query.shape[0]
    if N == 0 or K == 0:
        return np.zeros((N, K))
    bx1, by1, bx2, by2 = (boxes[:, i][:, None] for i in range(4))
    qx1, qy1, qx2, qy2 = (query[:, i][None, :] for i in range(4))
    q_area = (qx2 - qx1 + 1) * (qy2 - qy1 + 1)
    iw = np.minimum(bx2, qx2) - np.maximum(bx1, qx1) + 1
    ih = np.minimum(by2, qy2) - np.maximum(by1, qy1) + 1
    b_area = (bx2 - bx1 + 1) * (by2 - by1 + 1)
    ua = b_area + q_area - iw * ih
    with np.errstate(divide="ignore", invalid="ignore"):
        ov = iw * ih / ua
    return np.where((iw > 0) & (ih > 0), ov, 0.0)


# ----------------------------------------------------------------------------- T4
def compute_gt_annotations(anchors, bboxes, negative_overlap=0.4, positive_overlap=0.5):
    """anchors.py:290-318 -- first max wins (numpy argmax)."""
    ov = compute_overlap(anchors.astype(np.float64), bboxes.astype(np.float64))
    amax = np.argmax(ov, axis=1)
    mx = ov[np.arange(ov.shape[0]), amax]
    positive = mx >= positive_overlap
    ignore = (mx > negative_overlap) & ~positive
    return positive, ignore, amax


# ----------------------------------------------------------------------------- T5 helpers
def quat2mat(q):
    """transforms3d 0.3.1 ``quaternions.quat2mat`` (third-party, not in /root/reference;
    published algorithm restated; call site anchors.py:207)."""
    w, x, y, z = (float(v) for v in q)
    Nq = w * w + x * x + y * y + z * z
    if Nq < np.finfo(np.float64).eps:
        return np.eye(3)
    s = 2.0 / Nq
    X, Y, Z = x * s, y * s, z * s
    wX, wY, wZ = w * X, w * Y, w * Z
    xX, xY, xZ = x * X, x * Y, x * Z
    yY, yZ, zZ = y * Y, y * Z, z * Z
    return np.array([[1.0 - (yY + zZ), xY - wZ, xZ + wY],
                     [xY + wZ, 1.0 - (xX + zZ), yZ - wX],
                     [xZ - wY, yZ + wX, 1.0 - (xX + yY)]])


def project_box3d(pose, box8x3, cam):
    """anchors.py:207-214 + toPix_array :562-567 -> 16 pixel coordinates (float64)."""
    rot = np.asarray(quat2mat(pose[3:]), dtype=np.float32)          # :208 float32 rounding
    t = rot[:3, :3].dot(np.asarray(box8x3, dtype=np.float64).T).T   # :210
    t = t + np.repeat(np.asarray(pose[:3])[np.newaxis, :], 8, axis=0)
    fx, fy, cx, cy = cam
    xp = (t[:, 0] * fx) / t[:, 2] + cx
    yp = (t[:, 1] * fy) / t[:, 2] + cy
    return np.stack((xp, yp), axis=1).reshape(16)


def box3d_transform(anchors, gt16):
    """anchors.py:515-559 -- mean 0, std 0.2; corner j uses anchor x1,y1,x2,y2 alternately."""
    aw = anchors[:, 2] - anchors[:, 0]
    ah = anchors[:, 3] - anchors[:, 1]
    out = np.empty((anchors.shape[0], 16))
    for j in range(16):
        ref = anchors[:, j % 4]
        size = aw if j % 2 == 0 else ah
        out[:, j] = ((gt16[:, j] - ref) / size - 0) / 0.2
    return out


def pil_nearest_index(n_in, n_out):
    """Index map of PIL ``Image.resize(..., NEAREST)`` along one axis (call site anchors.py:158).

    Pillow (third-party) ImagingScaleAffine: start at ``scale*0.5``, truncate, then
    *accumulate* ``+= scale`` in double precision.  Pinned by tests/golden (pil_nearest_*).
    """
    scale = float(n_in) / float(n_out)
    out = np.empty(n_out, np.int64)
    xo = scale * 0.5
    for i in range(n_out):
        out[i] = int(xo)
        xo += scale
    return np.minimum(out, n_in - 1)


# ----------------------------------------------------------------------------- T5
def anchor_targets_bbox(anchors, image_shapes, annotations_group, num_classes,
                        negative_overlap=0.4, positive_overlap=0.5):
    """anchors.py:72-287.

    ``image_shapes``: list of (h, w) of the *unpadded* images (the reference takes the image
    arrays and only uses ``.shape``).  Mask level shape comes from image 0 (anchors.py:113-115).
    Returns (regression_3D [B,N,17], labels [B,N,C+1], mask [B,M,C+1]) float32.
    """
    B, N = len(image_shapes), anchors.shape[0]
    labels = np.zeros((B, N, num_classes + 1), np.float32)
    reg = np.zeros((B, N, 17), np.float32)
    mh, mw = guess_shapes(image_shapes[0], [3])[0]
    mask_b = np.zeros((B, mh * mw, num_classes + 1), np.float32)
    for b, (ishape, ann) in enumerate(zip(image_shapes, annotations_group)):
        if ann["bboxes"].shape[0]:
            pos, ign, amax = compute_gt_annotations(anchors, ann["bboxes"], negative_overlap, positive_overlap)
            labels[b, ign, -1] = -1
            labels[b, pos, -1] = 1
            reg[b, ign, -1] = -1
            reg[b, pos, -1] = 1
            cls_of = np.asarray(ann["labels"])[amax[pos]].astype(int)          # :148
            labels[b, np.nonzero(pos)[0], cls_of] = 1
            mask_img = np.asarray(ann["mask"][0])
            rows = pil_nearest_index(mask_img.shape[0], mh)
            cols = pil_nearest_index(mask_img.shape[1], mw)
            small = mask_img[rows][:, cols].reshape(-1)
            boxes16 = np.empty((0, 16))
            for k in range(len(ann["poses"])):
                cls = int(ann["labels"][k])
                sel = np.nonzero(small == int(ann["mask_ids"][k]))[0]
                if len(sel) > 1:                                               # :162
                    mask_b[b, sel, cls] = 1
                    mask_b[b, sel, -1] = 1
                boxes16 = np.concatenate([boxes16, [project_box3d(ann["poses"][k], ann["segmentations"][k],
                                                                  ann["cam_params"][k])]], axis=0)
            reg[b, :, :-1] = box3d_transform(anchors, boxes16[amax, :])        # :267
        # anchors whose centre falls outside the (unpadded) image -> ignore   (:279-285)
        cx = (anchors[:, 0] + anchors[:, 2]) / 2
        cy = (anchors[:, 1] + anchors[:, 3]) / 2
        outside = np.logical_or(cx >= ishape[1], cy >= ishape[0])
        labels[b, outside, -1] = -1
        reg[b, outside, -1] = -1
    return reg, labels, mask_b


# ----------------------------------------------------------------------------- D2 / D3
def box3d_transform_inv_f32(anchors, deltas):
    """backend/common.py:25-56 in float32, op by op (no fused multiply-add):
    ``corner_j = anchor[j%4] + (delta_j * 0.2 + 0) * (w or h)``."""
    anchors = np.asarray(anchors, np.float32)
    deltas = np.asarray(deltas, np.float32)
    w = anchors[..., 2] - anchors[..., 0]
    h = anchors[..., 3] - anchors[..., 1]
    out = np.empty_like(deltas)
    std, mean = np.float32(0.2), np.float32(0.0)
    for j in range(16):
        size = w if j % 2 == 0 else h
        out[..., j] = anchors[..., j % 4] + (deltas[..., j] * std + mean) * size
    return out


def score_threshold_indices(scores, thr=0.5):
    """utils/linemod_eval.py:317-319 -- per class ascending anchor indices with score > thr."""
    return [np.nonzero(scores[:, c] > thr)[0] for c in range(scores.shape[1])]
