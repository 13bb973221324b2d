"""Host-side mirror of the reference's ``PyraPose/utils/anchors.py`` API (same names, argument
meaning and return types) with the arithmetic done by the HIP kernels:

  AnchorParameters (:26-53)  generate_anchors (:447-478)  shift (:415-444)  guess_shapes (:357-369)
  anchors_for_shape (:372-412)  compute_gt_annotations (:290-318)  anchor_targets_bbox (:72-287)

numpy in / numpy out like the reference (the generator contract, preprocessing/generator.py:344-358);
``anchor_targets_bbox_device`` keeps everything on the GPU for the training loop.
"""
import numpy as np
import torch

from .. import ops
from ..runtime import default_context
from .compute_overlap import compute_overlap  # noqa: F401  (re-exported like the reference module)


class AnchorParameters:
    """utils/anchors.py:26-42."""

    def __init__(self, sizes, strides, ratios, scales):
        self.sizes, self.strides, self.ratios, self.scales = sizes, strides, ratios, scales

    def num_anchors(self):
        return len(self.ratios) * len(self.scales)


# utils/anchors.py:48-53 (keras.backend.floatx() == 'float32')
AnchorParameters.default = AnchorParameters(
    sizes=[32, 64, 128],
    strides=[8, 16, 32],
    ratios=np.array([0.5, 1, 2], np.float32),
    scales=np.array([2 ** 0, 2 ** (1.0 / 3.0), 2 ** (2.0 / 3.0)], np.float32),
)


# five-level variant for __create_pyramid_features (P3..P7): the RetinaNet sizes / strides the reference inherits from
# keras-retinanet (its own default above is cut down to the three levels of the sparse pyramid)
AnchorParameters.p3p7 = AnchorParameters(
    sizes=[32, 64, 128, 256, 512],
    strides=[8, 16, 32, 64, 128],
    ratios=np.array([0.5, 1, 2], np.float32),
    scales=np.array([2 ** 0, 2 ** (1.0 / 3.0), 2 ** (2.0 / 3.0)], np.float32),
)
# the commented-out dataset presets of utils/anchors.py:55-69 (4 scales -> 12 anchors per cell)
AnchorParameters.ycbv = AnchorParameters(
    sizes=[48, 96, 192], strides=[8, 16, 32], ratios=np.array([0.5, 1, 2], np.float32),
    scales=np.array([2 ** 0, 2 ** (1.0 / 3.0), 2 ** (2.0 / 3.0), 2 ** 1], np.float32))
AnchorParameters.homebrewed = AnchorParameters(
    sizes=[24, 64, 160], strides=[8, 16, 32], ratios=np.array([0.5, 1, 2], np.float32),
    scales=np.array([2 ** 0, 2 ** (1.0 / 3.0), 2 ** (2.0 / 3.0), 2 ** 1], np.float32))


def generate_anchors(base_size=16, ratios=None, scales=None):
    if ratios is None:
        ratios = AnchorParameters.default.ratios
    if scales is None:
        scales = AnchorParameters.default.scales
    return ops.generate_base_anchors(base_size, ratios, scales)


def guess_shapes(image_shape, pyramid_levels):
    image_shape = np.array(image_shape[:2])
    return [(image_shape + 2 ** x - 1) // (2 ** x) for x in pyramid_levels]


def shift(shape, stride, anchors):
    base = np.ascontiguousarray(anchors, np.float64)[None]
    out = ops.anchors_shift(default_context(), [(int(shape[0]), int(shape[1]))], [int(stride)], base, torch.float64)
    return out.cpu().numpy()


def anchors_for_shape_device(image_shape, pyramid_levels=None, anchor_params=None, shapes_callback=None, dtype=torch.float64):
    if pyramid_levels is None:
        pyramid_levels = [3, 4, 5]
    if anchor_params is None:
        anchor_params = AnchorParameters.default
    if shapes_callback is None:
        shapes_callback = guess_shapes
    shapes = [(int(s[0]), int(s[1])) for s in shapes_callback(image_shape, pyramid_levels)]
    if not shapes:
        return torch.zeros((0, 4), dtype=dtype, device="cuda")
    base = np.stack([generate_anchors(anchor_params.sizes[i], anchor_params.ratios, anchor_params.scales)
                     for i in range(len(pyramid_levels))])
    return ops.anchors_shift(default_context(), shapes, list(anchor_params.strides[: len(shapes)]), base, dtype)


def anchors_for_shape(image_shape, pyramid_levels=None, anchor_params=None, shapes_callback=None):
    return anchors_for_shape_device(image_shape, pyramid_levels, anchor_params, shapes_callback).cpu().numpy()


def compute_gt_annotations(anchors, annotations, negative_overlap=0.4, positive_overlap=0.5):
    ctx = default_context()
    a = torch.from_numpy(np.ascontiguousarray(anchors, np.float64)).cuda()
    g = torch.from_numpy(np.ascontiguousarray(np.asarray(annotations, np.float64)[:, :4])).cuda()
    argmax, state = ops.compute_gt_annotations(ctx, a, g, negative_overlap, positive_overlap)
    state = state.cpu().numpy()
    return state == 1, state == -1, argmax.cpu().numpy().astype(np.int64)


def pack_annotations(annotations_group):
    """Packs the reference's per-image annotation dicts (preprocessing/linemod.py:225) into flat arrays;
    evaluates the pose -> 16 projected corner pixels step (anchors.py:207-215) through the C ABI."""
    offs, boxes, labels, box3d, mids = [0], [], [], [], []
    for ann in annotations_group:
        for key in ("bboxes", "labels", "poses", "segmentations"):
            assert key in ann, "Annotations should contain %s." % key
        k = ann["bboxes"].shape[0]
        offs.append(offs[-1] + k)
        for i in range(k):
            boxes.append(np.asarray(ann["bboxes"][i], np.float64)[:4])
            labels.append(int(ann["labels"][i]))
            mids.append(int(ann["mask_ids"][i]))
            box3d.append(ops.project_box3d(ann["poses"][i], ann["segmentations"][i], ann["cam_params"][i]))
    G = offs[-1]
    return (offs, np.asarray(boxes, np.float64).reshape(G, 4), np.asarray(labels, np.int32),
            np.asarray(box3d, np.float64).reshape(G, 16), np.asarray(mids, np.int32))


def anchor_targets_bbox_device(anchors_dev, image_group, annotations_group, num_classes, negative_overlap=0.4,
                               positive_overlap=0.5, mask_transforms=None):
    """Device-resident variant: anchors_dev is a cuda float64 [N,4] tensor; returns cuda float32 tensors.
    mask_transforms: one 2x3 / 3x3 matrix per image -- the id masks are warped on the device first (apply_transform2mask,
    utils/image.py:219-230: cv2.warpAffine INTER_NEAREST, BORDER_CONSTANT 0; all masks must then have the same size)."""
    assert len(image_group) == len(annotations_group), "The length of the images and annotations need to be equal."
    assert len(annotations_group) > 0, "No data received to compute anchor targets for."
    ctx = default_context()
    offs, boxes, labels, box3d, mids = pack_annotations(annotations_group)
    image_hw = [(int(im.shape[0]), int(im.shape[1])) for im in image_group]
    mh, mw = (int(v) for v in guess_shapes(image_group[0].shape[:2], [3])[0])
    dev = lambda a: torch.from_numpy(a).cuda() if a.size else None
    masks, mask_hw = None, None
    if offs[-1] > 0:
        ms = [np.asarray(ann["mask"][0]).astype(np.uint8) for ann in annotations_group]
        mask_hw = [m.shape[:2] for m in ms]
        ph, pw = max(h for h, _ in mask_hw), max(w for _, w in mask_hw)
        plane = np.zeros((len(ms), ph, pw), np.uint8)   # each id mask sits in the top-left corner of its plane
        for i, m in enumerate(ms):
            plane[i, : m.shape[0], : m.shape[1]] = m
        masks = torch.from_numpy(plane).cuda()
        if mask_transforms is not None:
            # the reference (utils/image.py:216-230 apply_transform2mask + adjust_transform_for_mask) first resizes the mask
            # to the image's size and re-centres the matrix on the MASK's own shape; warping with the image's matrix is the
            # same operation only when mask and image have the same size -- refuse anything else instead of mis-aligning
            assert all(hw == mask_hw[0] for hw in mask_hw), "device mask augmentation needs equally sized masks"
            assert all(tuple(mhw) == tuple(ihw) for mhw, ihw in zip(mask_hw, image_hw)), \
                "device mask augmentation needs id masks of the image's size (got masks %s for images %s): resize them on the " \
                "host as apply_transform2mask does, or pass mask_transforms=None" % (mask_hw, image_hw)
            masks = ops.warp_affine_u8(ctx, masks, mask_transforms, "nearest", "constant", 0)
    return ops.anchor_targets(ctx, anchors_dev, offs, dev(boxes), dev(labels), dev(box3d), dev(mids), masks, mask_hw, image_hw,
                              num_classes, mh, mw, negative_overlap, positive_overlap)


def anchor_targets_bbox(anchors, image_group, annotations_group, num_classes, negative_overlap=0.4, positive_overlap=0.5):
    """Same contract as the reference (anchors.py:72-287): returns (regression_3D, labels_batch, mask_batch) numpy float32."""
    a = torch.from_numpy(np.ascontiguousarray(anchors, np.float64)).cuda()
    reg, lab, msk = anchor_targets_bbox_device(a, image_group, annotations_group, num_classes, negative_overlap, positive_overlap)
    return reg.cpu().numpy(), lab.cpu().numpy(), msk.cpu().numpy()
