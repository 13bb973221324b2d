"""Callable mirrors of the reference's custom Keras layers on the hot path (layers/_misc.py,
layers/filter_detections.py).  They act on cuda tensors and launch the HIP kernels."""
import numpy as np
import torch

from .. import backend, ops
from ..runtime import default_context
from ..utils import anchors as utils_anchors


class Anchors(object):
    """layers/_misc.py:24-93: anchors for a feature map, float32, tiled over the batch."""

    def __init__(self, size, stride, ratios=None, scales=None, *args, **kwargs):
        self.size, self.stride = size, stride
        self.ratios = utils_anchors.AnchorParameters.default.ratios if ratios is None else np.array(ratios)
        self.scales = utils_anchors.AnchorParameters.default.scales if scales is None else np.array(scales)
        self.num_anchors = len(self.ratios) * len(self.scales)
        self.anchors = utils_anchors.generate_anchors(base_size=size, ratios=self.ratios, scales=self.scales)

    def __call__(self, features):
        B, H, W = features.shape[0], features.shape[1], features.shape[2]
        a = backend.shift((H, W), self.stride, self.anchors)
        return a.unsqueeze(0).expand(B, -1, -1).contiguous()


class UpsampleLike(object):
    """layers/_misc.py:96-115."""

    def __call__(self, inputs):
        source, target = inputs
        return backend.resize_images(source, (target.shape[1], target.shape[2]), method="nearest")


class RegressBoxes3D(object):
    """layers/_misc.py:165-209."""

    def __init__(self, mean=None, std=None, *args, **kwargs):
        self.mean = np.zeros(16) if mean is None else np.asarray(mean)
        self.std = np.full(16, 0.2) if std is None else np.asarray(std)

    def __call__(self, inputs):
        anchors, regression = inputs
        return backend.box3D_transform_inv(anchors, regression, mean=self.mean, std=self.std)


class FilterDetections(object):
    """layers/filter_detections.py:121-234 (class-specific filter with NMS, per image)."""

    def __init__(self, nms=True, class_specific_filter=True, nms_threshold=0.5, score_threshold=0.05, max_detections=300, **kwargs):
        if not (nms and class_specific_filter):
            raise NotImplementedError("only nms=True, class_specific_filter=True (the layer's defaults) are built")
        self.nms_threshold, self.score_threshold, self.max_detections = nms_threshold, score_threshold, max_detections

    def __call__(self, inputs):
        boxes, boxes3D, classification = inputs[0], inputs[1], inputs[2]
        ctx = default_context()
        return list(ops.filter_detections_batch(ctx, boxes.contiguous(), boxes3D.contiguous(), classification.contiguous(),
                                                self.score_threshold, self.nms_threshold, self.max_detections))
