"""Device-side mirrors of the reference's ``PyraPose/backend`` wrappers that the hot path uses:
``shift`` (backend/common.py:93-116), ``box3D_transform_inv`` (backend/common.py:25-56) and
``resize_images(..., method='nearest')`` (backend/tf_backend.py:28-35).  Inputs/outputs are cuda tensors."""
import numpy as np
import torch

from .. import ops
from ..runtime import default_context


def shift(shape, stride, anchors):
    """float32 shifted anchors [K*A, 4] like the Keras Anchors layer computes them."""
    base = np.asarray(anchors.detach().cpu().numpy() if torch.is_tensor(anchors) else anchors, np.float64)[None]
    return ops.anchors_shift(default_context(), [(int(shape[0]), int(shape[1]))], [int(stride)], base, torch.float32)


def box3D_transform_inv(boxes, deltas, mean=None, std=None):
    """boxes (B,N,4) or (N,4) float32 anchors, deltas (B,N,16) -> (B,N,16).  mean 0 / std 0.2 only."""
    if mean is not None and np.any(np.asarray(mean) != 0):
        raise ValueError("box3D_transform_inv: only the reference's mean=0 is supported")
    if std is not None and np.any(np.abs(np.asarray(std) - 0.2) > 1e-12):
        raise ValueError("box3D_transform_inv: only the reference's std=0.2 is supported")
    a = boxes[0] if boxes.dim() == 3 else boxes
    return ops.box3d_decode(default_context(), a.contiguous(), deltas.contiguous())


def resize_images(images, size, method="nearest", align_corners=False):
    """images (B,H,W,C) float32 cuda; nearest only (tf.image.resize NEAREST, TF 2.1 half-pixel rule)."""
    if method != "nearest":
        raise NotImplementedError("only method='nearest' is on the hot path (layers/_misc.py:96-109)")
    B, H, W, C = images.shape
    th, tw = int(size[0]), int(size[1])
    out = torch.empty((B, th, tw, C), dtype=torch.float32, device=images.device)
    ops.upsample_add_fwd(default_context(), B, H, W, th, tw, C, images.contiguous(), None, out)
    return out
