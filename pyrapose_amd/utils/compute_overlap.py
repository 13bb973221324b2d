"""Drop-in for the reference's Cython ``PyraPose.utils.compute_overlap.compute_overlap``
(utils/compute_overlap.pyx:13-53): same signature, float64 in / float64 out, ValueError on a wrong
dtype or ndim exactly where the Cython buffer-typed arguments raise.  The IoU matrix is computed by
the HIP kernel ``pp_compute_overlap_f64`` (bit-exact with the Cython loop)."""
import numpy as np
import torch

from .. import ops
from ..runtime import default_context


def compute_overlap(boxes, query_boxes):
    for name, a in (("boxes", boxes), ("query_boxes", query_boxes)):
        if not isinstance(a, np.ndarray):
            raise TypeError("Argument '%s' has incorrect type (expected numpy.ndarray, got %s)" % (name, type(a).__name__))
        if a.dtype != np.float64:
            raise ValueError("Buffer dtype mismatch, expected 'double' but got '%s'" % a.dtype.name)
        if a.ndim != 2:
            raise ValueError("Buffer has wrong number of dimensions (expected 2, got %d)" % a.ndim)
    n, k = boxes.shape[0], query_boxes.shape[0]
    if n == 0 or k == 0:
        return np.zeros((n, k), dtype=np.float64)
    ctx = default_context()
    b = torch.from_numpy(np.ascontiguousarray(boxes[:, :4])).cuda()
    q = torch.from_numpy(np.ascontiguousarray(query_boxes[:, :4])).cuda()
    return ops.compute_overlap(ctx, b, q).cpu().numpy()
