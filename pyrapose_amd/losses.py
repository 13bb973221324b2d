"""Mirror of the reference's ``PyraPose/losses.py`` for the two losses the training graph compiles
(bin/train.py:95-102): ``focal`` (losses.py:22-68) and ``orthogonal_l1`` (losses.py:321-408).

Like the reference, each factory returns a functor ``loss(y_true, y_pred)``.  The functor carries its
hyper-parameters (the engine reads them at ``compile``) and, when called directly with device tensors
in the Keras layout, evaluates the fused HIP kernel and returns the scalar loss as a 0-d device tensor.
The other loss variants of the reference file are never compiled by train.py and are out of scope.
"""
import torch

from . import ops
from ._lib import RowSpace
from .runtime import default_context


class _Loss(object):
    kind = None

    def __repr__(self):
        return "<pyrapose_amd.losses.%s %s>" % (self.kind, self.__dict__)


class _Focal(_Loss):
    kind = "focal"

    def __init__(self, alpha=0.25, gamma=2.0):
        self.alpha, self.gamma = float(alpha), float(gamma)

    def __call__(self, y_true, y_pred_logits):
        """y_true (B,N,C+1) with the anchor state last; y_pred_logits (B,N,C) PRE-sigmoid scores."""
        ctx = default_context()
        B, N, C = y_pred_logits.shape
        rs = RowSpace.make(B, [(1, N)])
        logits = y_pred_logits.contiguous().view(B * N, C)
        if C % 4:
            pad = torch.zeros((B * N, (C + 3) // 4 * 4), dtype=torch.float32, device=logits.device)
            pad[:, :C] = logits
            logits = pad
        cnt = torch.zeros(4, dtype=torch.int32, device="cuda")
        ops.count_positives(ctx, None, y_true.contiguous(), None, cnt)
        out = torch.zeros(1, dtype=torch.float32, device="cuda")
        ops.focal(ctx, rs, 1, C, logits, y_true.contiguous(), self.alpha, self.gamma, cnt[1:2], 1.0, out, None)
        return out[0]


class _OrthL1(_Loss):
    kind = "orthogonal_l1"

    def __init__(self, weight=0.125, sigma=3.0):
        self.weight, self.sigma = float(weight), float(sigma)

    def __call__(self, y_true, y_pred):
        ctx = default_context()
        B, N, _ = y_pred.shape
        rs = RowSpace.make(B, [(1, N)])
        cnt = torch.zeros(4, dtype=torch.int32, device="cuda")
        ops.count_positives(ctx, y_true.contiguous(), None, None, cnt)
        out = torch.zeros(1, dtype=torch.float32, device="cuda")
        ops.orth_l1(ctx, rs, 1, y_pred.contiguous().view(B * N, 16), y_true.contiguous(), self.weight, self.sigma, cnt[0:1], 1.0, out, None)
        return out[0]


def focal(alpha=0.25, gamma=2.0):
    return _Focal(alpha, gamma)


def orthogonal_l1(weight=0.125, sigma=3.0):
    return _OrthL1(weight, sigma)
