"""Mirror of the pose-error functions the evaluation loops call (utils/pose_error.py:64-75, 210-246;
utils/linemod_eval.py:525-531): same names, arguments and float return values, computed by the HIP kernels."""
import numpy as np
import torch

from .. import ops
from ..runtime import default_context


def transform_pts_Rt(pts, R, t):
    """pose_error.py:64-75 (host: a 3x3 product is not worth a launch)."""
    pts = np.asarray(pts)
    assert pts.shape[1] == 3
    return (np.asarray(R).dot(pts.T) + np.asarray(t).reshape((3, 1))).T


def _one(R_est, t_est, R_gt, t_gt, pts, symmetric):
    dev = lambda a, shape: torch.from_numpy(np.ascontiguousarray(np.asarray(a, np.float64).reshape(shape))).cuda()
    pts = np.asarray(pts)
    if pts.ndim != 2 or pts.shape[1] != 3:
        raise ValueError("pts must be n x 3")
    out = ops.pose_errors(default_context(), dev(pts, (-1, 3)), dev(R_est, (1, 3, 3)), dev(t_est, (1, 3)), dev(R_gt, (1, 3, 3)),
                          dev(t_gt, (1, 3)), symmetric)
    return float(out.cpu()[0])


def add(R_est, t_est, R_gt, t_gt, pts):
    """Average Distance of Model Points (pose_error.py:210-228)."""
    return _one(R_est, t_est, R_gt, t_gt, pts, False)


def adi(R_est, t_est, R_gt, t_gt, pts):
    """Average Distance to the nearest model point, for objects with indistinguishable views (pose_error.py:231-246)."""
    return _one(R_est, t_est, R_gt, t_gt, pts, True)


def add_batch(R_est, t_est, R_gt, t_gt, pts, symmetric=False):
    """n poses of one object in one launch: R_* [n,3,3], t_* [n,3] -> float64 [n]."""
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(np.asarray(a, np.float64))).cuda()
    return ops.pose_errors(default_context(), dev(pts), dev(R_est), dev(np.reshape(t_est, (-1, 3))), dev(R_gt),
                           dev(np.reshape(t_gt, (-1, 3))), symmetric).cpu().numpy()
