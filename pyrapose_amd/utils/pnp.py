"""Pose from the network's corner votes: the RANSAC-PnP step of the evaluation loops (utils/linemod_eval.py:421-431,
479-485; the same block in occlusion_eval.py / ycbv_eval.py / tless_eval.py), on the device.

`solve_pnp_ransac` takes the arguments the reference passes to cv2.solvePnPRansac and returns the same tuple
(retval, rvec, tvec, inliers); `rodrigues` is cv2.Rodrigues(rvec)[0].  The estimator is this library's own
(csrc/pnp.hip) -- OpenCV is not part of the reference tree, parity with it is statistical only."""
import numpy as np
import torch

from .. import ops
from ..runtime import default_context


def rodrigues(rvec):
    """rotation vector -> 3x3 matrix (cv2.Rodrigues(rvec)[0])"""
    w = np.asarray(rvec, np.float64).reshape(3)
    th = float(np.sqrt((w * w).sum()))
    Kx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-12:
        return np.eye(3) + Kx
    return np.eye(3) + (np.sin(th) / th) * Kx + ((1 - np.cos(th)) / (th * th)) * (Kx @ Kx)


def rotation_vector(R):
    """3x3 matrix -> rotation vector [3,1] (cv2.Rodrigues(R)[0])"""
    R = np.asarray(R, np.float64)
    c = min(1.0, max(-1.0, (np.trace(R) - 1.0) / 2.0))
    th = float(np.arccos(c))
    v = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    if th < 1e-9:
        return (0.5 * v).reshape(3, 1)
    if np.pi - th < 1e-6:  # near pi: axis from the symmetric part
        A = (R + np.eye(3)) / 2.0
        ax = np.sqrt(np.maximum(np.diag(A), 0.0))
        k = int(np.argmax(ax))
        ax = A[:, k] / max(ax[k], 1e-300)
        if float(v @ ax) < 0:
            ax = -ax
        return (th * ax / np.linalg.norm(ax)).reshape(3, 1)
    return (th / (2.0 * np.sin(th)) * v).reshape(3, 1)


def solve_pnp_batch(problems, iterations=300, reproj_error=5.0, seed=0, points_per_vote=8, ctx=None):
    """problems: list of (obj [n,3], img [n,2], K 3x3) -> list of (ok, R [3,3], t [3], inlier indices); ONE launch."""
    if not problems:
        return []
    ctx = ctx or default_context()
    offs, objs, imgs, Ks = [0], [], [], []
    for obj, img, K in problems:
        obj = np.asarray(obj, np.float64).reshape(-1, 3)
        img = np.asarray(img, np.float64).reshape(-1, 2)
        if obj.shape[0] != img.shape[0]:
            raise ValueError("objectPoints and imagePoints differ in length")
        K = np.asarray(K, np.float64).reshape(3, 3)
        objs.append(obj); imgs.append(img); Ks.append([K[0, 0], K[1, 1], K[0, 2], K[1, 2]])
        offs.append(offs[-1] + obj.shape[0])
    dev = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dt).cuda()
    R, t, n_in, mask, ok = ops.pnp_ransac(ctx, dev(np.asarray(offs, np.int32), torch.int32), dev(np.concatenate(objs), torch.float64),
                                          dev(np.concatenate(imgs), torch.float64), dev(np.asarray(Ks), torch.float64), iterations,
                                          reproj_error, seed, points_per_vote)
    R, t, mask, ok = R.cpu().numpy(), t.cpu().numpy(), mask.cpu().numpy(), ok.cpu().numpy()
    return [(bool(ok[p]), R[p], t[p], np.nonzero(mask[offs[p]:offs[p + 1]])[0]) for p in range(len(problems))]


def solve_pnp_ransac(objectPoints, imagePoints, cameraMatrix, distCoeffs=None, rvec=None, tvec=None, useExtrinsicGuess=False,
                     iterationsCount=300, reprojectionError=5.0, confidence=0.99, flags=None, seed=0, points_per_vote=8):
    """cv2.solvePnPRansac as called at linemod_eval.py:479-484 -> (retval, rvec [3,1], tvec [3,1], inliers [m,1] int32).
    distCoeffs must be None (the reference passes None), no extrinsic guess; `confidence` is accepted and unused: all
    iterations run."""
    if distCoeffs is not None and np.any(np.asarray(distCoeffs) != 0):
        raise ValueError("solve_pnp_ransac: lens distortion is not supported (the reference passes distCoeffs=None)")
    if useExtrinsicGuess:
        raise ValueError("solve_pnp_ransac: useExtrinsicGuess is not supported")
    n = np.asarray(objectPoints).reshape(-1, 3).shape[0]
    ppv = points_per_vote if (points_per_vote and n % points_per_vote == 0) else 0
    ok, R, t, inl = solve_pnp_batch([(objectPoints, imagePoints, cameraMatrix)], iterationsCount, reprojectionError, seed, ppv)[0]
    return ok, rotation_vector(R), t.reshape(3, 1), inl.astype(np.int32).reshape(-1, 1)
