#!/usr/bin/env python3
"""Golden vectors for the ADD / ADD-S (ADI) pose errors (test infrastructure, runs ONLY in the build container).

Loads the reference's unmodified ``PyraPose/utils/pose_error.py`` from /root/reference (module-level imports of
``cv2`` and ``transforms3d`` are satisfied by empty stub modules -- `add`, `adi` and `transform_pts_Rt`, the three
functions exercised here (pose_error.py:64-75, 210-246), use neither) and writes tests/golden/pose_error.npz:
inputs (model points, estimated and ground-truth poses) and the reference's float64 outputs.  Nothing under
tests/ with ``-m gpu`` reads /root/reference.
"""
import importlib.util
import os
import sys
import types

import numpy as np

REF = os.environ.get("PYRAPOSE_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def load_reference():
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    t3d = types.ModuleType("transforms3d")
    t3d.quaternions = types.ModuleType("transforms3d.quaternions")
    t3d.quaternions.quat2mat = None
    t3d.quaternions.mat2quat = None
    sys.modules.setdefault("transforms3d", t3d)
    sys.modules.setdefault("transforms3d.quaternions", t3d.quaternions)
    spec = importlib.util.spec_from_file_location("ref_pose_error", os.path.join(REF, "PyraPose/utils/pose_error.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def rot(rng):
    q, r = np.linalg.qr(rng.standard_normal((3, 3)))
    q = q * np.sign(np.diag(r))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    return q


def main():
    ref = load_reference()
    rng = np.random.default_rng(2026)
    out = {}
    cases = []
    for ci, (n_pts, n_pose, noise_r, noise_t, scale) in enumerate([(257, 6, 0.02, 0.004, 0.1), (1500, 5, 0.3, 0.05, 0.08),
                                                                   (64, 4, 0.0, 0.0, 0.2), (3000, 3, 0.05, 0.01, 0.15)]):
        # an anisotropic "object" (metres) so that ADD and ADI differ; case 2 has est == gt exactly
        pts = rng.standard_normal((n_pts, 3)) * np.array([1.0, 0.6, 0.25]) * scale
        if ci == 3:  # symmetric point set: ADI << ADD under a 180 degree flip
            pts = np.concatenate([pts[: n_pts // 2], pts[: n_pts // 2] * np.array([-1.0, -1.0, 1.0])])
        R_gt = np.stack([rot(rng) for _ in range(n_pose)])
        t_gt = rng.uniform(-0.3, 0.3, size=(n_pose, 3)) + np.array([0, 0, 0.9])
        R_est, t_est = [], []
        for k in range(n_pose):
            d = rot(rng)
            w = noise_r
            Rk = R_gt[k] @ (np.eye(3) * (1 - w) + d * w)
            u, _, vt = np.linalg.svd(Rk)
            Rk = u @ vt
            if ci == 3 and k == 0:
                Rk = R_gt[k] @ np.diag([-1.0, -1.0, 1.0])
            R_est.append(Rk)
            t_est.append(t_gt[k] + rng.standard_normal(3) * (0.0 if (ci == 3 and k == 0) else noise_t))
        R_est, t_est = np.stack(R_est), np.stack(t_est)
        add = np.array([ref.add(R_est[k], t_est[k], R_gt[k], t_gt[k], pts) for k in range(n_pose)], np.float64)
        adi = np.array([ref.adi(R_est[k], t_est[k], R_gt[k], t_gt[k], pts) for k in range(n_pose)], np.float64)
        tr = ref.transform_pts_Rt(pts, R_est[0], t_est[0])
        for name, val in (("pts", pts), ("R_est", R_est), ("t_est", t_est), ("R_gt", R_gt), ("t_gt", t_gt), ("add", add), ("adi", adi),
                          ("transformed0", tr)):
            out["c%d_%s" % (ci, name)] = np.asarray(val, np.float64)
        cases.append(ci)
    out["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(HERE, "pose_error.npz"), **out)
    print("wrote pose_error.npz:", {k: v.shape for k, v in out.items() if k.endswith("add")})


if __name__ == "__main__":
    main()
