"""CPU restatement (numpy, float64) of the RANSAC-PnP step of the pose-decode tail.  TEST INFRASTRUCTURE ONLY: imported
by tests/, never by the product path (pyrapose_amd/csrc/pnp.hip is the product).

What it stands in for: `cv2.solvePnPRansac(objectPoints, imagePoints, K, None, iterationsCount=300,
reprojectionError=5.0, confidence=0.99, flags=cv2.SOLVEPNP_ITERATIVE)` at utils/linemod_eval.py:479-484 (same call in
occlusion_eval.py / ycbv_eval.py / tless_eval.py), fed with the k votes x 8 projected cuboid corners of one class
(linemod_eval.py:421-431) and followed by cv2.Rodrigues (:485).

**Parity unpinned.**  OpenCV is a third-party dependency that is not under /root/reference and not installed here
(opencv-python, unpinned in the reference's Dockerfile); its RANSAC draws from its own RNG and its minimal solver / final
refinement are internal, and the reference holds no test or golden vector for this call.  This file therefore restates
the algorithm of OUR kernel (documented below), not OpenCV's; what is checked against the reference's semantics is the
contract of the call -- inputs, the 5 px inlier rule, outputs (rotation, translation, inlier list) -- and, statistically,
that poses are recovered from the same kind of data (tests/test_oracle_pnp.py).

Algorithm (identical, operation for operation where order matters, in pnp.hip):
  1. hypotheses it = 0 .. iterations-1: a minimal sample -- with points_per_vote = 8 the eight corners of ONE vote (every
     vote once, then random votes; counter-based splitmix64 draws), otherwise six correspondences -- normalised DLT
     (object points centred and scaled to unit RMS, image points in normalised camera coordinates), null vector of the
     12x12 normal matrix by cyclic Jacobi, [R|t] by polar decomposition (Newton iteration) of the left 3x3 block, then
     5 damped Gauss-Newton steps on the sample itself (the DLT does not know that a pose has 6 degrees of freedom);
  2. score: points with squared reprojection error < reproj_error^2 in front of the camera; best count wins, ties go to
     the lower iteration;
  3. refine on the inliers: damped Gauss-Newton (left-multiplied rotation increments), 10 iterations, re-select inliers,
     10 more; the final inlier set is returned.
"""
import numpy as np

MASK64 = (1 << 64) - 1
POLISH_ITERS = 5


def splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & MASK64
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
    return z ^ (z >> 31)


def draw(seed, problem, it, j):
    key = ((seed & 0xFFFFFFFF) << 32) ^ ((problem & 0xFFF) << 20) ^ ((it & 0xFFFF) << 4) ^ (j & 0xF)
    return splitmix64(key) >> 11


def sample(seed, problem, it, n, ppv):
    """indices of one minimal sample: with vote structure the ppv points of ONE vote (the it-th vote for it < number of
    votes, a random one afterwards); without, six correspondences at a random start and stride.  None: impossible."""
    if ppv > 0:
        nv = n // ppv
        if ppv < 6 or nv < 1:
            return None
        v = it if it < nv else int(draw(seed, problem, it, 0) % nv)
        return [v * ppv + j for j in range(ppv)]
    if n < 6:
        return None
    start = int(draw(seed, problem, it, 0) % n)
    step = 1 + int(draw(seed, problem, it, 1) % max(1, (n - 1) // 6))
    return [(start + j * step) % n for j in range(6)]


def jacobi_eigh(S, sweeps=8):
    """cyclic Jacobi on a symmetric matrix (in place on copies); returns (eigenvalues, eigenvectors in columns)"""
    A = np.array(S, dtype=np.float64)
    n = A.shape[0]
    V = np.eye(n)
    for _ in range(sweeps):
        for p in range(n - 1):
            for q in range(p + 1, n):
                apq = A[p, q]
                if abs(apq) < 1e-300:
                    continue
                theta = (A[q, q] - A[p, p]) / (2.0 * apq)
                t = (1.0 if theta >= 0 else -1.0) / (abs(theta) + np.sqrt(theta * theta + 1.0))
                c = 1.0 / np.sqrt(t * t + 1.0)
                s = t * c
                colp, colq = A[:, p].copy(), A[:, q].copy()
                A[:, p] = c * colp - s * colq
                A[:, q] = s * colp + c * colq
                rowp, rowq = A[p, :].copy(), A[q, :].copy()
                A[p, :] = c * rowp - s * rowq
                A[q, :] = s * rowp + c * rowq
                vp, vq = V[:, p].copy(), V[:, q].copy()
                V[:, p] = c * vp - s * vq
                V[:, q] = s * vp + c * vq
    return np.diag(A).copy(), V


def inv3(M):
    a, b, c = M[0]; d, e, f = M[1]; g, h, i = M[2]
    A_, B_, C_ = e * i - f * h, c * h - b * i, b * f - c * e
    det = a * A_ + d * B_ + g * C_
    inv = np.array([[A_, B_, C_], [f * g - d * i, a * i - c * g, c * d - a * f], [d * h - e * g, b * g - a * h, a * e - b * d]]) / det
    return inv, det


def dlt_pose(X, xn):
    """X [m,3] object points (m >= 6, not coplanar), xn [m,2] normalised image points -> (R, t) or None"""
    c = X.mean(axis=0)
    d = X - c
    rms = np.sqrt((d * d).sum() / X.shape[0])
    if not rms > 0:
        return None
    s = 1.0 / rms
    Xn = d * s
    S = np.zeros((12, 12))
    for (Xi, (x, y)) in zip(Xn, xn):
        r1 = np.array([Xi[0], Xi[1], Xi[2], 1.0, 0, 0, 0, 0, -x * Xi[0], -x * Xi[1], -x * Xi[2], -x])
        r2 = np.array([0, 0, 0, 0, Xi[0], Xi[1], Xi[2], 1.0, -y * Xi[0], -y * Xi[1], -y * Xi[2], -y])
        S += np.outer(r1, r1) + np.outer(r2, r2)
    w, V = jacobi_eigh(S)
    k = int(np.argmin(w))
    p = V[:, k].reshape(3, 4)
    M = p[:, :3] * s
    p4 = p[:, 3] - M @ c
    _, det = inv3(M)
    if not abs(det) > 1e-300:
        return None
    if det < 0:
        M, p4 = -M, -p4
        det = -det
    lam = np.cbrt(det)
    R = M / lam
    for _ in range(12):  # polar decomposition: R <- (R + R^-T) / 2
        Ri, dR = inv3(R)
        if not abs(dR) > 1e-300:
            return None
        R = 0.5 * (R + Ri.T)
    lam = (R * M).sum() / 3.0
    if not lam > 0:
        return None
    t = p4 / lam
    return R, t


def reproj_sq(R, t, X, uv, K4):
    fx, fy, cx, cy = K4
    Xc = X @ R.T + t
    z = Xc[:, 2]
    ok = z > 1e-9
    zs = np.where(ok, z, 1.0)
    du = fx * Xc[:, 0] / zs + cx - uv[:, 0]
    dv = fy * Xc[:, 1] / zs + cy - uv[:, 1]
    return du * du + dv * dv, ok


def so3_exp(w):
    th = np.sqrt((w * w).sum())
    Kx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-12:
        return np.eye(3) + Kx
    return np.eye(3) + (np.sin(th) / th) * Kx + ((1 - np.cos(th)) / (th * th)) * (Kx @ Kx)


def solve6(H, g):
    """Cholesky solve of the 6x6 system H d = g (H symmetric positive definite); None when it is not"""
    L = np.zeros((6, 6))
    for i in range(6):
        for j in range(i + 1):
            s = H[i, j] - (L[i, :j] * L[j, :j]).sum()
            if i == j:
                if not s > 0:
                    return None
                L[i, i] = np.sqrt(s)
            else:
                L[i, j] = s / L[j, j]
    y = np.zeros(6)
    for i in range(6):
        y[i] = (g[i] - (L[i, :i] * y[:i]).sum()) / L[i, i]
    d = np.zeros(6)
    for i in reversed(range(6)):
        d[i] = (y[i] - (L[i + 1:, i] * d[i + 1:]).sum()) / L[i, i]
    return d


def refine(R, t, X, uv, K4, mask, iters=10):
    fx, fy, cx, cy = K4
    lam = 1e-3

    def cost(R_, t_):
        e, ok = reproj_sq(R_, t_, X, uv, K4)
        return float(np.where(ok, e, 1e12)[mask].sum())

    cur = cost(R, t)
    for _ in range(iters):
        Xr = X @ R.T
        Xc = Xr + t
        H = np.zeros((6, 6))
        g = np.zeros(6)
        for i in np.nonzero(mask)[0]:
            x, y, z = Xc[i]
            if not z > 1e-9:
                continue
            ru = fx * x / z + cx - uv[i, 0]
            rv = fy * y / z + cy - uv[i, 1]
            a, b, c = Xr[i]
            ju = np.array([fx / z, 0.0, -fx * x / (z * z)])
            jv = np.array([0.0, fy / z, -fy * y / (z * z)])
            # d Xc / d w = -[Xr]x ;  d Xc / d t = I
            Ju = np.array([ju[2] * b - ju[1] * c, ju[0] * c - ju[2] * a, ju[1] * a - ju[0] * b, ju[0], ju[1], ju[2]])
            Jv = np.array([jv[2] * b - jv[1] * c, jv[0] * c - jv[2] * a, jv[1] * a - jv[0] * b, jv[0], jv[1], jv[2]])
            H += np.outer(Ju, Ju) + np.outer(Jv, Jv)
            g -= Ju * ru + Jv * rv
        Hd = H + lam * np.diag(np.diag(H)) + 1e-12 * np.eye(6)
        d = solve6(Hd, g)
        if d is None:
            break
        R2 = so3_exp(d[:3]) @ R
        t2 = t + d[3:]
        c2 = cost(R2, t2)
        if c2 < cur:
            R, t, cur = R2, t2, c2
            lam = max(lam * 0.1, 1e-9)
        else:
            lam = min(lam * 10.0, 1e6)
    return R, t


def solve_pnp_ransac(obj, img, K4, iterations=300, reproj_error=5.0, seed=0, problem=0, points_per_vote=8):
    """obj [n,3], img [n,2] float64, K4 = (fx, fy, cx, cy) -> (ok, R [3,3], t [3], inlier mask [n] bool)"""
    obj = np.asarray(obj, np.float64); img = np.asarray(img, np.float64)
    n = obj.shape[0]
    fx, fy, cx, cy = [float(v) for v in K4]
    xn = np.stack([(img[:, 0] - cx) / fx, (img[:, 1] - cy) / fy], axis=1)
    thr2 = float(reproj_error) ** 2
    best = (-1, -1, None, None)
    for it in range(iterations):
        idx = sample(seed, problem, it, n, points_per_vote)
        if idx is None:
            break
        hyp = dlt_pose(obj[idx], xn[idx])
        if hyp is None:
            continue
        # the DLT ignores that [R|t] has 6 degrees of freedom: polish on the sample itself before scoring
        R, t = refine(hyp[0], hyp[1], obj[idx], img[idx], (fx, fy, cx, cy), np.ones(len(idx), bool), iters=POLISH_ITERS)
        e, ok = reproj_sq(R, t, obj, img, (fx, fy, cx, cy))
        cnt = int((ok & (e < thr2)).sum())
        if cnt > best[0]:
            best = (cnt, it, R, t)
    if best[0] < 4:
        return False, np.eye(3), np.zeros(3), np.zeros(n, bool)
    _, _, R, t = best
    for _ in range(2):
        e, ok = reproj_sq(R, t, obj, img, (fx, fy, cx, cy))
        mask = ok & (e < thr2)
        R, t = refine(R, t, obj, img, (fx, fy, cx, cy), mask)
    e, ok = reproj_sq(R, t, obj, img, (fx, fy, cx, cy))
    mask = ok & (e < thr2)
    return bool(mask.sum() >= 4), R, t, mask
