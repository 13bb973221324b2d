"""CPU: the RANSAC-PnP restatement (oracle/pnp_np.py).  Parity with cv2.solvePnPRansac is UNPINNED (OpenCV is neither in
the reference tree nor installed); what is checked is the contract of the call at utils/linemod_eval.py:479-485 on the
reference's data layout (k votes x 8 cuboid corners, 5 px inlier rule) -- exact answers on noise-free data, recovery
under noise and outlier votes, the ADD decision of linemod_eval.py:530 -- and the building blocks."""
import numpy as np
import pytest

from oracle import pnp_np as P

K4 = (572.4114, 573.57043, 325.2611, 242.04899)  # LineMOD intrinsics (preprocessing/generator.py:246)
BOX = np.array([[sx * 40.0, sy * 30.0, sz * 55.0] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)])


def project(R, t, X):
    Xc = X @ R.T + t
    return np.stack([K4[0] * Xc[:, 0] / Xc[:, 2] + K4[2], K4[1] * Xc[:, 1] / Xc[:, 2] + K4[3]], axis=1)


def make_votes(rng, k, noise, outlier_frac):
    R = P.so3_exp(rng.normal(size=3))
    t = np.array([rng.uniform(-150, 150), rng.uniform(-100, 100), rng.uniform(600, 1100)])
    votes = project(R, t, BOX)[None] + rng.normal(scale=noise, size=(k, 8, 2)) if noise > 0 else np.repeat(project(R, t, BOX)[None], k, 0)
    bad = rng.uniform(size=k) < outlier_frac
    votes[bad] += rng.normal(scale=40.0, size=(int(bad.sum()), 8, 2))
    return R, t, np.tile(BOX, (k, 1)), votes.reshape(-1, 2), np.repeat(~bad, 8)


def rot_err_deg(Ra, Rb):
    return float(np.degrees(np.arccos(np.clip((np.trace(Ra.T @ Rb) - 1) / 2, -1, 1))))


def test_splitmix_known_answer_and_sampler():
    assert P.splitmix64(0) == 0xE220A8397B1DCDAF  # reference value of the published splitmix64 for state 0
    for it in range(50):
        idx = P.sample(3, 1, it, 80, 8)
        assert len(idx) == 8 and idx[0] % 8 == 0 and idx == list(range(idx[0], idx[0] + 8)) and idx[7] < 80
        assert it >= 10 or idx[0] == 8 * it  # every vote once first
        idx = P.sample(3, 1, it, 57, 0)
        assert len(set(idx)) == 6 and all(0 <= i < 57 for i in idx)
    assert P.sample(0, 0, 0, 5, 0) is None


def test_jacobi_and_small_solvers():
    rng = np.random.default_rng(0)
    A = rng.normal(size=(12, 12)); S = A @ A.T
    w, V = P.jacobi_eigh(S)
    assert np.allclose(np.sort(w), np.linalg.eigvalsh(S), rtol=1e-10, atol=1e-10)
    assert np.allclose(V @ np.diag(w) @ V.T, S, atol=1e-9)
    H = rng.normal(size=(6, 6)); H = H @ H.T + np.eye(6); g = rng.normal(size=6)
    assert np.allclose(P.solve6(H, g), np.linalg.solve(H, g), atol=1e-12)
    assert P.solve6(-np.eye(6), g) is None
    M = rng.normal(size=(3, 3))
    inv, det = P.inv3(M)
    assert np.allclose(inv, np.linalg.inv(M)) and np.isclose(det, np.linalg.det(M))
    R = P.so3_exp(np.array([0.3, -1.2, 0.5]))
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-14) and np.isclose(np.linalg.det(R), 1.0)


def test_noise_free_votes_give_the_exact_pose():
    rng = np.random.default_rng(1)
    R, t, obj, img, clean = make_votes(rng, 5, 0.0, 0.0)
    ok, Re, te, mask = P.solve_pnp_ransac(obj, img, K4, iterations=20, seed=1)
    assert ok and mask.all()
    assert rot_err_deg(Re, R) < 1e-4 and np.abs(te - t).max() < 1e-5  # (arccos near 1 resolves ~1e-6 deg)


@pytest.mark.parametrize("seed", [0, 1])
def test_recovers_pose_under_noise_and_outlier_votes(seed):
    rng = np.random.default_rng(10 + seed)
    R, t, obj, img, clean = make_votes(rng, 40, 1.5, 0.3)
    ok, Re, te, mask = P.solve_pnp_ransac(obj, img, K4, iterations=300, reproj_error=5.0, seed=seed)
    assert ok
    assert rot_err_deg(Re, R) < 1.5 and np.abs(te - t)[:2].max() < 2.0 and abs(te[2] - t[2]) < 0.02 * t[2]
    assert (mask & clean).sum() >= 0.97 * clean.sum() and (mask & ~clean).sum() <= 0.05 * max(1, (~clean).sum())
    # the decision of linemod_eval.py:530: ADD of the cuboid corners below 10 % of the diameter
    add = np.linalg.norm((BOX @ Re.T + te) - (BOX @ R.T + t), axis=1).mean()
    assert add < 0.1 * np.linalg.norm(BOX.max(0) - BOX.min(0))


def test_too_few_points_fail_cleanly():
    ok, R, t, mask = P.solve_pnp_ransac(BOX[:5], np.zeros((5, 2)), K4, iterations=10, points_per_vote=0)
    assert not ok and not mask.any() and np.allclose(R, np.eye(3))
