"""GPU: pp_pose_add_f64 / pp_pose_adi_f64 through the C ABI against the reference's vectors and the oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "pose_error.npz"))


@pytest.fixture(scope="module")
def ctx():
    from pyrapose_amd.runtime import default_context
    return default_context()


def test_add_adi_match_reference_vectors(ctx):
    from pyrapose_amd import ops
    from pyrapose_amd.utils import pose_error as PE
    for c in G["cases"]:
        k = lambda n: G["c%d_%s" % (c, n)]
        dev = lambda n: torch.from_numpy(k(n)).cuda()
        for sym, key in ((False, "add"), (True, "adi")):
            got = ops.pose_errors(ctx, dev("pts"), dev("R_est"), dev("t_est"), dev("R_gt"), dev("t_gt"), sym).cpu().numpy()
            np.testing.assert_allclose(got, k(key), rtol=1e-12, atol=1e-15)
            # the evaluation decision (linemod_eval.py:530: err < 0.1 * diameter) for a sweep of diameters
            for dia in (0.05, 0.1, 0.2, 0.5, 1.0):
                assert np.array_equal(got < 0.1 * dia, k(key) < 0.1 * dia)
        # reference-named single-pose functions (numpy in, float out)
        assert abs(PE.add(k("R_est")[0], k("t_est")[0], k("R_gt")[0], k("t_gt")[0], k("pts")) - k("add")[0]) <= 1e-12 * max(k("add")[0], 1e-3)
        assert abs(PE.adi(k("R_est")[0], k("t_est")[0].reshape(3, 1), k("R_gt")[0], k("t_gt")[0].reshape(3, 1), k("pts")) - k("adi")[0]) <= 1e-12 * max(k("adi")[0], 1e-3)


def test_large_model_matches_oracle_and_is_deterministic(ctx):
    from oracle import pose_np as P
    from pyrapose_amd import ops
    rng = np.random.default_rng(3)
    pts = rng.standard_normal((5003, 3)) * 0.07
    q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    R_gt = np.stack([np.eye(3), q * np.sign(np.linalg.det(q))])
    R_est = np.stack([q * np.sign(np.linalg.det(q)), np.eye(3)])
    t_gt = rng.uniform(-0.1, 0.1, (2, 3)); t_est = t_gt + 0.01
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    for sym, fn in ((False, P.add), (True, P.adi)):
        a = ops.pose_errors(ctx, dev(pts), dev(R_est), dev(t_est), dev(R_gt), dev(t_gt), sym)
        b = ops.pose_errors(ctx, dev(pts), dev(R_est), dev(t_est), dev(R_gt), dev(t_gt), sym)
        assert torch.equal(a, b)
        want = np.array([fn(R_est[i], t_est[i], R_gt[i], t_gt[i], pts) for i in range(2)])
        np.testing.assert_allclose(a.cpu().numpy(), want, rtol=1e-12)


def test_bad_arguments(ctx):
    from pyrapose_amd.utils import pose_error as PE
    with pytest.raises(ValueError):
        PE.add(np.eye(3), np.zeros(3), np.eye(3), np.zeros(3), np.zeros((4, 2)))
