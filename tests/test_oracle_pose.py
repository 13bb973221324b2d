"""CPU: the ADD / ADI restatement (oracle/pose_np.py) against vectors produced by the reference's own functions."""
import os

import numpy as np

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "pose_error.npz"))


def test_oracle_add_adi_match_reference_vectors():
    from oracle import pose_np as P
    for c in G["cases"]:
        k = lambda n: G["c%d_%s" % (c, n)]
        pts = k("pts")
        np.testing.assert_allclose(P.transform_pts_Rt(pts, k("R_est")[0], k("t_est")[0]), k("transformed0"), rtol=0, atol=1e-15)
        for i in range(k("R_est").shape[0]):
            a = P.add(k("R_est")[i], k("t_est")[i], k("R_gt")[i], k("t_gt")[i], pts)
            s = P.adi(k("R_est")[i], k("t_est")[i], k("R_gt")[i], k("t_gt")[i], pts)
            assert abs(a - k("add")[i]) <= 1e-13 * max(k("add")[i], 1e-3)
            assert abs(s - k("adi")[i]) <= 1e-13 * max(k("adi")[i], 1e-3)


def test_known_answers():
    from oracle import pose_np as P
    c = lambda n: G["c2_" + n]                      # est == gt
    assert np.all(c("add") < 1e-15) and np.all(c("adi") < 1e-15)   # (the generator re-orthonormalises R_est: ~1e-17)
    k = lambda n: G["c3_" + n]                      # symmetric object, pose 0 flipped by 180 degrees about z
    assert k("adi")[0] < 1e-12 and k("add")[0] > 0.01
    assert np.all(G["c0_adi"] <= G["c0_add"] + 1e-15)  # nearest-point distance never exceeds the matched-point distance
