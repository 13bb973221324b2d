"""CPU (no GPU): the C-ABI library loads and exports every symbol include/pyrapose_hip.h declares; the
host-only entry points are checked against the golden vectors / the oracle.  No device compute here."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "pyrapose_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pp_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from pyrapose_amd import _lib
    names = header_functions()
    assert len(names) >= 30
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), "libpyrapose_hip.so does not export %s" % n
    # the Python binding declares a signature for each of them
    assert sorted(_lib.EXPORTS) == names


def test_version_and_error_codes():
    from pyrapose_amd import _lib
    assert b"gfx950" in _lib.lib.pp_version()
    with pytest.raises(ValueError):
        _lib.check(-2, None, "x")
    with pytest.raises(RuntimeError):
        _lib.check(700, None, "x")
    # a NULL context is an argument error, not a crash
    assert _lib.lib.pp_add_n(None, 4, None, None, None, None) == -4


def test_no_gpu_means_loud_failure():
    import torch
    from pyrapose_amd import ops
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        ops.Context(0)


@pytest.mark.parametrize("size", [32, 64, 128, 48, 96, 192, 24, 160])
def test_base_anchors_host_bit_exact(golden, size):
    from pyrapose_amd.utils import anchors as UA
    got = UA.generate_anchors(size)
    assert got.dtype == np.float64 and np.array_equal(got, golden["base_anchors_%d" % size])


def test_base_anchors_four_scales(golden):
    from pyrapose_amd.utils import anchors as UA
    got = UA.generate_anchors(48, UA.AnchorParameters.default.ratios, golden["scales4"])
    assert np.array_equal(got, golden["base_anchors_48_s4"])


def test_default_anchor_parameters(golden):
    from pyrapose_amd.utils import anchors as UA
    p = UA.AnchorParameters.default
    assert np.array_equal(p.ratios, golden["default_ratios"]) and np.array_equal(p.scales, golden["default_scales"])
    assert p.num_anchors() == 9 and p.sizes == [32, 64, 128] and p.strides == [8, 16, 32]
    assert [tuple(s) for s in UA.guess_shapes((480, 640), [3, 4, 5])] == [(60, 80), (30, 40), (15, 20)]


@pytest.mark.parametrize("hw", [(480, 640), (540, 720), (97, 131), (333, 517)])
def test_pil_nearest_host(golden, hw):
    from oracle import anchors_np as O
    from pyrapose_amd import ops
    H, W = hw
    mh, mw = O.guess_shapes(hw, [3])[0]
    assert np.array_equal(ops.pil_nearest_index(H, mh), golden["pil_nearest_%dx%d_rows" % hw])
    assert np.array_equal(ops.pil_nearest_index(W, mw), golden["pil_nearest_%dx%d_cols" % hw])


def test_project_box3d_host_vs_oracle():
    from oracle import anchors_np as O
    from pyrapose_amd import ops
    rng = np.random.default_rng(0)
    for _ in range(20):
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        pose = np.array([rng.uniform(-100, 100), rng.uniform(-100, 100), rng.uniform(400, 1200), *q])
        box = rng.uniform(-80, 80, size=(8, 3)).astype(np.float32)
        cam = np.array([572.4114, 573.57043, 325.2611, 242.04899])
        got = ops.project_box3d(pose, box, cam)
        want = O.project_box3d(pose, box, cam)
        np.testing.assert_allclose(got, want, rtol=1e-14, atol=1e-11)
    # identity rotation: stub-independent
    pose = np.array([10.0, -20.0, 800.0, 1.0, 0.0, 0.0, 0.0])
    assert np.array_equal(ops.project_box3d(pose, box, cam), O.project_box3d(pose, box, cam))


def test_c_oracle_matches_numpy_oracle(golden):
    """oracle/c/pp_oracle.c (built by __graft_entry__.build) against the golden IoU vectors."""
    path = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
    if not os.path.exists(path):
        pytest.skip("oracle C library not built")
    lib = ctypes.CDLL(path)
    boxes = np.ascontiguousarray(golden["anchors_480x640"])
    q = np.ascontiguousarray(golden["iou_k5_query"])
    out = np.empty((boxes.shape[0], q.shape[0]))
    dp = ctypes.POINTER(ctypes.c_double)
    lib.oracle_compute_overlap(boxes.ctypes.data_as(dp), boxes.shape[0], q.ctypes.data_as(dp), q.shape[0], out.ctypes.data_as(dp))
    assert np.array_equal(out, golden["iou_k5_overlaps"])
    am = np.empty(boxes.shape[0], np.int32); st = np.empty(boxes.shape[0], np.int8)
    lib.oracle_gt_annotations(out.ctypes.data_as(dp), boxes.shape[0], q.shape[0], ctypes.c_double(0.4), ctypes.c_double(0.5),
                              am.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), st.ctypes.data_as(ctypes.POINTER(ctypes.c_byte)))
    assert np.array_equal(am, golden["iou_k5_argmax"])
    assert np.array_equal(st == 1, golden["iou_k5_positive"]) and np.array_equal(st == -1, golden["iou_k5_ignore"])


def test_arch_matches_baseline_numbers():
    """Shape walk vs BASELINE.md §3: forward 234.2 GFLOP/img, 42.45 M parameters at C=13."""
    from pyrapose_amd import arch
    fwd, bwd = arch.conv_flops(13, 480, 640)
    assert abs(fwd / 1e9 - 234.2) < 0.1
    # BASELINE counts a data gradient into the frozen C2 for res3a (1.57 GFLOP) that nothing consumes
    assert abs((fwd + bwd) / 1e9 + 1.573 - 683.2) < 0.2
    W = arch.init_weights(13, 0)
    n = sum(v.size for k, v in W.items() if k.endswith("/kernel") or k.endswith("/bias"))
    assert abs(n / 1e6 - 42.45) < 0.01
    assert arch.level_shapes(540, 720) == [(68, 90), (34, 45), (17, 23)]
