"""CPU: the numpy oracle against the golden vectors generated from the reference's own code."""
import hashlib

import numpy as np
import pytest

from oracle import anchors_np as O
from conftest import unpack_annotations


def test_default_params(golden):
    p = O.default_anchor_parameters()
    assert np.array_equal(p["ratios"], golden["default_ratios"])
    assert np.array_equal(p["scales"], golden["default_scales"])


@pytest.mark.parametrize("size", [32, 64, 128, 48, 96, 192, 24, 160])
def test_base_anchors(golden, size):
    p = O.default_anchor_parameters()
    got = O.generate_anchors(size, p["ratios"], p["scales"])
    assert got.dtype == np.float64
    assert np.array_equal(got, golden[f"base_anchors_{size}"])


def test_base_anchors_four_scales(golden):
    p = O.default_anchor_parameters()
    got = O.generate_anchors(48, p["ratios"], golden["scales4"])
    assert np.array_equal(got, golden["base_anchors_48_s4"])


@pytest.mark.parametrize("hw", [(480, 640), (540, 720), (97, 131)])
def test_anchor_grid(golden, hw):
    H, W = hw
    assert np.array_equal(np.array(O.guess_shapes(hw, [3, 4, 5])), golden[f"shapes_{H}x{W}"])
    got = O.anchors_for_shape(hw)
    assert np.array_equal(got, golden[f"anchors_{H}x{W}"])


def test_shapes_p3_p7(golden):
    assert np.array_equal(np.array(O.guess_shapes((480, 640), [3, 4, 5, 6, 7])), golden["shapes_480x640_p37"])


def test_first_anchor_known_answer():
    a = O.anchors_for_shape((480, 640))
    assert a.shape == (56700, 4)
    np.testing.assert_allclose(a[0], [-18.627417, -7.3137085, 26.627417, 15.3137085], rtol=0, atol=1e-6)


@pytest.mark.parametrize("name", ["k1", "k5", "edge"])
def test_overlap_and_assignment(golden, name):
    boxes = golden["iou_edge_boxes"] if name == "edge" else golden["anchors_480x640"]
    q = golden[f"iou_{name}_query"]
    ov = O.compute_overlap(boxes, q)
    assert np.array_equal(ov, golden[f"iou_{name}_overlaps"])
    pos, ign, amax = O.compute_gt_annotations(boxes, q)
    assert np.array_equal(pos, golden[f"iou_{name}_positive"])
    assert np.array_equal(ign, golden[f"iou_{name}_ignore"])
    assert np.array_equal(amax, golden[f"iou_{name}_argmax"])


def test_overlap_self_is_one_and_errors():
    b = np.array([[10.0, 20.0, 50.0, 80.0]])
    assert O.compute_overlap(b, b)[0, 0] == 1.0
    with pytest.raises(ValueError):
        O.compute_overlap(b.astype(np.float32), b)
    with pytest.raises(ValueError):
        O.compute_overlap(b[0], b)
    assert O.compute_overlap(np.zeros((0, 4)), b).shape == (0, 1)
    assert O.compute_overlap(b, np.zeros((0, 4))).shape == (1, 0)


def test_box3d_transform(golden):
    got = O.box3d_transform(golden["b3d_anchors"], golden["b3d_gt"])
    assert np.array_equal(got, golden["b3d_targets"])


@pytest.mark.parametrize("hw", [(480, 640), (540, 720), (97, 131), (333, 517)])
def test_pil_nearest(golden, hw):
    H, W = hw
    mh, mw = O.guess_shapes(hw, [3])[0]
    assert np.array_equal(O.pil_nearest_index(H, mh), golden[f"pil_nearest_{H}x{W}_rows"])
    assert np.array_equal(O.pil_nearest_index(W, mw), golden[f"pil_nearest_{H}x{W}_cols"])


@pytest.mark.parametrize("prefix", ["tgt_small_identity", "tgt_small_general", "tgt_identity",
                                    "tgt_general", "tgt_tless_identity"])
def test_anchor_targets(golden, prefix):
    H, W, C = (int(v) for v in golden[f"{prefix}_meta"])
    shapes = [tuple(int(x) for x in s) for s in golden[f"{prefix}_image_shapes"]]
    anns = unpack_annotations(golden, prefix, len(shapes))
    anchors = O.anchors_for_shape((H, W))
    reg, lab, msk = O.anchor_targets_bbox(anchors, shapes, anns, C)
    assert np.array_equal(lab, golden[f"{prefix}_labels"])
    assert np.array_equal(msk, golden[f"{prefix}_mask"])
    assert np.array_equal(reg[:, :, -1].astype(np.int8), golden[f"{prefix}_reg_state"])
    if f"{prefix}_regression" in golden.files:
        assert np.array_equal(reg, golden[f"{prefix}_regression"])
    else:
        idx = golden[f"{prefix}_reg_rows_idx"]
        assert np.array_equal(reg[idx[:, 0], idx[:, 1], :], golden[f"{prefix}_reg_rows"])
    sha = np.frombuffer(hashlib.sha256(np.ascontiguousarray(reg).tobytes()).digest(), np.uint8)
    assert np.array_equal(sha, golden[f"{prefix}_reg_sha256"])


def test_decode_round_trip():
    # box3D_transform_inv(box3D_transform(x)) == x   (anchors.py:515-559 <-> backend/common.py:25-56)
    rng = np.random.default_rng(1)
    anchors = O.anchors_for_shape((97, 131))
    gt = rng.uniform(0, 131, size=(anchors.shape[0], 16))
    t = O.box3d_transform(anchors, gt)
    back = O.box3d_transform_inv_f32(anchors.astype(np.float32), t.astype(np.float32))
    np.testing.assert_allclose(back, gt, rtol=0, atol=2e-3)
