"""GPU: anchor generation / IoU / target assignment / decode kernels through the C ABI against the
golden vectors of the reference's own code and against the numpy oracle on seeded inputs.
Bar: bit-exact for indices, states, labels, masks, float64 anchors and IoU; regression targets
within 1 float32 ulp (the reference rotates the cuboid with a BLAS float64 dot)."""
import numpy as np
import pytest
import torch

from conftest import unpack_annotations

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from pyrapose_amd.runtime import default_context
    return default_context()


@pytest.mark.parametrize("hw", [(480, 640), (540, 720), (97, 131)])
def test_anchor_grid_f64_bit_exact(ctx, golden, hw):
    from pyrapose_amd.utils import anchors as UA
    got = UA.anchors_for_shape(hw)
    assert got.dtype == np.float64
    assert np.array_equal(got, golden["anchors_%dx%d" % hw])


def test_anchor_grid_f32_matches_keras_layer_path(ctx):
    from oracle import anchors_np as O
    from pyrapose_amd.utils import anchors as UA
    for hw in ((480, 640), (540, 720)):
        got = UA.anchors_for_shape_device(hw, dtype=torch.float32).cpu().numpy()
        assert np.array_equal(got, O.anchors_for_shape_f32(hw))


def test_shift_single_level(ctx, golden):
    from oracle import anchors_np as O
    from pyrapose_amd.utils import anchors as UA
    base = golden["base_anchors_64"]
    assert np.array_equal(UA.shift((30, 40), 16, base), O.shift((30, 40), 16, base))


@pytest.mark.parametrize("name", ["k1", "k5", "edge"])
def test_overlap_and_assignment_bit_exact(ctx, golden, name):
    from pyrapose_amd.utils import anchors as UA
    boxes = golden["iou_edge_boxes"] if name == "edge" else golden["anchors_480x640"]
    q = golden["iou_%s_query" % name]
    ov = UA.compute_overlap(np.ascontiguousarray(boxes), np.ascontiguousarray(q))
    assert np.array_equal(ov, golden["iou_%s_overlaps" % name])
    pos, ign, amax = UA.compute_gt_annotations(boxes, q)
    assert np.array_equal(pos, golden["iou_%s_positive" % name])
    assert np.array_equal(ign, golden["iou_%s_ignore" % name])
    assert np.array_equal(amax, golden["iou_%s_argmax" % name])


def test_overlap_errors_and_empty(ctx):
    from pyrapose_amd.utils.compute_overlap import compute_overlap
    b = np.array([[10.0, 20.0, 50.0, 80.0]])
    assert compute_overlap(b, b)[0, 0] == 1.0
    with pytest.raises(ValueError):
        compute_overlap(b.astype(np.float32), b)
    with pytest.raises(ValueError):
        compute_overlap(b[0], b)
    assert compute_overlap(np.zeros((0, 4)), b).shape == (0, 1)
    assert compute_overlap(b, np.zeros((0, 4))).shape == (1, 0)


def ulp_diff_f32(a, b):
    ai = np.asarray(a, np.float32).view(np.int32).astype(np.int64)
    bi = np.asarray(b, np.float32).view(np.int32).astype(np.int64)
    ai = np.where(ai < 0, -(ai & 0x7fffffff), ai)
    bi = np.where(bi < 0, -(bi & 0x7fffffff), bi)
    return np.abs(ai - bi)


@pytest.mark.parametrize("prefix", ["tgt_small_identity", "tgt_small_general", "tgt_identity", "tgt_general",
                                    "tgt_tless_identity"])
def test_anchor_targets_vs_golden(ctx, golden, prefix):
    from pyrapose_amd.utils import anchors as UA
    H, W, C = (int(v) for v in golden[prefix + "_meta"])
    shapes = [tuple(int(x) for x in s) for s in golden[prefix + "_image_shapes"]]
    anns = unpack_annotations(golden, prefix, len(shapes))
    images = [np.zeros((h, w, 3), np.float32) for h, w in shapes]
    anchors = UA.anchors_for_shape((H, W))
    reg, lab, msk = UA.anchor_targets_bbox(anchors, images, anns, C)
    assert reg.dtype == np.float32 and lab.dtype == np.float32 and msk.dtype == np.float32
    assert np.array_equal(lab, golden[prefix + "_labels"])
    assert np.array_equal(msk, golden[prefix + "_mask"])
    assert np.array_equal(reg[:, :, -1].astype(np.int8), golden[prefix + "_reg_state"])
    if prefix + "_regression" in golden.files:
        want, got = golden[prefix + "_regression"], reg
    else:
        idx = golden[prefix + "_reg_rows_idx"]
        want, got = golden[prefix + "_reg_rows"], reg[idx[:, 0], idx[:, 1], :]
    assert ulp_diff_f32(got, want).max() <= 1


def test_anchor_targets_seeded_vs_oracle(ctx):
    """Fresh seeded annotations (not in the fixtures), incl. an empty image and a cropped one."""
    from oracle import anchors_np as O
    from pyrapose_amd.utils import anchors as UA
    rng = np.random.default_rng(123)
    H, W, C = 480, 640, 21
    anns, images = [], []
    for (h, w, K) in ((480, 640, 5), (480, 640, 0), (420, 560, 2), (480, 640, 1)):
        mask = np.zeros((h, w), np.uint8)
        a = {"mask": [mask], "labels": np.empty((0,)), "bboxes": np.empty((0, 4)), "poses": np.empty((0, 7)),
             "segmentations": np.empty((0, 8, 3)), "cam_params": np.empty((0, 4)), "mask_ids": np.empty((0,))}
        for k in range(K):
            bw, bh = rng.uniform(40, 160, 2)
            x1, y1 = rng.uniform(0, w - bw), rng.uniform(0, h - bh)
            mask[int(y1):int(y1 + bh), int(x1):int(x1 + bw)] = k + 1
            q = rng.normal(size=4); q /= np.linalg.norm(q)
            box = rng.uniform(-80, 80, size=(8, 3)).astype(np.float32)
            a["labels"] = np.concatenate([a["labels"], [float(rng.integers(0, C))]])
            a["bboxes"] = np.concatenate([a["bboxes"], [[x1, y1, x1 + bw, y1 + bh]]])
            a["poses"] = np.concatenate([a["poses"], [[rng.uniform(-100, 100), rng.uniform(-100, 100), 800.0, *q]]])
            a["segmentations"] = np.concatenate([a["segmentations"], [box]])
            a["cam_params"] = np.concatenate([a["cam_params"], [[572.4114, 573.57043, 325.2611, 242.04899]]])
            a["mask_ids"] = np.concatenate([a["mask_ids"], [float(k + 1)]])
        anns.append(a); images.append(np.zeros((h, w, 3), np.float32))
    anchors = O.anchors_for_shape((H, W))
    reg_o, lab_o, msk_o = O.anchor_targets_bbox(anchors, [im.shape[:2] for im in images], anns, C)
    reg, lab, msk = UA.anchor_targets_bbox(anchors, images, anns, C)
    assert np.array_equal(lab, lab_o) and np.array_equal(msk, msk_o)
    assert np.array_equal(reg[:, :, -1], reg_o[:, :, -1])
    assert ulp_diff_f32(reg, reg_o).max() <= 1


def test_decode_bit_exact_and_round_trip(ctx):
    from oracle import anchors_np as O
    from pyrapose_amd import ops
    from pyrapose_amd.utils import anchors as UA
    rng = np.random.default_rng(7)
    anc = UA.anchors_for_shape_device((480, 640), dtype=torch.float32)
    reg = torch.as_tensor(rng.standard_normal((2, anc.shape[0], 16)), dtype=torch.float32).cuda()
    got = ops.box3d_decode(ctx, anc, reg).cpu().numpy()
    want = O.box3d_transform_inv_f32(anc.cpu().numpy()[None], reg.cpu().numpy())
    assert np.array_equal(got, want)


def test_score_threshold_compaction(ctx):
    from oracle import anchors_np as O
    from pyrapose_amd import ops
    rng = np.random.default_rng(11)
    B, N, C = 3, 56700, 8
    scores = rng.uniform(0, 0.505, size=(B, N, C)).astype(np.float32)   # ~1 % above 0.5
    scores[1] = 0.0            # empty image
    scores[2, :, 3] = 0.9      # every anchor fires for one class (maximum size)
    idx, cnt = ops.score_threshold_compact(ctx, torch.from_numpy(scores).cuda(), 0.5)
    idx, cnt = idx.cpu().numpy(), cnt.cpu().numpy()
    for b in range(B):
        want = O.score_threshold_indices(scores[b], 0.5)
        for c in range(C):
            assert cnt[b, c] == len(want[c])
            assert np.array_equal(idx[b, c, : cnt[b, c]], want[c])
            assert np.all(idx[b, c, cnt[b, c]:] == -1)


def test_filter_detections_vs_oracle(ctx):
    from oracle import detect_np as D
    from pyrapose_amd import ops
    rng = np.random.default_rng(13)
    N, C = 3000, 5
    ctr = rng.uniform(50, 400, size=(N, 2)); wh = rng.uniform(20, 120, size=(N, 2))
    boxes = np.concatenate([ctr - wh / 2, ctr + wh / 2], axis=1).astype(np.float32)
    boxes3d = rng.uniform(0, 640, size=(N, 16)).astype(np.float32)
    scores = (rng.uniform(0, 1, size=(N, C)) ** 6).astype(np.float32)
    scores[10:20, 2] = scores[10, 2]          # exact score ties
    scores[:, 4] = 0.0                        # a class with no candidates
    got = ops.filter_detections(ctx, torch.from_numpy(boxes).cuda(), torch.from_numpy(boxes3d).cuda(),
                                torch.from_numpy(scores).cuda(), 0.05, 0.5, 300)
    want = D.filter_detections(boxes, boxes3d, scores, 0.05, 300, 0.5)
    for g, w in zip(got, want):
        assert np.array_equal(g.cpu().numpy(), w)


def test_filter_detections_batch_matches_per_image(ctx):
    """pp_filter_detections_batch (grid.y = image) == the single-image entry, image by image (different candidate counts)."""
    from pyrapose_amd import ops
    rng = np.random.default_rng(21)
    B, N, C = 3, 2500, 4
    ctr = rng.uniform(50, 400, size=(B, N, 2)); wh = rng.uniform(20, 120, size=(B, N, 2))
    boxes = torch.from_numpy(np.concatenate([ctr - wh / 2, ctr + wh / 2], axis=-1).astype(np.float32)).cuda()
    boxes3d = torch.from_numpy(rng.uniform(0, 640, size=(B, N, 16)).astype(np.float32)).cuda()
    sc = rng.uniform(0, 1, size=(B, N, C)) ** np.array([2.0, 6.0, 12.0]).reshape(B, 1, 1)
    scores = torch.from_numpy(sc.astype(np.float32)).cuda()
    got = ops.filter_detections_batch(ctx, boxes, boxes3d, scores, 0.05, 0.5, 300)
    for b in range(B):
        want = ops.filter_detections(ctx, boxes[b].contiguous(), boxes3d[b].contiguous(), scores[b].contiguous(), 0.05, 0.5, 300)
        for g, w in zip(got, want):
            assert torch.equal(g[b], w)
