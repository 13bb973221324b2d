"""numpy restatement of layers/filter_detections.py:21-118 (class-specific filter, NMS on).
TEST INFRASTRUCTURE.  "PARITY UNPINNED": tf.image.non_max_suppression / tf.nn.top_k (TF 2.1.1) are
third-party and absent; their published semantics are restated: greedy NMS in descending score order
(ties -> lower index first), suppress when IoU > threshold with TF's corner-order-agnostic IoU,
at most max_output_size survivors; top_k sorted descending, ties -> lower index first."""
import numpy as np


def tf_iou(a, b):
    a = np.asarray(a, np.float32); b = np.asarray(b, np.float32)
    ay0, ax0 = min(a[0], a[2]), min(a[1], a[3]); ay1, ax1 = max(a[0], a[2]), max(a[1], a[3])
    by0, bx0 = min(b[0], b[2]), min(b[1], b[3]); by1, bx1 = max(b[0], b[2]), max(b[1], b[3])
    area_a = np.float32(ay1 - ay0) * np.float32(ax1 - ax0)
    area_b = np.float32(by1 - by0) * np.float32(bx1 - bx0)
    if area_a <= 0 or area_b <= 0:
        return np.float32(0)
    iy0, ix0 = max(ay0, by0), max(ax0, bx0); iy1, ix1 = min(ay1, by1), min(ax1, bx1)
    inter = np.float32(max(np.float32(iy1 - iy0), np.float32(0))) * np.float32(max(np.float32(ix1 - ix0), np.float32(0)))
    return np.float32(inter / np.float32(np.float32(area_a + area_b) - inter))


def non_max_suppression(boxes, scores, max_output_size, iou_threshold):
    order = sorted(range(len(scores)), key=lambda i: (-float(scores[i]), i))
    keep = []
    for i in order:
        if len(keep) >= max_output_size:
            break
        if all(not (tf_iou(boxes[i], boxes[j]) > np.float32(iou_threshold)) for j in keep):
            keep.append(i)
    return np.asarray(keep, np.int64)


def filter_detections(boxes, boxes3d, classification, score_threshold=0.05, max_detections=300, nms_threshold=0.5):
    all_idx = []
    for c in range(classification.shape[1]):
        scores = classification[:, c]
        idx = np.nonzero(scores > np.float32(score_threshold))[0]               # :56
        if len(idx):
            keep = non_max_suppression(boxes[idx], scores[idx], max_detections, nms_threshold)   # :63
            idx = idx[keep]
        all_idx.extend((int(i), c) for i in idx)                               # :69-70, :83
    sc = np.asarray([classification[i, c] for i, c in all_idx], np.float32)
    k = min(max_detections, len(all_idx))
    top = sorted(range(len(all_idx)), key=lambda t: (-float(sc[t]), t))[:k]    # :92-94 top_k
    ob = -np.ones((max_detections, 4), np.float32); o3 = -np.ones((max_detections, 16), np.float32)
    osc = -np.ones((max_detections,), np.float32); ol = -np.ones((max_detections,), np.int32)
    for r, t in enumerate(top):
        i, c = all_idx[t]
        ob[r] = boxes[i]; o3[r] = boxes3d[i]; osc[r] = sc[t]; ol[r] = c
    return ob, o3, osc, ol
