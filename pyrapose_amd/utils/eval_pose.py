"""Pose evaluation loop: the arithmetic of utils/linemod_eval.py:evaluate_linemod (263-660) -- and of its siblings
occlusion_eval.py / ycbv_eval.py / tless_eval.py, which differ in the vote threshold and the class tables -- without the file
loaders, progress bars and OpenCV drawing around it.  Per image: network outputs -> per-class votes (score > threshold, at
least `min_votes`) -> RANSAC-PnP on the 8 cuboid corners -> ADD (ADD-S for the symmetric classes) against the ground
truth pose -> "true pose" when the error is below 10 % of the model diameter (:525-531).  Counters and rates as at :259-262
and :639-660 (index = class id + 1, as there)."""
import numpy as np

from . import pose_decode, pose_error


def quat2mat(q):
    """unit quaternion (w, x, y, z) -> rotation matrix; what the reference gets from transforms3d.quaternions.quat2mat
    (linemod_eval.py:511) for the normalised quaternions of its annotations."""
    w, x, y, z = [float(v) for v in q]
    n = w * w + x * x + y * y + z * z
    if n < 1e-12:
        return np.eye(3)
    s = 2.0 / n
    X, Y, Z = x * s, y * s, z * s
    wX, wY, wZ, xX, xY, xZ, yY, yZ, zZ = w * X, w * Y, w * Z, x * X, x * Y, x * Z, y * Y, y * Z, z * Z
    return np.array([[1.0 - (yY + zZ), xY - wZ, xZ + wY], [xY + wZ, 1.0 - (xX + zZ), yZ - wX], [xZ - wY, yZ + wX, 1.0 - (xX + yY)]])


def evaluate_add(generator, predict_on_batch, threeD_boxes, model_points, model_diameters, K=None, threshold=0.5, min_votes=10,
                 symmetric_classes=(), gt_translation_scale=0.001, seed=0):
    """generator: load_image / preprocess_image / resize_image / load_annotations / size() (preprocessing/generator.py);
    predict_on_batch: the prediction model's method (x [1,H,W,3] -> [boxes3D, scores, mask]);
    threeD_boxes [C,8,3], model_points: list of [n_c,3], model_diameters [C] -- all in the unit of the estimated translation
    (the reference works in metres and scales the annotation's millimetres by 0.001, :516);
    K: 3x3 intrinsics (default: LineMOD, :423); symmetric_classes: 0-based class ids scored with ADD-S (the reference's
    cls == 10 or 11, 1-based, :525).  Returns dict(allPoses, trueDets, truePoses, recall, detections, recall_all,
    detections_all, errors) with the reference's 1-based class indexing."""
    C = len(model_diameters)
    if K is None:
        K = np.array([[572.4114, 0.0, 325.2611], [0.0, 573.57043, 242.04899], [0.0, 0.0, 1.0]])
    allPoses, truePoses, trueDets = (np.zeros((C + 1,), np.uint32) for _ in range(3))
    errors = []
    for index in range(generator.size()):
        image = generator.preprocess_image(generator.load_image(index))
        image, _scale = generator.resize_image(image)
        anno = generator.load_annotations(index)
        if len(anno["labels"]) < 1:
            continue
        labels = [int(l) for l in anno["labels"]]
        for lab in labels:
            allPoses[lab + 1] += 1
        boxes3D, scores, _mask = predict_on_batch(np.expand_dims(image, axis=0))
        poses = pose_decode.poses_from_outputs(boxes3D, scores, threeD_boxes, K, threshold=threshold, min_votes=min_votes, seed=seed + index)
        for det in poses:
            cls = det["cls"]
            if cls not in labels:  # the reference only scores the annotated class (:327-329)
                continue
            trueDets[cls + 1] += 1
            gi = labels.index(cls)
            pose = np.asarray(anno["poses"][gi], np.float64)
            R_gt, t_gt = quat2mat(pose[3:]), pose[:3] * gt_translation_scale
            fn = pose_error.adi if cls in symmetric_classes else pose_error.add
            err = fn(det["R"], det["t"].reshape(3, 1), R_gt, t_gt.reshape(3, 1), model_points[cls])
            errors.append((index, cls, float(err)))
            if det["ok"] and err < model_diameters[cls] * 0.1:
                truePoses[cls + 1] += 1
    with np.errstate(divide="ignore", invalid="ignore"):
        recall = np.nan_to_num(truePoses / allPoses.astype(np.float64))
        detections = np.nan_to_num(trueDets / allPoses.astype(np.float64))
    seen = max(int((allPoses[1:] > 0).sum()), 1)
    return dict(allPoses=allPoses, trueDets=trueDets, truePoses=truePoses, recall=recall, detections=detections,
                recall_all=float(recall[1:].sum() / seen), detections_all=float(detections[1:].sum() / seen), errors=errors)
