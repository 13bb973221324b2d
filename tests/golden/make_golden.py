#!/usr/bin/env python3
"""Golden-vector generator (test infrastructure, runs ONLY in the build container).

Imports the reference's *unmodified* ``PyraPose/utils/anchors.py`` and a local Cython
build of its ``PyraPose/utils/compute_overlap.pyx`` straight from ``/root/reference``
(nothing is copied into this repo) and writes small ``.npz`` fixtures next to this
script.  Procedure = SURVEY.md §8(c):

* a scratch package tree under a temp dir: empty ``PyraPose/__init__.py`` and
  ``PyraPose/utils/__init__.py``, a *symlink* to the reference ``anchors.py`` and the
  freshly cythonized ``compute_overlap`` extension (built from the reference path);
* three stub modules, none of which does arithmetic on this path except ``quat2mat``:
  ``keras.backend.floatx() -> 'float32'`` (anchors.py:51-52,110-115), ``cv2`` (imported,
  unused) and ``transforms3d.quaternions.quat2mat`` (anchors.py:207).  The quat2mat stub
  is the published transforms3d 0.3.1 formula restated by us, so fixtures whose key ends
  in ``_identity`` use only identity rotations (stub-independent); the ``_general`` ones
  depend on the restated formula and are flagged as such in DESIGN.md.

The fixtures are data (inputs + expected outputs); this script is the only thing that
ever touches /root/reference, and nothing under tests/ with ``-m gpu`` reads it.
"""
import os
import subprocess
import sys
import tempfile
import types

import numpy as np

REF = os.environ.get("PYRAPOSE_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def _quat2mat(q):
    # transforms3d.quaternions.quat2mat (transforms3d 0.3.1), published algorithm.
    w, x, y, z = q
    Nq = w * w + x * x + y * y + z * z
    if Nq < np.finfo(np.float64).eps:
        return np.eye(3)
    s = 2.0 / Nq
    X = x * s
    Y = y * s
    Z = z * s
    wX = w * X; wY = w * Y; wZ = w * Z
    xX = x * X; xY = x * Y; xZ = x * Z
    yY = y * Y; yZ = y * Z; zZ = z * Z
    return np.array(
        [[1.0 - (yY + zZ), xY - wZ, xZ + wY],
         [xY + wZ, 1.0 - (xX + zZ), yZ - wX],
         [xZ - wY, yZ + wX, 1.0 - (xX + yY)]])


def import_reference():
    tmp = tempfile.mkdtemp(prefix="pyrapose_ref_")
    pkg = os.path.join(tmp, "PyraPose")
    utils = os.path.join(pkg, "utils")
    os.makedirs(utils)
    open(os.path.join(pkg, "__init__.py"), "w").close()
    open(os.path.join(utils, "__init__.py"), "w").close()
    os.symlink(os.path.join(REF, "PyraPose/utils/anchors.py"), os.path.join(utils, "anchors.py"))
    # cythonize the reference .pyx in place-less mode: C file and .so land in the temp dir
    pyx = os.path.join(REF, "PyraPose/utils/compute_overlap.pyx")
    c_out = os.path.join(utils, "compute_overlap.c")
    subprocess.check_call([sys.executable, "-m", "cython", "-3", pyx, "-o", c_out])
    import sysconfig
    inc = sysconfig.get_paths()["include"]
    so = os.path.join(utils, "compute_overlap" + sysconfig.get_config_var("EXT_SUFFIX"))
    subprocess.check_call(
        ["gcc", "-O2", "-shared", "-fPIC", "-w", "-I", inc, "-I", np.get_include(), c_out, "-o", so])

    keras = types.ModuleType("keras")
    keras.backend = types.ModuleType("keras.backend")
    keras.backend.floatx = lambda: "float32"
    sys.modules["keras"] = keras
    sys.modules["keras.backend"] = keras.backend
    sys.modules["cv2"] = types.ModuleType("cv2")
    t3d = types.ModuleType("transforms3d")
    t3d.quaternions = types.ModuleType("transforms3d.quaternions")
    t3d.quaternions.quat2mat = _quat2mat
    sys.modules["transforms3d"] = t3d
    sys.modules["transforms3d.quaternions"] = t3d.quaternions
    sys.path.insert(0, tmp)
    import PyraPose.utils.anchors as ref_anchors
    from PyraPose.utils.compute_overlap import compute_overlap
    return ref_anchors, compute_overlap


# LineMOD intrinsics (reference preprocessing/generator.py:246)
FX, FY, CX, CY = 572.4114, 573.57043, 325.2611, 242.04899


def synth_annotations(rng, H, W, K, num_classes, general_rot=False, z_mm=800.0, lo=40, hi=160):
    """Seeded synthetic annotation dict in the reference generator's format
    (preprocessing/linemod.py:225: mask, labels, bboxes, poses, segmentations, cam_params, mask_ids)."""
    mask = np.zeros((H, W), dtype=np.uint8)
    ann = {'mask': [mask], 'labels': np.empty((0,)), 'bboxes': np.empty((0, 4)), 'poses': np.empty((0, 7)),
           'segmentations': np.empty((0, 8, 3)), 'cam_params': np.empty((0, 4)), 'mask_ids': np.empty((0,))}
    for k in range(K):
        w = rng.uniform(lo, hi); h = rng.uniform(lo, hi)
        x1 = rng.uniform(0, W - w); y1 = rng.uniform(0, H - h)
        mid = k + 1
        mask[int(y1):int(y1 + h), int(x1):int(x1 + w)] = mid
        # metric cuboid whose projection roughly covers the bbox
        sx = w * z_mm / FX; sy = h * z_mm / FY; sz = rng.uniform(40, 120)
        xm, ym, zm = -sx / 2, -sy / 2, -sz / 2
        xp, yp, zp = sx / 2, sy / 2, sz / 2
        box = np.array([[xp, yp, zp], [xp, yp, zm], [xp, ym, zm], [xp, ym, zp],
                        [xm, yp, zp], [xm, yp, zm], [xm, ym, zm], [xm, ym, zp]], dtype=np.float32)
        tx = ((x1 + w / 2) - CX) * z_mm / FX; ty = ((y1 + h / 2) - CY) * z_mm / FY
        if general_rot:
            q = rng.normal(size=4); q /= np.linalg.norm(q)
        else:
            q = np.array([1.0, 0.0, 0.0, 0.0])
        ann['labels'] = np.concatenate([ann['labels'], [float(rng.integers(0, num_classes))]])
        ann['bboxes'] = np.concatenate([ann['bboxes'], [[x1, y1, x1 + w, y1 + h]]])
        ann['poses'] = np.concatenate([ann['poses'], [[tx, ty, z_mm, q[0], q[1], q[2], q[3]]]])
        ann['segmentations'] = np.concatenate([ann['segmentations'], [box]])
        ann['cam_params'] = np.concatenate([ann['cam_params'], [[FX, FY, CX, CY]]])
        ann['mask_ids'] = np.concatenate([ann['mask_ids'], [float(mid)]])
    return ann


def pack_annotations(prefix, anns, out):
    for i, a in enumerate(anns):
        for key in ('labels', 'bboxes', 'poses', 'segmentations', 'cam_params', 'mask_ids'):
            out[f"{prefix}_ann{i}_{key}"] = np.asarray(a[key])
        out[f"{prefix}_ann{i}_mask"] = a['mask'][0]


def main():
    ra, compute_overlap = import_reference()
    from PIL import Image
    out = {}

    # ---- T1: base anchors (anchors.py:447-478) ----
    P = ra.AnchorParameters.default
    out["default_ratios"] = np.asarray(P.ratios)
    out["default_scales"] = np.asarray(P.scales)
    for size in (32, 64, 128, 48, 96, 192, 24, 160):
        out[f"base_anchors_{size}"] = ra.generate_anchors(size, P.ratios, P.scales)
    scales4 = np.array([2 ** 0, 2 ** (1.0 / 3.0), 2 ** (2.0 / 3.0), 2 ** 1], "float32")
    out["scales4"] = scales4
    out["base_anchors_48_s4"] = ra.generate_anchors(48, P.ratios, scales4)

    # ---- T2: level shapes and full grids (anchors.py:357-412) ----
    for (H, W) in ((480, 640), (540, 720), (97, 131)):
        out[f"shapes_{H}x{W}"] = np.array(ra.guess_shapes((H, W), [3, 4, 5]))
        out[f"anchors_{H}x{W}"] = ra.anchors_for_shape((H, W))
    out["shapes_480x640_p37"] = np.array(ra.guess_shapes((480, 640), [3, 4, 5, 6, 7]))

    # ---- T3/T4: IoU + gt assignment (compute_overlap.pyx:13-53, anchors.py:290-318) ----
    rng = np.random.default_rng(0)
    anchors = out["anchors_480x640"]
    for name, K in (("k1", 1), ("k5", 5)):
        a = synth_annotations(rng, 480, 640, K, 13)
        q = a['bboxes'].astype(np.float64)
        ov = compute_overlap(anchors, q)
        pos, ign, amax = ra.compute_gt_annotations(anchors, a['bboxes'])
        out[f"iou_{name}_query"] = q
        out[f"iou_{name}_overlaps"] = ov
        out[f"iou_{name}_positive"] = pos
        out[f"iou_{name}_ignore"] = ign
        out[f"iou_{name}_argmax"] = amax.astype(np.int64)
    # edge cases on a small anchor set: identical boxes (IoU 1 under +1 convention), touching
    # boxes (iw == 0 -> stays 0), degenerate zero-area queries, ties in argmax (first max wins)
    small = anchors[::997].copy()
    q = np.vstack([small[3], small[3], small[10] + [0.25, 0, 0.25, 0],
                   [small[5][2] + 1 - 1, small[5][1], small[5][2] + 50, small[5][3]],   # overlaps by exactly 1 px col
                   [small[5][2] + 1, small[5][1], small[5][2] + 50, small[5][3]],       # iw == 0
                   [100.0, 100.0, 100.0, 100.0],                                        # zero-size box (+1 -> area 1)
                   [200.0, 200.0, 150.0, 150.0]])                                       # inverted box
    out["iou_edge_boxes"] = small
    out["iou_edge_query"] = q
    out["iou_edge_overlaps"] = compute_overlap(small, q)
    pos, ign, amax = ra.compute_gt_annotations(small, q)
    out["iou_edge_positive"] = pos; out["iou_edge_ignore"] = ign; out["iou_edge_argmax"] = amax.astype(np.int64)

    # ---- box3D_transform directly (anchors.py:515-559) ----
    gt16 = rng.uniform(0, 640, size=(anchors.shape[0] // 100 + 1, 16))
    sub = anchors[::100]
    out["b3d_anchors"] = sub
    out["b3d_gt"] = gt16[: sub.shape[0]]
    out["b3d_targets"] = ra.box3D_transform(sub, gt16[: sub.shape[0]], 13)

    # ---- PIL NEAREST index maps (anchors.py:158) ----
    for (H, W) in ((480, 640), (540, 720), (97, 131), (333, 517)):
        sh = ra.guess_shapes((H, W), [3])[0]
        ramp_y = np.repeat(np.arange(H, dtype=np.int32)[:, None], W, axis=1)
        ramp_x = np.repeat(np.arange(W, dtype=np.int32)[None, :], H, axis=0)
        ry = np.asarray(Image.fromarray(ramp_y).resize((sh[1], sh[0]), Image.NEAREST))
        rx = np.asarray(Image.fromarray(ramp_x).resize((sh[1], sh[0]), Image.NEAREST))
        out[f"pil_nearest_{H}x{W}_rows"] = ry[:, 0].astype(np.int32)
        out[f"pil_nearest_{H}x{W}_cols"] = rx[0, :].astype(np.int32)
        assert (ry == ry[:, :1]).all() and (rx == rx[:1, :]).all()

    # ---- T5: full target assignment (anchors.py:72-287) ----
    import hashlib

    def run_targets(prefix, H, W, C, shapes_and_K, general_rot, full=False, lo=40, hi=160):
        anns, images = [], []
        for (h, w, K) in shapes_and_K:
            # images smaller than the padded batch exercise anchors.py:279-285
            anns.append(synth_annotations(rng, h, w, K, C, general_rot, lo=lo, hi=hi))
            images.append(np.zeros((h, w, 3), np.float32))
        anc = ra.anchors_for_shape((H, W))
        # the reference computes mask level shapes from image_group[0] (anchors.py:114): keep image 0 full size
        reg, lab, msk = ra.anchor_targets_bbox(anc, images, anns, C)
        pack_annotations(prefix, anns, out)
        out[f"{prefix}_image_shapes"] = np.array([im.shape[:2] for im in images])
        out[f"{prefix}_meta"] = np.array([H, W, C])
        out[f"{prefix}_labels"] = lab          # mostly zeros: compresses to a few KB
        out[f"{prefix}_mask"] = msk
        out[f"{prefix}_reg_state"] = reg[:, :, -1].astype(np.int8)
        out[f"{prefix}_reg_sha256"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(reg).tobytes()).digest(), np.uint8)
        if full:
            out[f"{prefix}_regression"] = reg
        else:
            # dense regression rows do not compress; keep every non-background row plus a stride sample
            keep = (reg[:, :, -1] != 0)
            keep[:, ::53] = True
            idx = np.argwhere(keep)
            out[f"{prefix}_reg_rows_idx"] = idx.astype(np.int32)
            out[f"{prefix}_reg_rows"] = reg[idx[:, 0], idx[:, 1], :]

    run_targets("tgt_small_identity", 160, 224, 13, [(160, 224, 2), (160, 224, 0), (120, 200, 2)], False, full=True, lo=30, hi=90)
    run_targets("tgt_small_general", 160, 224, 13, [(160, 224, 3)], True, full=True, lo=30, hi=90)
    run_targets("tgt_identity", 480, 640, 13, [(480, 640, 2), (480, 640, 0), (400, 600, 3)], False)
    run_targets("tgt_general", 480, 640, 13, [(480, 640, 3), (480, 640, 1)], True)
    run_targets("tgt_tless_identity", 540, 720, 30, [(540, 720, 8)], False)

    path = os.path.join(HERE, "anchors_targets.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) / 1e6, "MB,", len(out), "arrays")


if __name__ == "__main__":
    main()
