"""From the prediction model's outputs to object poses: the per-image, per-class block of the evaluation loops
(utils/linemod_eval.py:303-333 threshold + vote count, :421-431 correspondences, :479-485 PnP), batched on the device.

    for inv_cls in range(scores.shape[2]):
        cls_indices = np.where(scores[0, :, inv_cls] > threshold)        # ascending anchor order (D3)
        if len(cls_indices[0]) < 10: continue                            # < 1 in occlusion_eval.py:359-371
        est_points = boxes3D[0, cls_indices, :].reshape(k * 8, 1, 2);  obj_points = repeat(threeD_boxes[cls], k)
        retval, rvec, tvec, inliers = cv2.solvePnPRansac(obj_points, est_points, K, None, 300, 5.0, 0.99, ITERATIVE)
"""
import numpy as np
import torch

from .. import ops
from ..runtime import default_context


def poses_from_outputs(boxes3D, scores, threeD_boxes, K, threshold=0.5, min_votes=10, iterations=300, reproj_error=5.0, seed=0,
                       ctx=None):
    """boxes3D [B,N,16], scores [B,N,C] (numpy or cuda float32 tensors: predict_on_batch outputs); threeD_boxes [C,8,3]
    cuboid corners per class (model units); K 3x3 or [B,3,3].  Returns one dict per (image, class) that reached
    `min_votes` votes, image-major then class ascending like the reference loop:
    {image, cls (0-based), votes (ascending anchor indices), ok, R [3,3], t [3], inliers (indices into votes x 8 corners)}."""
    ctx = ctx or default_context()
    dev = lambda a: a.cuda() if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
    boxes3D, scores = dev(boxes3D).float().contiguous(), dev(scores).float().contiguous()
    B, N, C = scores.shape
    corners = torch.as_tensor(np.asarray(threeD_boxes, np.float64).reshape(C, 8, 3), device="cuda")
    K = np.asarray(K, np.float64)
    Ks = np.broadcast_to(K.reshape(-1, 3, 3), (B, 3, 3)) if K.ndim == 3 or K.size == 9 else None
    if Ks is None:
        raise ValueError("K must be 3x3 or [B,3,3]")
    idx, cnt = ops.score_threshold_compact(ctx, scores, float(threshold))  # bit-exact np.where order
    cnt_h = cnt.cpu().numpy()
    sel = [(b, c) for b in range(B) for c in range(C) if cnt_h[b, c] >= max(int(min_votes), 1)]
    if not sel:
        return []
    offs, objs, imgs, K4, votes = [0], [], [], [], []
    for b, c in sel:
        k = int(cnt_h[b, c])
        v = idx[b, c, :k].long()
        votes.append(v)
        imgs.append(boxes3D[b, v, :].double().reshape(k * 8, 2))
        objs.append(corners[c].repeat(k, 1))
        K4.append([Ks[b][0, 0], Ks[b][1, 1], Ks[b][0, 2], Ks[b][1, 2]])
        offs.append(offs[-1] + 8 * k)
    R, t, n_in, mask, ok = ops.pnp_ransac(ctx, torch.tensor(offs, dtype=torch.int32, device="cuda"), torch.cat(objs), torch.cat(imgs),
                                          torch.tensor(K4, dtype=torch.float64, device="cuda"), iterations, reproj_error, seed, 8)
    R, t, mask, ok = R.cpu().numpy(), t.cpu().numpy(), mask.cpu().numpy(), ok.cpu().numpy()
    return [dict(image=b, cls=c, votes=votes[p].cpu().numpy(), ok=bool(ok[p]), R=R[p], t=t[p],
                 inliers=np.nonzero(mask[offs[p]:offs[p + 1]])[0]) for p, (b, c) in enumerate(sel)]
