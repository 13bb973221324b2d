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
    # everything below stays on the device until the poses come back: one gather for all (image, class) problems
    sel = cnt >= max(int(min_votes), 1)                                        # [B, C]
    if not bool(sel.any()):
        return []
    cap = idx.shape[2]
    live = (torch.arange(cap, device="cuda")[None, None, :] < cnt[:, :, None]) & sel[:, :, None]  # [B, C, cap], (b, c, vote) order
    b_of, c_of, _ = torch.nonzero(live, as_tuple=True)
    anchor = idx[live].long()
    img = boxes3D[b_of, anchor, :].double().reshape(-1, 2)
    obj = corners[c_of].reshape(-1, 3)
    k = cnt[sel].long()                                                        # votes per problem, (b, c) order
    offs = torch.zeros((k.numel() + 1,), dtype=torch.int32, device="cuda")
    offs[1:] = (8 * torch.cumsum(k, 0)).to(torch.int32)
    pb, pc = torch.nonzero(sel, as_tuple=True)
    K_all = torch.as_tensor(np.stack([[Kb[0, 0], Kb[1, 1], Kb[0, 2], Kb[1, 2]] for Kb in Ks]), dtype=torch.float64, device="cuda")
    R, t, n_in, mask, ok = ops.pnp_ransac(ctx, offs, obj.contiguous(), img.contiguous(), K_all[pb].contiguous(), iterations, reproj_error, seed, 8)
    R, t, mask, ok = R.cpu().numpy(), t.cpu().numpy(), mask.cpu().numpy(), ok.cpu().numpy()
    offs_h, anchor_h, k_h = offs.cpu().numpy(), anchor.cpu().numpy(), k.cpu().numpy()
    pb, pc = pb.cpu().numpy(), pc.cpu().numpy()
    out, v0 = [], 0
    for p in range(len(k_h)):
        out.append(dict(image=int(pb[p]), cls=int(pc[p]), votes=anchor_h[v0: v0 + k_h[p]], ok=bool(ok[p]), R=R[p], t=t[p],
                        inliers=np.nonzero(mask[offs_h[p]:offs_h[p + 1]])[0]))
        v0 += int(k_h[p])
    return out
