"""PCIe-inclusive rates: the same work as bench.py / tools/bench_infer.py, but every step starts from
HOST (numpy) inputs and, for inference, ends with HOST outputs -- what a caller of the reference-shaped API pays
(model.train_on_batch / predict_on_batch with numpy arrays).  Pinned staging buffers, no overlap tricks."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyrapose_amd import ops  # noqa: E402
from pyrapose_amd.engine import Engine  # noqa: E402


def main():
    B, C, H, W = 8, 13, 480, 640
    ctx = ops.Context(0)
    eng = Engine(ctx, C, B, H, W)
    rng = np.random.default_rng(0)
    x = (rng.integers(0, 256, (B, H, W, 3)).astype(np.float32) - 110.0)
    y_box = np.zeros((B, eng.N, 17), np.float32)
    y_cls = np.zeros((B, eng.N, C + 1), np.float32); y_cls[:, ::50, 3] = 1; y_cls[:, ::50, -1] = 1
    y_mask = np.zeros((B, eng.M3, C + 1), np.float32); y_mask[:, ::7, 2] = 1; y_mask[:, ::7, -1] = 1
    y_box[:, ::50, -1] = 1
    host = [torch.from_numpy(a).pin_memory() for a in (x, y_box, y_cls, y_mask)]

    def train_host():
        dev = [t.cuda(non_blocking=True) for t in host]
        eng.train_step(dev[0], dev[1:])

    def train_resident():
        eng.train_step()

    def timed(fn, n=10):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    t_res, t_host = timed(train_resident), timed(train_host)
    # the same host batches through the background uploader (pyrapose_amd/prefetch.py): numpy in, 13 steps
    from pyrapose_amd.prefetch import DevicePrefetcher
    npb = (x, [y_box, y_cls, y_mask])

    def prefetched(n):
        for xd, yd in DevicePrefetcher(lambda i: npb, n, depth=3):
            eng.train_step(xd, list(yd))
    prefetched(3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    prefetched(13)
    torch.cuda.synchronize()
    t_pref = (time.perf_counter() - t0) / 13
    # the lean feed: uint8 images + raw annotations cross PCIe, mean subtraction / packing and target assignment run on the device
    import bench as _bench
    from pyrapose_amd.utils import anchors as UA
    _, images, anns = _bench.synth_batch(B, H, W, C, seed=5)
    anchors = UA.anchors_for_shape_device((H, W))
    x_u8 = torch.from_numpy(rng.integers(0, 256, (B, H, W, 3)).astype(np.uint8)).pin_memory()

    def train_lean():
        xd = x_u8.cuda(non_blocking=True)
        eng.set_targets(*UA.anchor_targets_bbox_device(anchors, images, anns, C))
        eng.forward_u8(xd)
        eng.loss_and_backward()
        eng.optimizer_step()

    t_lean = timed(train_lean)
    h2d_mb = sum(t.numel() * 4 for t in host) / 1e6
    inf = Engine(ctx, C, B, H, W, train=False)
    xin = host[0]

    def infer_host():
        b, c, m = inf.predict_on_batch(xin.cuda(non_blocking=True))
        return b.cpu(), c.cpu(), m.cpu()

    def infer_resident():
        inf.predict_on_batch(None)

    inf.x_in.copy_(xin.cuda())
    ti_res, ti_host = timed(infer_resident), timed(infer_host)
    d2h_mb = (B * inf.N * 16 + B * inf.N * C + B * inf.M3 * C) * 4 / 1e6
    print(json.dumps({"train_images_per_sec_resident": B / t_res, "train_images_per_sec_host_inputs": B / t_host, "train_images_per_sec_host_inputs_prefetched": B / t_pref,
                      "train_images_per_sec_u8_images_plus_annotations": B / t_lean, "train_lean_h2d_MB_per_step": x_u8.numel() / 1e6, "train_h2d_MB_per_step": h2d_mb,
                      "infer_images_per_sec_resident": B / ti_res, "infer_images_per_sec_host_in_out": B / ti_host,
                      "infer_h2d_MB": host[0].numel() * 4 / 1e6, "infer_d2h_MB": d2h_mb, "batch": B}))


if __name__ == "__main__":
    main()
