"""Wall time of the three phases of a train step (forward, losses + backward, optimizer), HIP events on lane 0."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyrapose_amd import ops  # noqa: E402
from pyrapose_amd.engine import Engine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--bench-targets", action="store_true", help="the synthetic annotations of bench.py (SURVEY 8d config 2) instead of a dense positive pattern")
    args = ap.parse_args()
    ctx = ops.Context(0)
    eng = Engine(ctx, 13, args.batch, 480, 640)
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.rand((args.batch, 480, 640, 3), device="cuda", generator=g) * 255 - 110
    eng.y_cls[..., -1] = 1
    eng.y_cls[:, ::50, 3] = 1
    eng.y_box[..., -1] = 0
    eng.y_box[:, ::50, -1] = 1
    eng.y_mask[..., -1] = 1
    eng.y_mask[:, ::7, 2] = 1
    if args.bench_targets:
        import bench
        from pyrapose_amd.utils import anchors as UA
        _, images, anns = bench.synth_batch(args.batch, 480, 640, 13, seed=1000)
        eng.set_targets(*UA.anchor_targets_bbox_device(UA.anchors_for_shape_device((480, 640)), images, anns, 13))
    ev = lambda: torch.cuda.Event(enable_timing=True)
    acc = np.zeros(3)
    for it in range(args.steps + 2):
        e = [ev() for _ in range(4)]
        e[0].record()
        eng.forward(x)
        e[1].record()
        eng.loss_and_backward()
        e[2].record()
        eng.optimizer_step()
        e[3].record()
        torch.cuda.synchronize()
        if it >= 2:
            acc += [e[i].elapsed_time(e[i + 1]) for i in range(3)]
    acc /= args.steps
    print("fwd %.2f ms  bwd %.2f ms  opt %.2f ms  total %.2f ms  (%.1f img/s)" % (acc[0], acc[1], acc[2], acc.sum(), args.batch * 1e3 / acc.sum()))


if __name__ == "__main__":
    main()
