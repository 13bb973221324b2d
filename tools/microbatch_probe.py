"""Probe: does the GPU run two concurrent half-batch training pipelines (2 engines x 4 images, own streams) faster than one
engine with 8 images?  Throughput only (the two engines do not share weights here).  --stagger starts the second pipeline's
step half a step later (its trunk then overlaps the first one's heads)."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pyrapose_amd import arch, ops  # noqa: E402
from pyrapose_amd.engine import Engine  # noqa: E402
from pyrapose_amd.utils import anchors as UA  # noqa: E402


def make(B, seed, stream):
    with torch.cuda.stream(stream):
        ctx = ops.Context(0, stream)
        eng = Engine(ctx, 13, B, 480, 640, weights=W, train=True)
        x, images, anns = bench.synth_batch(B, 480, 640, 13, seed=seed)
        eng.set_targets(*UA.anchor_targets_bbox_device(UA.anchors_for_shape_device((480, 640)), images, anns, 13))
        eng.x_in.copy_(torch.from_numpy(x).cuda())
    return eng


ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--pipes", type=int, default=2)
ap.add_argument("--batch", type=int, default=4)
args = ap.parse_args()
W = arch.init_weights(13, seed=0)
streams = [torch.cuda.Stream() for _ in range(args.pipes)]
engs = [make(args.batch, 1000 + i, s) for i, s in enumerate(streams)]
torch.cuda.synchronize()


def run(steps):
    for _ in range(steps):
        # interleave the phases of the pipelines on the host so that every stream has work queued
        for e, s in zip(engs, streams):
            with torch.cuda.stream(s):
                e.forward()
        for e, s in zip(engs, streams):
            with torch.cuda.stream(s):
                e.loss_and_backward()
                e.optimizer_step()


run(3)
torch.cuda.synchronize()
t0 = time.perf_counter()
run(args.steps)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("%d pipelines x batch %d: %.1f images/s (%.2f ms per round of %d images)" %
      (args.pipes, args.batch, args.steps * args.pipes * args.batch / dt, 1e3 * dt / args.steps, args.pipes * args.batch))
