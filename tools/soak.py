"""Soak run: N training steps at the bench configuration with NEW synthetic images + annotations every step (lean feed:
uint8 images + raw annotations, device preprocessing / target assignment, sparse 3D-box backward with a different support
each step).  Prints the loss trajectory and the rate; fails on a non-finite loss."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pyrapose_amd import arch  # noqa: E402
from pyrapose_amd.engine import Engine  # noqa: E402
from pyrapose_amd.runtime import default_context  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=120)
ap.add_argument("--lr", type=float, default=1e-4)
ap.add_argument("--augment", action="store_true", help="a fresh random affine transform per image and step, applied on the device "
                "(image: bilinear warp, id mask: nearest warp) -- SURVEY 8f3")
args = ap.parse_args()
B, H, W, C = 8, 480, 640, 13
eng = Engine(default_context(), C, B, H, W, weights=arch.init_weights(C, seed=0), train=True, lr=args.lr)
rng = np.random.default_rng(0)
batches = []
for i in range(8):  # eight different batches, cycled
    _, images, anns = bench.synth_batch(B, H, W, C, seed=100 + i)
    batches.append((torch.from_numpy(rng.integers(0, 256, (B, H, W, 3)).astype(np.uint8)).pin_memory(), anns))
hist = []
torch.cuda.synchronize()
t0 = time.perf_counter()
for s in range(args.steps):
    u8, anns = batches[s % len(batches)]
    tf = None
    if args.augment:  # utils/transform.py: scaling 0.9 .. 1.1, translation +-10 % about the image centre (bin/train.py:189-199)
        tf = []
        for _ in range(B):
            sc, t = rng.uniform(0.9, 1.1), rng.uniform(-0.1, 0.1, 2) * np.array([W, H])
            c = np.array([0.5 * W, 0.5 * H])
            tf.append(np.array([[sc, 0, c[0] - sc * c[0] + t[0]], [0, sc, c[1] - sc * c[1] + t[1]], [0, 0, 1.0]]))
    eng.train_step_from_annotations(u8, anns, transforms=tf)
    if s % 10 == 0 or s + 1 == args.steps:
        l = eng.losses()
        assert all(np.isfinite(v) for v in l.values()), (s, l)
        hist.append((s, round(l["total"], 4), round(l["3Dbox"], 4), round(l["cls"], 4), round(l["mask"], 4)))
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("steps %d in %.2f s = %.1f images/s (losses read every 10 steps)" % (args.steps, dt, args.steps * B / dt))
for h in hist:
    print("step %4d  total %.4f  3Dbox %.4f  cls %.4f  mask %.4f" % h)
