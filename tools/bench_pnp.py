#!/usr/bin/env python3
"""Throughput of the device RANSAC-PnP (pp_pnp_ransac_f64) on the reference's evaluation workload: per image up to C classes,
each with k corner votes (utils/linemod_eval.py:421-431; 300 iterations, 5 px).  Prints problems/s and votes/s.
Usage: python3 tools/bench_pnp.py [--problems 104] [--votes 400] [--iters 10]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyrapose_amd import ops  # noqa: E402
from pyrapose_amd.runtime import default_context  # noqa: E402
from tests.test_oracle_pnp import K4, make_votes, rot_err_deg  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--problems", type=int, default=104)  # 8 images x 13 classes
    ap.add_argument("--votes", type=int, default=400)
    ap.add_argument("--iters", type=int, default=10)
    args = ap.parse_args()
    ctx = default_context()
    rng = np.random.default_rng(0)
    objs, imgs, offs, truth = [], [], [0], []
    for _ in range(args.problems):
        R, t, obj, img, clean = make_votes(rng, args.votes, 1.5, 0.3)
        objs.append(obj); imgs.append(img); offs.append(offs[-1] + len(obj)); truth.append((R, t))
    dev = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dt).cuda()
    offsets, obj, img = dev(np.asarray(offs, np.int32), torch.int32), dev(np.concatenate(objs), torch.float64), dev(np.concatenate(imgs), torch.float64)
    K = dev(np.tile(np.asarray(K4), (args.problems, 1)), torch.float64)
    out = ops.pnp_ransac(ctx, offsets, obj, img, K, 300, 5.0, 1, 8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        out = ops.pnp_ransac(ctx, offsets, obj, img, K, 300, 5.0, 1, 8)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.iters
    R = out[0].cpu().numpy()
    worst = max(rot_err_deg(R[p], truth[p][0]) for p in range(args.problems))
    print("pnp_ransac: %d problems x %d votes (x8 corners), 300 iterations: %.2f ms per batch = %.0f problems/s, %.2f M votes/s; "
          "all ok %s, worst rotation error %.2f deg" % (args.problems, args.votes, dt * 1e3, args.problems / dt,
                                                       args.problems * args.votes / dt / 1e6, bool(out[4].all()), worst))


if __name__ == "__main__":
    main()
