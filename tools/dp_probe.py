"""Rehearsal probe for the data-parallel step (several ranks may share one GPU): wall time per phase with host syncs.
   PP_DIST_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/dp_probe.py"""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyrapose_amd import ops  # noqa: E402
from pyrapose_amd.engine import Engine  # noqa: E402
from pyrapose_amd.parallel import DataParallel  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = int(os.environ["LOCAL_RANK"]) % torch.cuda.device_count()
torch.cuda.set_device(dev)
backend = os.environ.get("PP_DIST_BACKEND", "nccl")
dist.init_process_group(backend) if backend != "nccl" else dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
ctx = ops.Context(dev)
eng = Engine(ctx, 13, 2, 480, 640)
dp = DataParallel(eng)
eng.y_cls[..., -1] = 1; eng.y_box[..., -1] = 0; eng.y_mask[..., -1] = 1
sync = torch.cuda.synchronize
for it in range(4):
    sync(); dist.barrier(); t0 = time.perf_counter()
    eng.forward(); sync(); t1 = time.perf_counter()
    eng.loss_and_backward(); sync(); t2 = time.perf_counter()
    dp.finish(); sync(); t3 = time.perf_counter()
    eng.optimizer_step(); sync(); t4 = time.perf_counter()
    if rank == 0:
        print("it %d lanes %d: fwd %.1f  bwd(+launch allreduce) %.1f  finish %.1f  opt %.1f ms; streams %s" % (
            it, eng.n_lanes, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3,
            [hex(s.cuda_stream) for s in eng.streams]), flush=True)
dist.destroy_process_group()
