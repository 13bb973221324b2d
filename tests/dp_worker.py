"""Helper of tests/test_gpu_model.py::test_two_rank_data_parallel_equals_global_batch: one data-parallel rank.
usage: python tests/dp_worker.py RANK WORLD PORT OUTDIR   (gloo backend: the ranks share the one GPU of the test box)"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def global_batch():
    from tests.test_gpu_model import random_targets, synth_input
    from pyrapose_amd import arch
    B, H, W, C = 4, 64, 96, 5
    rng = np.random.default_rng(21)
    Wt = arch.init_weights(C, seed=22)
    x = synth_input(rng, B, H, W)
    N = sum(((H + 2 ** l - 1) // 2 ** l) * ((W + 2 ** l - 1) // 2 ** l) for l in (3, 4, 5)) * 9
    M3 = ((H + 7) // 8) * ((W + 7) // 8)
    tg = random_targets(rng, B, N, M3, C, pos_frac=0.05)
    return B, H, W, C, Wt, x, tg


def main():
    rank, world, port, outdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", port
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pyrapose_amd.engine import Engine
    from pyrapose_amd.parallel import DataParallel
    from pyrapose_amd.runtime import default_context
    B, H, W, C, Wt, x, tg = global_batch()
    per = B // world
    sl = slice(rank * per, (rank + 1) * per)
    eng = Engine(default_context(), C, per, H, W, weights=Wt, train=True)
    assert (eng.N, eng.M3) == (tg[0].shape[1], tg[2].shape[1])
    DataParallel(eng, bucket_bytes=8 << 20)
    eng.train_step(torch.from_numpy(x[sl]).cuda(), [torch.from_numpy(a[sl]).cuda() for a in tg])
    torch.cuda.synchronize()
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), grad=eng.params.grad.cpu().numpy(), w=eng.params.w_master.cpu().numpy(),
             counts=eng.counts.cpu().numpy(), losses=np.array([eng.losses()[k] for k in ("3Dbox", "cls", "mask")]))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
