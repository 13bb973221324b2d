"""debug: per-phase wall times of the 2-rank gloo rehearsal (forward / loss+backward+allreduce / finish / optimizer), synchronised"""
import os, sys, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "RANK" not in os.environ:
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29544")
        procs.append(subprocess.Popen([sys.executable, __file__], env=env))
    sys.exit(max(p.wait() for p in procs))
sys.path.insert(0, ROOT)
import numpy as np, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("gloo")
import bench
from pyrapose_amd import arch
from pyrapose_amd.engine import Engine
from pyrapose_amd.parallel import DataParallel
from pyrapose_amd.runtime import default_context
from pyrapose_amd.utils import anchors as UA
B, H, W, C = 8, 480, 640, 13
ctx = default_context(0)
x, images, anns = bench.synth_batch(B, H, W, C, seed=1000 + dist.get_rank())
tg = UA.anchor_targets_bbox_device(UA.anchors_for_shape_device((H, W)), images, anns, C)
eng = Engine(ctx, C, B, H, W, weights=arch.init_weights(C, seed=0), train=True)
dp = DataParallel(eng)
eng.set_targets(*tg)
eng.x_in.copy_(torch.from_numpy(x).cuda())
def T():
    torch.cuda.synchronize(); return time.perf_counter()
for step in range(5):
    t0 = T(); eng.forward(); t1 = T(); eng.loss_and_backward(); t2 = T(); dp.finish(); t3 = T(); eng.optimizer_step(); t4 = T()
    if dist.get_rank() == 0:
        print("step %d: fwd %.1f  loss+bwd %.1f  finish %.1f  opt %.1f ms" % (step, 1e3*(t1-t0), 1e3*(t2-t1), 1e3*(t3-t2), 1e3*(t4-t3)), flush=True)
t0 = T()
for step in range(5):
    eng.train_step()
t1 = T()
if dist.get_rank() == 0: print("train_step x5: %.1f ms per step" % (1e3 * (t1 - t0) / 5))
dist.destroy_process_group()
