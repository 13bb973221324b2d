import sys, time
sys.path.insert(0, '/root/repo')
import torch
from pyrapose_amd import ops
from pyrapose_amd.engine import Engine
ctx = ops.Context(0)
eng = Engine(ctx, 13, 8, 480, 640)
eng.y_cls[..., -1] = 1; eng.y_box[..., -1] = 0; eng.y_mask[..., -1] = 1
for _ in range(3): eng.train_step()
torch.cuda.synchronize()
t0 = time.perf_counter(); eng.train_step(); t1 = time.perf_counter(); eng.train_step(); t2 = time.perf_counter()
torch.cuda.synchronize(); t3 = time.perf_counter()
print("host time per step: %.2f ms, %.2f ms; drain %.2f ms" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3))
