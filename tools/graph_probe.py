"""Does capturing the forward plan in a HIP graph shorten it?  (eager launches vs graph replay, same kernels)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyrapose_amd import ops  # noqa: E402
from pyrapose_amd.engine import Engine  # noqa: E402

ctx = ops.Context(0)
eng = Engine(ctx, 13, 8, 480, 640, train=False)
x = torch.randn((8, 480, 640, 3), device="cuda") * 50
eng.x_in.copy_(x)
ev = lambda: torch.cuda.Event(enable_timing=True)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = ev(), ev()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


t_eager = timeit(lambda: eng.forward())
# capture on a side stream (torch requirement); the engine's lane 0 is the legacy default stream, so point it at the capture stream
cap = torch.cuda.Stream()
eng.streams[0] = cap
eng.ctxs[0].use_stream(cap)
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(cap):
    eng.forward()
    torch.cuda.synchronize()
    g.capture_begin()
    eng.forward()
    g.capture_end()
torch.cuda.synchronize()
t_graph = timeit(lambda: g.replay())
print("forward eager %.3f ms   graph replay %.3f ms" % (t_eager, t_graph))
