"""Where the time of device-side target assignment goes (utils.anchors.anchor_targets_bbox_device, one batch of 8)."""
import os, sys, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pyrapose_amd import ops
from pyrapose_amd.runtime import default_context
from pyrapose_amd.utils import anchors as UA
B, H, W, C = 8, 480, 640, 13
x, images, anns = bench.synth_batch(B, H, W, C, seed=1000)
anchors = UA.anchors_for_shape_device((H, W))
def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("whole call            %.2f ms" % timed(lambda: UA.anchor_targets_bbox_device(anchors, images, anns, C)))
print("pack_annotations      %.2f ms" % timed(lambda: UA.pack_annotations(anns)))
def planes():
    ms = [np.asarray(a["mask"][0]).astype(np.uint8) for a in anns]
    plane = np.zeros((len(ms), H, W), np.uint8)
    for i, m in enumerate(ms): plane[i, :m.shape[0], :m.shape[1]] = m
    return plane
print("mask planes (host)    %.2f ms" % timed(planes))
pl = planes()
print("mask upload           %.2f ms" % timed(lambda: torch.from_numpy(pl).cuda()))
offs, boxes, labels, box3d, mids = UA.pack_annotations(anns)
dev = lambda a: torch.from_numpy(a).cuda()
args = (dev(boxes), dev(labels), dev(box3d), dev(mids), torch.from_numpy(pl).cuda())
ctx = default_context()
mh, mw = 60, 80
print("kernel + output alloc %.2f ms" % timed(lambda: ops.anchor_targets(ctx, anchors, offs, *args, [(H, W)] * B, [(H, W)] * B, C, mh, mw)))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(10): UA.anchor_targets_bbox_device(anchors, images, anns, C)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
print("whole call (again)    %.2f ms" % timed(lambda: UA.anchor_targets_bbox_device(anchors, images, anns, C)))
t0 = time.perf_counter()
for _ in range(10): UA.anchor_targets_bbox_device(anchors, images, anns, C)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("10 calls: host %.2f ms, + final sync %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
