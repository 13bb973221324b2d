"""debug: gradient of one step with split-K on / off (and planes on / off): where do two runs differ?"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.dp_worker import global_batch  # noqa: E402
from pyrapose_amd.engine import Engine  # noqa: E402
from pyrapose_amd.runtime import default_context  # noqa: E402

B, H, W, C, Wt, x, tg = global_batch()
ctx = default_context()
runs = {}
for name, env in (("base", {}), ("nosplit", {"PP_SPLITK_MB": "0"}), ("f32store", {"PP_PLANES": "0"}), ("f32store_nosplit", {"PP_PLANES": "0", "PP_SPLITK_MB": "0"})):
    for k, v in env.items():
        os.environ[k] = v
    eng = Engine(ctx, C, B, H, W, weights=Wt, train=True)
    for k in env:
        os.environ.pop(k)
    eng.set_targets(*[torch.from_numpy(a).cuda() for a in tg])
    eng.forward(torch.from_numpy(x).cuda())
    eng.loss_and_backward()
    torch.cuda.synchronize()
    acts = {n: a.f32(eng.ctx)[:, :a.C].clone() for n, a in eng.acts.items() if n in ("res3a", "res4a", "res5c", "pyramid", "reg_conv3", "cls_conv3", "fpn_mid3", "fpn_sum3")}
    runs[name] = (eng.params.grad.clone(), eng.params.export(eng.params.grad), acts, eng.reg_out.t.clone())
    eng.close()

def cmp(a, b):
    ga, gb = runs[a][1], runs[b][1]
    worst = []
    for k in ga:
        d = np.abs(ga[k] - gb[k]).max()
        s = np.abs(gb[k]).max()
        worst.append((d / max(s, 1e-30), d, s, k))
    worst.sort(reverse=True)
    print("==", a, "vs", b, "| total rel max:", float((runs[a][0] - runs[b][0]).abs().max() / runs[b][0].abs().max()))
    for w in worst[:6]:
        print("   %-28s rel %.2e  (diff %.2e of %.2e)" % (w[3], w[0], w[1], w[2]))
    for n in runs[a][2]:
        ta, tb = runs[a][2][n], runs[b][2][n]
        print("   act %-12s rel max %.2e" % (n, float((ta - tb).abs().max() / tb.abs().max())))
    print("   reg_out rel max %.2e" % float((runs[a][3] - runs[b][3]).abs().max() / runs[b][3].abs().max()))

cmp("base", "nosplit")
cmp("f32store", "f32store_nosplit")
cmp("base", "f32store")
