import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from pyrapose_amd import arch
from pyrapose_amd.engine import Engine
from pyrapose_amd.runtime import default_context
from test_gpu_model import synth_input, random_targets
ctx = default_context()
B, H, W, C = 2, 64, 96, 13
rng = np.random.default_rng(4)
Wt = arch.init_weights(C, seed=5)
x = synth_input(rng, B, H, W)
engs = {}
for mode in ("f32", "bf16x3"):
    e = Engine(ctx, C, B, H, W, weights=Wt, train=True, conv_mode=mode)
    if mode == "f32":
        tg = random_targets(np.random.default_rng(99), B, e.N, e.M3, C)
    e.set_targets(*[torch.from_numpy(a).cuda() for a in tg])
    e.forward(torch.from_numpy(x).cuda())
    e.loss_and_backward()
    torch.cuda.synchronize()
    engs[mode] = e
a, b = engs["f32"], engs["bf16x3"]
def rl2(u, v):
    u = u.double().cpu().numpy(); v = v.double().cpu().numpy()
    return float(np.sqrt(((u - v) ** 2).sum() / max((v ** 2).sum(), 1e-300)))
print("activations (rel L2, bf16x3 vs f32):")
for name in ("res3d", "res4f", "res5c", "fpn_lat5", "P3", "P4", "P5", "reg_conv3", "cls_conv3", "mask_conv3", "reg_out", "cls_out", "mask_out"):
    ta, tb = a.acts[name].t, b.acts[name].t
    c = a.acts[name].C
    print("  %-12s %.3e" % (name, rl2(tb[:, :c], ta[:, :c])))
print("loss grads:", rl2(b.g_reg[:, :144], a.g_reg[:, :144]), rl2(b.g_cls[:, :117], a.g_cls[:, :117]), rl2(b.g_mask[:, :13], a.g_mask[:, :13]))
ga, gb = a.params.export(a.params.grad), b.params.export(b.params.grad)
rows = []
for k in ga:
    if np.any(ga[k]):
        rows.append((rl2(torch.from_numpy(gb[k]), torch.from_numpy(ga[k])), k))
rows.sort(reverse=True)
print("head/fpn gradient tensors in order:")
d = dict((k, v) for v, k in rows)
for k in ga:
    if k.split("_")[0] in ("mask", "cls", "reg") or k.startswith("fpn") or k[:2] in ("P3", "P4", "P5"):
        if k in d: print("  %-22s %.3e" % (k, d[k]))
print("activations of heads:")
for name in a.acts:
    if name.split("_")[0] in ("mask", "cls", "reg"):
        c = a.acts[name].C
        print("  %-12s %.3e" % (name, rl2(b.acts[name].t[:, :c], a.acts[name].t[:, :c])))
print("gradient tensors, worst first:")
for r in rows[:25]:
    print("  %-26s %.3e" % (r[1], r[0]))
print("median", rows[len(rows)//2])
