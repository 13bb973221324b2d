"""debug: per-tensor comparison of the engine's arithmetics on the same weights / inputs (forward + gradients)"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pyrapose_amd import ops
from pyrapose_amd.engine import Engine
B, H, W, C = 2, 96, 128, 5
from pyrapose_amd.runtime import default_context
ctx = default_context()
torch.manual_seed(0)
ref = Engine(ctx, C, B, H, W, train=True, conv_mode="bf16x3", seed=3)
Wt = ref.get_weights() if hasattr(ref, "get_weights") else None
x = torch.randn(B, H, W, 3, device="cuda")
outs = {}
for mode in ("bf16x3", "f16c8", "mixed"):
    e = ref if mode == "bf16x3" else Engine(ctx, C, B, H, W, train=True, conv_mode=mode, seed=3)
    e.forward(x)
    torch.cuda.synchronize()
    outs[mode] = {n: a.f32(e.ctx).clone() for n, a in e.acts.items() if ":fmt" not in n}
for mode in ("f16c8", "mixed"):
    print("==", mode)
    for n, r in outs["bf16x3"].items():
        if n in outs[mode]:
            t = outs[mode][n]
            d = (t - r).abs().max().item() / max(r.abs().max().item(), 1e-30)
            if d > 1e-3 or n in ("reg_out", "cls_out", "mask_out", "C3", "C5"):
                print("  %-28s %.3e  (max %.3e)" % (n, d, r.abs().max().item()))
