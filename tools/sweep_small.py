#!/usr/bin/env python3
"""Tile x split-K sweep of the small backbone launches on packed planes: us per launch for every (shape, direction, tile,
splits) next to the cost model's own pick.  One process: PP_CONV3_TILE / PP_CONV3_SPLITS are read at every dispatch."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyrapose_amd import ops  # noqa: E402
from tools.conv_bench import SHAPES  # noqa: E402


def main():
    shapes = (sys.argv[1] if len(sys.argv) > 1 else "res3,res4,res5,res3c,res4c,res5c,res5a,res4a").split(",")
    SHAPES.setdefault("res4a", (8, [(30, 40)], 1024, 256, 1, 1, 0))
    ctx = ops.Context(0)
    ctx.set_workspace(64 << 20)
    for name in shapes:
        B, shp, cin, cout, k, stride, pad = SHAPES[name]
        rows = sum(B * h * w for h, w in shp)
        d = ops.make_conv_desc(B, shp, shp, cin, cout, k, stride, pad, pad, cin, cout, cout)
        g = torch.Generator(device="cuda").manual_seed(0)
        x = torch.randn((rows, cin), device="cuda", generator=g)
        dy = torch.randn((rows, cout), device="cuda", generator=g)
        w = torch.randn((k * k * cin, cout), device="cuda", generator=g) * 0.02
        i16 = dict(dtype=torch.int16, device="cuda")
        fh, fl = torch.zeros((k * k, cout, cin), **i16), torch.zeros((k * k, cout, cin), **i16)
        dh, dl = torch.zeros((k * k, cin, cout), **i16), torch.zeros((k * k, cin, cout), **i16)
        ops.conv_split_weights3(ctx, d, w, fh, fl, dh, dl)
        xp, gp = ops.new_planes(rows, cin), ops.new_planes(rows, cout)
        ops.split_planes3(ctx, x, *xp)
        ops.split_planes3(ctx, dy, *gp)
        yp, dxp = ops.new_planes(rows, cout), ops.new_planes(rows, cin)
        rp = ops.new_planes(rows, cout)
        bias = torch.zeros((cout,), device="cuda")
        fns = {"fwd": lambda: ops.conv_fwd3(ctx, d, None, fh, fl, bias, None, True, None, x_planes=xp, y_planes=yp, res_planes=rp),
               "dgrad": lambda: ops.conv_bwd_data3(ctx, d, None, dh, dl, None, None, None, dy_planes=gp, dx_planes=dxp, addend_planes=xp,
                                                   relu_src_hi=xp[0])}

        def t(fn, iters=20):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(iters):
                fn()
            e.record()
            torch.cuda.synchronize()
            return s.elapsed_time(e) * 1e3 / iters
        for mode, fn in fns.items():
            os.environ.pop("PP_CONV3_TILE", None)
            os.environ.pop("PP_CONV3_SPLITS", None)
            base = t(fn)
            best = (base, "model")
            out = []
            for tile in ("2,2", "1,2", "2,1", "1,1"):
                row = []
                for sp in (1, 2, 3, 4, 6, 8, 12, 16):
                    os.environ["PP_CONV3_TILE"], os.environ["PP_CONV3_SPLITS"] = tile, str(sp)
                    us = t(fn)
                    row.append("s%d=%.1f" % (sp, us))
                    if us < best[0]:
                        best = (us, "%s s%d" % (tile, sp))
                out.append("   %s: %s" % (tile, "  ".join(row)))
            print("== %-6s %-5s rows=%d %dx%d k%d: model pick %.1f us; best %.1f us (%s)" % (name, mode, rows, cin, cout, k, base, best[0], best[1]))
            print("\n".join(out), flush=True)


if __name__ == "__main__":
    main()
