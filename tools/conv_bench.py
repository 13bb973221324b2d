#!/usr/bin/env python3
"""Micro-benchmark of one convolution shape through the C ABI (fwd / bwd-data / bwd-weight).
Usage: python3 tools/conv_bench.py [--shape reg|cls|res5|res4|res2c] [--iters 20] [--mode fwd,dgrad,wgrad]
Prints one line per mode: avg us, TFLOP/s, fraction of the 157.3 TFLOP/s f32-MFMA peak."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyrapose_amd import ops  # noqa: E402

SHAPES = {
    # name: (B, shapes, cin, cout, k, stride, pad)
    "reg": (8, [(60, 80), (30, 40), (15, 20)], 512, 512, 3, 1, 1),
    "reg0": (8, [(60, 80), (30, 40), (15, 20)], 256, 512, 3, 1, 1),
    "regout": (8, [(60, 80), (30, 40), (15, 20)], 512, 144, 3, 1, 1),
    "cls": (8, [(60, 80), (30, 40), (15, 20)], 256, 256, 3, 1, 1),
    "mask": (8, [(60, 80)], 256, 256, 3, 1, 1),
    "cls14": (14, [(60, 80), (30, 40), (15, 20)], 256, 256, 3, 1, 1),  # rows of the class head + the mask head in ONE launch (88 200): 2.7 rounds of 256 x 128 tiles
    "res5": (8, [(15, 20)], 512, 512, 3, 1, 1),
    "res5c": (8, [(15, 20)], 512, 2048, 1, 1, 0),
    "res5a": (8, [(15, 20)], 2048, 512, 1, 1, 0),
    "res4": (8, [(30, 40)], 256, 256, 3, 1, 1),
    "res4c": (8, [(30, 40)], 256, 1024, 1, 1, 0),
    "res4a": (8, [(30, 40)], 1024, 256, 1, 1, 0),
    "res3a": (8, [(60, 80)], 512, 128, 1, 1, 0),
    "fpnmid": (8, [(60, 80)], 256, 256, 3, 1, 1),
    "lat3": (8, [(60, 80)], 512, 256, 1, 1, 0),
    "res3": (8, [(60, 80)], 128, 128, 3, 1, 1),
    "res3c": (8, [(60, 80)], 128, 512, 1, 1, 0),
    "res2c": (8, [(120, 160)], 64, 256, 1, 1, 0),
    "res2b": (8, [(120, 160)], 64, 64, 3, 1, 1),
    # occupancy probes: exactly 1024 / 512 / 256 / 2048 workgroups of 128x128 at cout 512
    "r3": (18, [(64, 64)], 512, 512, 3, 1, 1),     # 128x128 tiles: 576*4 = 2304 workgroups = exactly 3 rounds of 3 per CU
    "r2": (12, [(64, 64)], 512, 512, 3, 1, 1),     # exactly 2 rounds
    "r2t": (13, [(64, 64)], 512, 512, 3, 1, 1),    # 2 rounds + 8 %
    "occ4": (8, [(64, 64)], 512, 512, 3, 1, 1),
    "occ2": (4, [(64, 64)], 512, 512, 3, 1, 1),
    "t32": (8, [(15, 20)], 32, 256, 1, 1, 0),       # fixed-cost probes: 2400 rows, 1 / 8 / 32 k-steps
    "t256": (8, [(15, 20)], 256, 256, 1, 1, 0),
    "t1024": (8, [(15, 20)], 1024, 256, 1, 1, 0),
    "u32": (8, [(60, 80)], 32, 256, 1, 1, 0),       # 38400 rows, 1 / 8 k-steps
    "u256": (8, [(60, 80)], 256, 256, 1, 1, 0),
    "down3": (8, [(60, 80)], 256, 256, 3, 2, 0),    # fpn_down3: 3x3 stride 2 'same' (pad_t = pad_l = 0)
    "down4": (8, [(30, 40)], 256, 256, 3, 2, 0),
    "r5a1": (8, [(30, 40)], 1024, 2048, 1, 2, 0),   # res5a_branch1: 1x1 stride 2
    "r4a1": (8, [(60, 80)], 512, 1024, 1, 2, 0),
    "r4a2a": (8, [(60, 80)], 512, 256, 1, 2, 0),
    # tail-quantisation probes: 48384 rows = 384 row tiles of 126 -> exactly 2 (cout 512) / 1 (cout 256) rounds of 768 workgroups
    "reg2r": (8, [(72, 84)], 512, 512, 3, 1, 1),
    "cls1r": (8, [(72, 84)], 256, 256, 3, 1, 1),
    "occ1": (2, [(64, 64)], 512, 512, 3, 1, 1),
    "occ8": (16, [(64, 64)], 512, 512, 3, 1, 1),
    "big2": (16, [(64, 64)], 512, 512, 3, 1, 1),   # 256x128 tiles: 256*4 = 1024 workgroups = 2 per CU x 2 rounds
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="reg")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--mode", default="fwd,dgrad,wgrad")
    ap.add_argument("--check", action="store_true", help="print the error of the forward modes against a float64 convolution")
    ap.add_argument("--fmt", type=int, default=0, choices=[0, 1],
                    help="plane format / arithmetic of the *3* modes: 0 = bf16 pairs (bf16x3), 1 = P16 (f16c8)")
    ap.add_argument("--ab", default="", help="ENV=a,b[,c]: time every mode once per value of an environment switch that the library reads "
                    "at every launch (PP_CONV3_DMA, PP_CONV3_DMA2, PP_CONV4P, PP_CONV4P_NST, PP_WGRAD4): interleaved rounds in ONE process")
    ap.add_argument("--no-blocker", action="store_true", help="do not queue the timed launches behind a spinning kernel (round 1-3 behaviour)")
    ap.add_argument("--rounds", type=int, default=3, help="with --ab: rounds over the values (the fastest round of each is printed too)")
    args = ap.parse_args()
    ctx = ops.Context(0)
    if args.fmt:
        ops.set_planes_format(ctx, args.fmt)
    # (the engine gives every launch lane 256 MB of scratch: split-K of small layers, the split-K remainder of an igemm4x launch)
    ws_mb = int(os.environ.get("PP_SPLITK_MB", "256"))
    if ws_mb > 0:
        ctx.set_workspace(ws_mb << 20)
    for name in args.shape.split(","):
        if name.startswith("c:"):  # c:B:H:W:cin:cout:k -- one level, stride 1, "same" padding (tile-count experiments)
            B_, H_, W_, ci_, co_, k_ = [int(v) for v in name[2:].split(":")]
            SHAPES[name] = (B_, [(H_, W_)], ci_, co_, k_, 1, k_ // 2)
        B, shapes, cin, cout, k, stride, pad = SHAPES[name]
        rows_in = sum(B * h * w for h, w in shapes)
        out_shapes = [(-(-h // stride), -(-w // stride)) for h, w in shapes]
        rows = sum(B * h * w for h, w in out_shapes)
        ld_w = (cout + 15) // 16 * 16
        d = ops.make_conv_desc(B, shapes, out_shapes, cin, cout, k, stride, pad, pad, cin, ld_w, ld_w)
        g = torch.Generator(device="cuda").manual_seed(0)
        x = torch.randn((rows_in, cin), device="cuda", generator=g)
        w = torch.randn((k * k * cin, ld_w), device="cuda", generator=g) * 0.02
        y = torch.empty((rows, ld_w), device="cuda")
        dy = torch.randn((rows, ld_w), device="cuda", generator=g)
        if ld_w != cout:
            dy[:, cout:] = 0
            w[:, cout:] = 0
        dx = torch.empty((rows_in, cin), device="cuda")
        dw = torch.zeros((k * k * cin, ld_w), device="cuda")
        db = torch.zeros((ld_w,), device="cuda")
        bias = torch.zeros((ld_w,), device="cuda")
        flops = 2.0 * rows * k * k * cin * cout
        taps = k * k
        i16 = dict(dtype=torch.int16, device="cuda")
        if cin % 32 == 0 and cout % 32 == 0:
            fh, fl = torch.zeros((taps, cout, cin), **i16), torch.zeros((taps, cout, cin), **i16)
            dh, dl = torch.zeros((taps, cin, cout), **i16), torch.zeros((taps, cin, cout), **i16)
            ops.conv_split_weights3(ctx, d, w, fh, fl, dh, dl)
        # packed (hi, lo) planes: the storage format of the engine's default mode (ops.new_planes)
        xh, xl = ops.new_planes(rows_in, cin)
        gh, gl = ops.new_planes(rows, ld_w)
        ops.split_planes3(ctx, x, xh, xl)
        ops.split_planes3(ctx, dy, gh, gl)
        yh, yl = ops.new_planes(rows, ld_w)
        dxh, dxl = ops.new_planes(rows_in, cin)
        # a sparse gradient like the 3D-box head's: non-zero in a few square patches per image of the first level
        dys = torch.zeros_like(dy)
        h0, w0 = out_shapes[0]
        for n in range(B):
            for _ in range(2):
                cy, cx, r = int(torch.randint(0, h0, (1,))), int(torch.randint(0, w0, (1,))), int(os.environ.get("PP_PATCH", "8"))
                for yy in range(max(0, cy - r), min(h0, cy + r)):
                    a = (n * h0 + yy) * w0
                    dys[a + max(0, cx - r): a + min(w0, cx + r)] = dy[a + max(0, cx - r): a + min(w0, cx + r)]
        skip = ops.row_block_list(ctx, dys, cout)
        skip_p = ops.row_block_list(ctx, dys, cout)
        sh, sl = ops.new_planes(rows, ld_w)
        ops.split_planes3(ctx, dys, sh, sl)
        fns = {"fwd3pp": lambda: ops.conv_fwd3(ctx, d, None, fh, fl, bias, None, True, None, x_planes=(xh, xl), y_planes=(yh, yl)),
               "dgrad3pp": lambda: ops.conv_bwd_data3(ctx, d, None, dh, dl, None, None, None, dy_planes=(gh, gl), dx_planes=(dxh, dxl),
                                                      relu_src_hi=xh),
               "dgrad3sp": lambda: ops.conv_bwd_data3(ctx, d, None, dh, dl, None, None, None, dy_planes=(sh, sl), dx_planes=(dxh, dxl),
                                                      relu_src_hi=xh, dy_skip=skip_p),
               "wgrad3sp": lambda: ops.conv_bwd_weight3(ctx, d, None, None, dw, db, x_planes=(xh, xl), dy_planes=(sh, sl), dy_skip=skip_p),
               "dgrad3s": lambda: ops.conv_bwd_data3(ctx, d, dys, dh, dl, None, x, dx, dy_skip=skip),
               "wgrad3s": lambda: ops.conv_bwd_weight3(ctx, d, x, dys, dw, db, dy_skip=skip),
               "dgrad3d": lambda: ops.conv_bwd_data3(ctx, d, dys, dh, dl, None, x, dx),
               "wgrad3d": lambda: ops.conv_bwd_weight3(ctx, d, x, dys, dw, db),
               "rowlist": lambda: ops.row_block_list(ctx, dys, cout, skip[0], skip[1]),
               "fwd3c": lambda: ops.conv_fwd3(ctx, d, x, fh, fl, bias, None, True, y, x_capture=(xh, xl)),
               "dgrad3c": lambda: ops.conv_bwd_data3(ctx, d, dy, dh, dl, None, x, dx, dy_capture=(gh, gl)),
               "fwd3p": lambda: ops.conv_fwd3(ctx, d, None, fh, fl, bias, None, True, y, x_planes=(xh, xl)),
               "dgrad3p": lambda: ops.conv_bwd_data3(ctx, d, None, dh, dl, None, x, dx, dy_planes=(gh, gl)),
               "wgrad3p": lambda: ops.conv_bwd_weight3(ctx, d, None, None, dw, db, x_planes=(xh, xl), dy_planes=(gh, gl)),
               "splitx": lambda: ops.split_planes3(ctx, x, xh, xl),
               "fwd3": lambda: ops.conv_fwd3(ctx, d, x, fh, fl, bias, None, True, y),
               "dgrad3": lambda: ops.conv_bwd_data3(ctx, d, dy, dh, dl, None, x, dx),
               "wgrad3": lambda: ops.conv_bwd_weight3(ctx, d, x, dy, dw, db),
               "split3": lambda: ops.conv_split_weights3(ctx, d, w, fh, fl, dh, dl),
               "fwd": lambda: ops.conv_fwd(ctx, d, x, w, bias, None, True, y),
               "dgrad": lambda: ops.conv_bwd_data(ctx, d, dy, w, None, x, dx),
               "wgrad": lambda: ops.conv_bwd_weight(ctx, d, x, dy, dw, db)}
        for mode in args.mode.split(","):
            fn = fns[mode]

            def timed():
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                # a launch through ctypes costs the host 20-30 us: behind a spinning kernel the host runs ahead, and the events
                # bracket back-to-back GPU work (without it every launch under ~30 us measures the host, not the kernel)
                if not args.no_blocker:
                    torch.cuda._sleep(int(2.0e6 * (1 + args.iters * 0.04)))
                s.record()
                for _ in range(args.iters):
                    fn()
                e.record()
                torch.cuda.synchronize()
                return s.elapsed_time(e) * 1e3 / args.iters
            if args.ab:
                key, vals = args.ab.split("=")
                vals = vals.split(",")
                res = {v: [] for v in vals}
                for _ in range(args.rounds):
                    for v in vals:
                        os.environ[key] = v
                        res[v].append(timed())
                os.environ.pop(key, None)
                for v in vals:
                    med, best = sorted(res[v])[len(res[v]) // 2], min(res[v])
                    print("%-7s %-8s %s=%-3s rows=%d cin=%d cout=%d k=%d  median %.1f us (%.1f TFLOP/s)  best %.1f us" %
                          (name, mode, key, v, rows, cin, cout, k, med, flops / med / 1e6, best), flush=True)
                continue
            us = timed()
            tf = flops / us / 1e6
            print("%-7s %-6s rows=%d cin=%d cout=%d k=%d  %.1f us  %.1f TFLOP/s  %.1f%% of f32-MFMA peak" %
                  (name, mode, rows, cin, cout, k, us, tf, 100 * tf / 157.3), flush=True)
            if args.check and mode in ("wgrad3p", "wgrad3", "wgrad") and k == 3 and stride == 1:
                # the centre tap of the weight gradient against float64: dW[1][1] = x^T dy over all rows (no shift, no padding)
                dw.zero_()
                fn()
                torch.cuda.synchronize()
                c = (k * k) // 2
                ref = x.double().t() @ dy[:, :cout].double()
                got = dw[c * cin:(c + 1) * cin, :cout].double()
                err = got - ref
                print("        centre tap vs float64: rel-L2 %.3e  max|err|/max|ref| %.3e" %
                      (float(err.norm() / ref.norm()), float(err.abs().max() / ref.abs().max())), flush=True)
            if args.check and mode == "fwd3pp":
                # planes out: decode and compare every row of level 0 (all images) against float64; also a checksum of the raw planes
                import torch.nn.functional as F
                h0, w0 = shapes[0]
                got = ops.planes_to_f32((yh, yl), args.fmt)[: B * h0 * w0, :cout].double()
                xi = ops.planes_to_f32((xh, xl), args.fmt)[: B * h0 * w0].double().reshape(B, h0, w0, cin).permute(0, 3, 1, 2)
                wt = w[:, :cout].double().reshape(k, k, cin, cout).permute(3, 2, 0, 1)
                ref = F.relu(F.conv2d(xi, wt, None, stride=stride, padding=pad)).permute(0, 2, 3, 1).reshape(-1, cout)
                err = got - ref
                print("        planes out vs float64 (level 0, all images): rel-L2 %.3e  max|err|/max|ref| %.3e  checksum %d" %
                      (float(err.norm() / ref.norm()), float(err.abs().max() / ref.abs().max()),
                       int(yh.to(torch.int64).sum() * 3 + yl.to(torch.int64).sum())), flush=True)
            if args.check and mode in ("fwd", "fwd3", "fwd3p"):
                # error of the launch against float64 (torch on the device, a sample of the output rows of the first level)
                import torch.nn.functional as F
                h0, w0 = shapes[0]
                xi = x[: h0 * w0].double().reshape(1, h0, w0, cin).permute(0, 3, 1, 2)
                wt = w[:, :cout].double().reshape(k, k, cin, cout).permute(3, 2, 0, 1)
                ref = F.relu(F.conv2d(xi, wt, None, stride=stride, padding=pad)).permute(0, 2, 3, 1).reshape(-1, cout)
                got = y[: ref.shape[0], :cout].double()
                err = (got - ref)
                print("        vs float64 (image 0, level 0): rel-L2 %.3e  max|err|/max|ref| %.3e" %
                      (float(err.norm() / ref.norm()), float(err.abs().max() / ref.abs().max())), flush=True)


if __name__ == "__main__":
    main()
