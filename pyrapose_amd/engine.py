"""Execution engine of the PyraPose hot path on one MI355X: builds the launch plan once for a fixed
(batch, height, width, classes) and replays it.  Every FLOP runs in the HIP library; this file only
owns buffers (torch tensors) and the order of C-ABI calls.

Forward graph = reference models/resnet.py:79-110 + models/retinanet.py:180-214,224-229,296-299.
Backward = reverse-mode over the same op list with the sum-of-consumers and the ReLU mask fused
into the last data-gradient launch of each tensor (see DESIGN.md §Backward dataflow).
Loss / optimizer = bin/train.py:95-102.
"""
import os as _os
from collections import OrderedDict

import numpy as np
import torch

from . import arch, ops
from ._lib import ParamDesc, RowSpace


def _ru(v, m):
    return (v + m - 1) // m * m


def tf_same_pad(n_in, k, s):
    out = -(-n_in // s)
    total = max((out - 1) * s + k - n_in, 0)
    return total // 2


class Op(object):
    """One C-ABI launch of the plan: `fn()` enqueues it on the ctx stream."""
    __slots__ = ("fn", "kind", "name", "flops", "wrange", "lane", "reads", "waits", "done_ev")

    def __init__(self, fn, kind, name="", flops=0.0, wrange=None, lane=0):
        self.fn, self.kind, self.name, self.flops, self.wrange, self.lane = fn, kind, name, flops, wrange, lane
        self.reads = ()      # forward plan: the activations this launch reads ...
        self.waits = ()      # ... -> events of their producers on OTHER lanes, awaited before the launch
        self.done_ev = None  # recorded after the launch when a consumer on another lane needs it

    def __call__(self):
        self.fn()


class Act(object):
    """A level-major activation matrix [rows, ld] (see pp_rowspace), stored as float32 (`t`) or -- what the bf16x3 convs
    produce and consume in the default mode -- as a pair of bf16 planes (`pl` = (hi, lo) int16 [rows, ld], value = hi + lo)."""

    def __init__(self, name, n_img, shapes, C, ld=None, t=None, needs_grad=False, relu=False, pl=None, planes=False, fmt=0):
        self.name, self.n_img, self.shapes, self.C = name, n_img, list(shapes), C
        self.fmt = int(fmt)  # plane format of `pl`: 0 = bf16 pairs, 1 = P16 (csrc/planes_fmt.h)
        self.ld = ld if ld is not None else C
        self.rows = sum(n_img * h * w for h, w in self.shapes)
        if planes:
            self.t = None
            self.pl = pl if pl is not None else _new_planes(self.rows, self.ld)
        else:
            self.t = t if t is not None else torch.empty((self.rows, self.ld), dtype=torch.float32, device="cuda")
            self.pl = None
        self.needs_grad, self.relu = needs_grad, relu
        self.contribs = []
        self.prod_ops = []  # forward launches that write this tensor (an alias lists those of its parts)

    def rowspace(self):
        return RowSpace.make(self.n_img, self.shapes)

    def view(self):
        return _view(self.t, self.pl)

    def f32(self, ctx):
        """the matrix as float32 (a copy merged from the planes when that is how it is stored): tests / inspection"""
        if self.t is not None:
            return self.t
        out = torch.empty((self.rows, self.ld), dtype=torch.float32, device="cuda")
        ops.merge_planes3(ctx.twin(self.fmt), self.pl, out)
        return out


def _new_planes(rows, ld):
    return ops.new_planes(rows, ld)


def _view(t, pl):
    """input view of a tensor that exists as float32 and / or planes (the float32 copy wins: it is the exact one)"""
    return ops.tview(t, None) if t is not None else ops.tview(None, pl)


class Grad(object):
    """A gradient matrix: float32 (`t`) and / or bf16 planes (`pl`); rows(a, b) = the same for a row range."""
    __slots__ = ("t", "pl", "within", "fmt", "lazy")

    def __init__(self, t, pl=None, fmt=0):
        self.t, self.pl, self.fmt = t, pl, int(fmt)  # fmt: plane format of `pl` (a P16 gradient also carries the factor 2^G)
        self.within = None  # uint8 flags of the only 32-row blocks that can hold a non-zero (a sparse data gradient without addend)
        self.lazy = False   # the rows outside `within` were never written (pp_ctx_set_row_block_lazy): every reader goes by flags

    def rows(self, r0, r1):
        return Grad(None if self.t is None else self.t[r0:r1], None if self.pl is None else (self.pl[0][r0:r1], self.pl[1][r0:r1]), self.fmt)

    def view(self):
        return _view(self.t, self.pl)

    def shape(self):
        if self.t is not None:
            return tuple(self.t.shape)
        return (self.pl[0].shape[0], ops.planes_ld(self.pl))  # packed planes: views [rows, ld / 8, 8] of one buffer

    def contiguous(self):
        """whole rows, one after the other (a row range of a matrix qualifies; packed planes always do)"""
        return self.t.is_contiguous() if self.t is not None else True


class ParamStore(object):
    """Flat float32 parameter / gradient / Adam-state buffers + the table that cuts them into tensors."""

    ALIGN = 64

    def __init__(self, specs, train):
        self.specs = OrderedDict((s.name, s) for s in specs)
        self.entries = OrderedDict()  # key -> dict(offset, rows, ld, count, ...)
        off = 0
        soff = 0
        for s in specs:
            cin_eff = 4 if s.cin == 3 else s.cin
            rows = _ru(s.k * s.k * cin_eff, 16)
            ld = _ru(s.cout, 16)
            self.entries[s.name + "/kernel"] = dict(offset=off, rows=rows, ld=ld, count=rows * ld, trainable=s.trainable,
                                                    scale_off=(soff if s.bn else -1), l2=s.l2)
            off = _ru(off + rows * ld, self.ALIGN)
            # bias slot: the conv bias (trainable) or the frozen-BN shift
            self.entries[s.name + "/bias"] = dict(offset=off, rows=1, ld=ld, count=ld, trainable=(s.bias and s.trainable),
                                                  scale_off=-1, l2=0.0)
            off = _ru(off + ld, self.ALIGN)
            if s.bn:
                soff += ld
        self.total = off
        self.n_scales = max(soff, 16)
        z = lambda n: torch.zeros((n,), dtype=torch.float32, device="cuda")
        self.w_master, self.w_eff = z(self.total), z(self.total)
        self.scales = torch.ones((self.n_scales,), dtype=torch.float32, device="cuda")
        self.train = train
        if train:
            self.grad, self.m, self.v = z(self.total), z(self.total), z(self.total)

    def view(self, buf, key):
        e = self.entries[key]
        return buf[e["offset"]: e["offset"] + e["count"]].view(e["rows"], e["ld"])

    def descs(self):
        out = []
        for key, e in self.entries.items():
            d = ParamDesc()
            d.offset, d.count, d.ld, d.trainable, d.scale_off, d.l2 = e["offset"], e["count"], e["ld"], int(e["trainable"]), e["scale_off"], e["l2"]
            out.append(d)
        return out

    def load(self, W):
        """W: dict in Keras layout (arch.init_weights).  Folds frozen BN into (scale, shift)."""
        wm = np.zeros((self.total,), np.float32)
        sc = np.ones((self.n_scales,), np.float32)
        for name, s in self.specs.items():
            ek, eb = self.entries[name + "/kernel"], self.entries[name + "/bias"]
            k = np.asarray(W[name + "/kernel"], np.float32)
            assert k.shape == (s.k, s.k, s.cin, s.cout), (name, k.shape)
            if s.cin == 3:
                k = np.concatenate([k, np.zeros((s.k, s.k, 1, s.cout), np.float32)], axis=2)
            k2 = k.reshape(-1, s.cout)
            buf = np.zeros((ek["rows"], ek["ld"]), np.float32)
            buf[: k2.shape[0], : s.cout] = k2
            wm[ek["offset"]: ek["offset"] + ek["count"]] = buf.reshape(-1)
            b = np.zeros((eb["ld"],), np.float32)
            if s.bn:
                g, be = np.asarray(W[s.bn + "/gamma"], np.float64), np.asarray(W[s.bn + "/beta"], np.float64)
                mu, var = np.asarray(W[s.bn + "/mean"], np.float64), np.asarray(W[s.bn + "/var"], np.float64)
                scale = g / np.sqrt(var + arch.BN_EPS)
                sc[ek["scale_off"]: ek["scale_off"] + s.cout] = scale.astype(np.float32)
                b[: s.cout] = (be - mu * scale).astype(np.float32)
            elif s.bias:
                b[: s.cout] = np.asarray(W[name + "/bias"], np.float32)
            wm[eb["offset"]: eb["offset"] + eb["count"]] = b
        self._bn_raw = {k: np.asarray(v) for k, v in W.items() if k.split("/")[1] in ("gamma", "beta", "mean", "var")}
        self.w_master.copy_(torch.from_numpy(wm))
        self.scales.copy_(torch.from_numpy(sc))
        self.refresh_eff()

    def refresh_eff(self):
        """w_eff = w_master * scale[co] for frozen-BN convs (host-driven; the optimizer does this on device)."""
        self.w_eff.copy_(self.w_master)
        for name, s in self.specs.items():
            if s.bn:
                ek = self.entries[name + "/kernel"]
                sv = self.scales[ek["scale_off"]: ek["scale_off"] + ek["ld"]]
                self.view(self.w_eff, name + "/kernel").mul_(sv.view(1, -1))

    def export(self, buf=None):
        """Back to the Keras-layout dict (kernels HWIO, biases; frozen BN statistics passed through)."""
        buf = self.w_master if buf is None else buf
        host = buf.detach().cpu().numpy()
        W = OrderedDict()
        for name, s in self.specs.items():
            ek, eb = self.entries[name + "/kernel"], self.entries[name + "/bias"]
            cin_eff = 4 if s.cin == 3 else s.cin
            k = host[ek["offset"]: ek["offset"] + ek["count"]].reshape(ek["rows"], ek["ld"])[: s.k * s.k * cin_eff, : s.cout]
            k = k.reshape(s.k, s.k, cin_eff, s.cout)[:, :, : s.cin, :]
            W[name + "/kernel"] = np.ascontiguousarray(k)
            if s.bias:
                W[name + "/bias"] = host[eb["offset"]: eb["offset"] + s.cout].copy()
        W.update(getattr(self, "_bn_raw", {}))
        return W


    def trainable_keys(self):
        """'<layer>/kernel' and '<layer>/bias' of every tensor the optimizer updates, in layout (= graph) order"""
        return [k for k, e in self.entries.items() if e["trainable"]]

    def export_trainable(self, buf):
        """the slices of a flat per-parameter buffer (Adam m / v) that belong to trainable tensors, in the Keras layout of export()"""
        host = buf.detach().cpu().numpy()
        out = OrderedDict()
        for key in self.trainable_keys():
            name, var = key.rsplit("/", 1)
            s, e = self.specs[name], self.entries[key]
            if var == "kernel":
                cin_eff = 4 if s.cin == 3 else s.cin
                k = host[e["offset"]: e["offset"] + e["count"]].reshape(e["rows"], e["ld"])[: s.k * s.k * cin_eff, : s.cout]
                out[key] = np.ascontiguousarray(k.reshape(s.k, s.k, cin_eff, s.cout)[:, :, : s.cin, :])
            else:
                out[key] = host[e["offset"]: e["offset"] + s.cout].copy()
        return out

    def import_trainable(self, buf, tensors):
        """the inverse: {key: array in Keras layout} -> the flat buffer (padding stays zero); every trainable key must be there"""
        host = np.zeros((self.total,), np.float32)
        for key in self.trainable_keys():
            name, var = key.rsplit("/", 1)
            s, e = self.specs[name], self.entries[key]
            a = np.asarray(tensors[key], np.float32)
            if var == "kernel":
                if a.shape != (s.k, s.k, s.cin, s.cout):
                    raise ValueError("optimizer state of %s: shape %s, expected %s" % (key, a.shape, (s.k, s.k, s.cin, s.cout)))
                if s.cin == 3:
                    a = np.concatenate([a, np.zeros((s.k, s.k, 1, s.cout), np.float32)], axis=2)
                k2 = a.reshape(-1, s.cout)
                b = np.zeros((e["rows"], e["ld"]), np.float32)
                b[: k2.shape[0], : s.cout] = k2
                host[e["offset"]: e["offset"] + e["count"]] = b.reshape(-1)
            else:
                if a.shape != (s.cout,):
                    raise ValueError("optimizer state of %s: shape %s, expected %s" % (key, a.shape, (s.cout,)))
                host[e["offset"]: e["offset"] + s.cout] = a
        buf.copy_(torch.from_numpy(host))


class Engine(object):
    def __init__(self, ctx, num_classes, batch, height, width, backbone="resnet50", weights=None, train=True, seed=0,
                 lr=1e-5, clipnorm=0.001, freeze_backbone=False, conv_mode=None, pyramid="sparse", anchor_params=None,
                 loss_params=None):
        """pyramid: 'sparse' (__create_sparceFPN, what the reference's retinanet() builds), 'fpn' (__create_FPN) or 'p3p7'
        (__create_pyramid_features: P3..P7).  anchor_params: utils.anchors.AnchorParameters (sizes / strides per level);
        default: the reference default for 3 levels, the 5-level RetinaNet sizes 32..512 for 'p3p7'.
        loss_params: dict(box=(weight, sigma), cls=(alpha, gamma), mask=(alpha, gamma)) -- the hyper-parameters of the three
        compiled losses (bin/train.py:95-102; defaults = the reference's: orthogonal_l1(0.125, 3.0), focal(0.25, 2.0))."""
        # The engine owns a PRIVATE context bound to the stream that is current now: a later default_context() call under
        # another torch stream re-binds the shared context, never this one, so lane 0 and the fork/join events of the other
        # lanes always speak about the same stream.  Work of the caller's current stream is ordered around every public
        # entry point by _enter() / _leave().
        self._own_ctx = ops.Context(ctx.device, ctx.stream)
        ctx = self._own_ctx
        self.ctx, self.C, self.B, self.H, self.W = ctx, int(num_classes), int(batch), int(height), int(width)
        lp = dict(box=(0.125, 3.0), cls=(0.25, 2.0), mask=(0.25, 2.0))
        lp.update(loss_params or {})
        self.loss_params = {k: (float(v[0]), float(v[1])) for k, v in lp.items()}
        assert pyramid in arch.PYRAMIDS, pyramid
        self.pyramid = pyramid
        from .utils import anchors as _ua
        if anchor_params is None:
            anchor_params = _ua.AnchorParameters.default if pyramid != "p3p7" else _ua.AnchorParameters.p3p7
        self.anchor_params = anchor_params
        self.A = anchor_params.num_anchors()
        assert len(anchor_params.sizes) == len(arch.PYRAMID_LEVELS[pyramid]), "anchor_params need one size / stride per pyramid level"
        self.backbone = backbone
        self.train = train
        self.lr, self.clipnorm = lr, clipnorm
        self.beta1, self.beta2, self.eps = 0.9, 0.999, 1e-7
        self.specs = arch.all_specs(self.C, backbone, pyramid, self.A)
        if freeze_backbone:  # --freeze-backbone: utils/model.py:18-27 freeze() applied to the ResNet (bin/train.py:74)
            for s in self.specs:
                if s.bn:
                    s.trainable = False
        self.params = ParamStore(self.specs, train)
        self.params.load(weights if weights is not None else arch.init_weights(self.C, seed, backbone, pyramid, self.A))
        # conv arithmetic: "bf16x3" (default) = 3 x bf16 MFMA per product, f32 accumulation (conv3.hip; ~2^-16 per
        # product, whole-graph outputs within 2e-5 of the f32 path) for every conv except the 3-channel stem;
        # "f32" = exact f32 MFMA everywhere (conv.hip).  Both hold the 1e-3 head-output bar.
        # conv_mode / PP_CONV_MODE: "f32" (exact f32 MFMA), or the plane-stored family with one of three arithmetics:
        #   "bf16x3"  every layer on bf16 pairs, three bf16 MFMAs per product (4.5e-6 per launch against float64);
        #   "f16c8"   every layer on P16 planes, one f16 MFMA + half a block-scaled e5m2 MFMA per 16-deep step (2 units per product
        #             instead of 3, 2.1e-5 per launch, 1.15-1.19x faster on MFMA-bound launches);
        #   "mixed"   (default) the backbone on bf16x3 -- its ~50 HBM- / latency-bound layers are where rounding accumulates and
        #             where MFMA work is not what costs -- and the MFMA-bound FPN + heads on f16c8; C3 / C4 / C5 and the gradients
        #             flowing back into them are re-encoded at the boundary (pp_convert_planes).
        mode = conv_mode or _os.environ.get("PP_CONV_MODE", "mixed")
        assert mode in ("f32", "bf16x3", "f16c8", "mixed"), mode
        if mode in ("f16c8", "mixed") and _os.environ.get("PP_PLANES", "1") == "0":
            mode = "bf16x3"  # float32 storage (PP_PLANES=0) exists for the bf16x3 arithmetic only
        self.arith = mode
        self.conv_mode = "f32" if mode == "f32" else "bf16x3"  # the kernel family (csrc/conv.hip or csrc/conv3.hip)
        # Storage format of activations and gradients in bf16x3 mode (PP_PLANES=0 turns it off): every tensor that a bf16x3 conv
        # produces is written as bf16 (hi, lo) planes ONLY -- 4 bytes per element like float32 -- and stays in that format
        # through the FPN's adds / resampling; convs read their gathered operand, the weight gradients both operands, and the
        # epilogues their residual / addend / ReLU source from planes, so no kernel converts f32 -> bf16 inside its loop.
        # The products are the same as with the in-loop split (same hi, lo); a value read back from planes (residuals, adds)
        # is within 2^-17 of the float32 it was split from.  float32 stays for: the image, conv1 / pool1 (frozen prefix), the
        # three head outputs (losses, export) and the loss gradients.
        self.po = self.conv_mode == "bf16x3" and _os.environ.get("PP_PLANES", "1") != "0"
        self.use_act_planes = False  # (the round-1 "both formats" mode is gone: planes-only supersedes it)
        # split capture (PP_CAPTURE=1): the forward / bwd-data launch of a 3x3 stride-1 conv also stores the bf16 split of
        # its gathered operand (it has just computed it), and the weight-gradient launch of the layer, enqueued after the
        # layer's bwd-data launch, reads both operands pre-split.  Off by default: the weight-gradient kernel alone gains
        # 11-15 % and the capture costs the 512-channel launches nothing, but the training step does not move (one lane:
        # +0.6 %; two lanes: -1 % -- with both lanes busy the MFMA pipes, not the conversion VALU, are what is shared).
        self.capture = (self.conv_mode == "bf16x3" and not self.po and _os.environ.get("PP_CAPTURE", "0") == "1")
        # Sparse backward of the 3D-box head (PP_SPARSE_BWD=0 disables): orthogonal_l1 keeps the rows with anchor state 1 only
        # (losses.py:332-333), so the gradient entering that head -- and, dilated by one pixel per 3x3 layer, every gradient
        # inside it -- is exactly zero away from the positive anchors.  One scan per layer lists the 32-row blocks that hold a
        # non-zero (pp_row_block_list); the weight gradient reduces over those blocks only and the data gradient skips the
        # output tiles that cannot see one.  Exact: a zero row adds 0.0 to every sum.
        self.sparse_bwd = tuple(t for t in _os.environ.get("PP_SPARSE_BWD", "reg").split(",") if t and t != "0") \
            if self.conv_mode == "bf16x3" else ()
        # opt-in: the FORWARD of that head on the same blocks in training steps (_plan_sparse_forward)
        self.sparse_fwd = (self.train and self.po and "reg" in self.sparse_bwd and _os.environ.get("PP_SPARSE_FWD", "0") == "1")
        self.capture_min_cin = int(_os.environ.get("PP_CAPTURE_MIN_CIN", "64"))
        self.capture_skip = tuple(t for t in _os.environ.get("PP_CAPTURE_SKIP", "").split(",") if t)
        self.planes = OrderedDict()  # spec name -> dict(desc, fwd_hi, fwd_lo, dg_hi, dg_lo)
        self.fwd_ops, self.graph_ops, self.bwd_ops = [], [], []
        # Launch lanes: lane 0 is the ctx stream; lanes 1-2 are side streams.  Independent kernel chains (the three
        # heads in forward; weight gradients vs the data-gradient chain in backward) are enqueued on different
        # lanes so that the tail of one launch (last, partially filled round of workgroups) is covered by
        # workgroups of another -- see DESIGN.md §Concurrency.  PP_LANES=1 serialises everything on lane 0.
        # Measured (tools/phase_times.py, batch 8): forward 8.20 / 8.02 / 8.44 ms and backward 15.45 / 14.19 / 14.13 ms
        # at 1 / 2 / 3 lanes -> two lanes (a third concurrent head chain only adds cache pressure).
        os = _os
        self.n_lanes = max(1, min(3, int(os.environ.get("PP_LANES", "2"))))
        self.streams = [ctx.stream] + [torch.cuda.Stream(device=ctx.device) for _ in range(self.n_lanes - 1)]
        self.ctxs = [ctx] + [ops.Context(ctx.device, st) for st in self.streams[1:]]
        # split-K scratch (one per lane): the small-M convs of res5 / FPN level 5 cannot fill 256 CUs with output tiles
        # alone (3x3 512->512 on 2400 rows: 112 -> 174 TFLOP/s with 4 slices).  Deterministic (fixed-order slice sum).
        # (also the slices of the weight-gradient launches: splits x |dW| in f32, e.g. 16 x 9.4 MB for the 512-wide head convs)
        ws_mb = int(os.environ.get("PP_SPLITK_MB", "256"))
        if self.conv_mode == "bf16x3" and ws_mb > 0:
            for c in self.ctxs:
                c.set_workspace(ws_mb << 20)
        # The gradient chain of the plane-stored mode travels multiplied by a power of two (the hi plane holds IEEE halves, which
        # stop at 6e-8; loss gradients are ~1e-7): {2^G, 2^-G} lives on the device, is refreshed from the positive counts of every
        # step (pp_grad_scale_from_counts), multiplies the three loss gradients when they are split into planes, rides through
        # every bwd-data / pointwise launch, and is divided out by the weight-gradient launches (pp_ctx_set_grad_scale).
        self.gscale = None
        # loss weights above the reference's defaults (orthogonal_l1 weight 0.125, focal alpha 0.25) scale the loss gradients up: take
        # that factor out of the power-of-two gradient scale, so that the headroom under the P16 clamp stays what it is at the defaults
        import math as _math
        ratio = max(self.loss_params["box"][0] / 0.125, self.loss_params["cls"][0] / 0.25, self.loss_params["mask"][0] / 0.25, 1.0)
        self.gscale_adjust = -int(_math.ceil(_math.log2(ratio)))
        from ._lib import MISSING as _missing
        if self.train and self.po and self.arith in ("f16c8", "mixed") and "pp_grad_scale_from_counts" not in _missing:
            self.gscale = torch.ones((2,), dtype=torch.float32, device="cuda")
            for c in self.ctxs:
                ops.set_grad_scale(c.twin(1), self.gscale)  # (bf16-pair gradients are unscaled: only the P16 contexts divide it out)
        self._lane = 0
        self.trunk_lanes = os.environ.get("PP_TRUNK_LANES", "1") != "0"  # backbone shortcut / FPN level 4-5 chains on lane 1
        # Prefix lane (training with conv1 + res2 frozen, the reference's setting): the frozen prefix is a pure function of
        # the input batch -- no weight of it ever changes -- so the prefix of batch i+1 can run while batch i is still in
        # its trunk / heads (forward(next_x=...): software pipelining over steps; the trunk's launches are small and leave
        # most of the chip idle, the prefix is HBM-bound).  Its launches get a stream and a native context of their own; all but the LAST of them (res2c_branch2c, which writes the tensor that res3a's weight
        # gradients read until the end of the step) only touch buffers that nothing else reads.  Opt-in (PP_PREFETCH=1): bit-identical
        # results (tests/test_gpu_pipeline.py), but the bench step gains only 0.5-1 % -- the HBM-bound prefix raises the memory latency
        # that the trunk's small launches are bound by, and released later (beside the heads or the backward: PP_PREFETCH_AFTER) it
        # gains nothing.  Default: the prefix is part of lane 0.
        self.prefix_lane = None
        self.n_prefix_early = 0
        self._prefetched = None
        frozen_prefix = all(not sp.trainable for sp in self.specs if sp.name == "conv1" or sp.name.startswith("res2"))
        if self.train and frozen_prefix and self.conv_mode == "bf16x3" and os.environ.get("PP_PREFETCH", "0") == "1":
            pst = torch.cuda.Stream(device=ctx.device)
            self.streams.append(pst)
            self.ctxs.append(ops.Context(ctx.device, pst))
            if ws_mb > 0:
                self.ctxs[-1].set_workspace(ws_mb << 20)  # (same launch choices -- split-K at small sizes -- as on lane 0: same bits)
            self.prefix_lane = len(self.streams) - 1
        self.acts = OrderedDict()
        self._plane_grads = []
        self.step_count = 0
        self._build_forward()
        self.levels = arch.level_shapes(self.H, self.W, arch.PYRAMID_LEVELS[pyramid])
        assert [tuple(s) for s in self.pyr.shapes] == [tuple(s) for s in self.levels], (self.pyr.shapes, self.levels)
        self.N = sum(h * w for h, w in self.levels) * self.A
        self.M3 = self.levels[0][0] * self.levels[0][1]
        f32 = dict(dtype=torch.float32, device="cuda")
        self.out_box = torch.empty((self.B, self.N, 16), **f32)
        self.out_cls = torch.empty((self.B, self.N, self.C), **f32)
        self.out_mask = torch.empty((self.B, self.M3, self.C), **f32)
        self.anchors_f32 = None
        if train:
            self.counts = torch.zeros((4,), dtype=torch.int32, device="cuda")
            self.loss_sums = torch.zeros((4,), **f32)  # box, cls, mask, l2
            self.gnorm_sq = torch.zeros((1,), **f32)
            self.y_box = torch.zeros((self.B, self.N, 17), **f32)
            self.y_cls = torch.zeros((self.B, self.N, self.C + 1), **f32)
            self.y_mask = torch.zeros((self.B, self.M3, self.C + 1), **f32)
            self.opt = ops.Optimizer(ctx, self.params.descs(), self.params.total)
            self._build_backward()
        self.grad_sync = None  # set by parallel.DataParallel
        self.refresh_planes()

    def close(self):
        """Release the native handles (contexts of every lane, optimizer state); the torch buffers go with the object."""
        opt = getattr(self, "opt", None)
        if opt is not None:
            opt.close()
            self.opt = None
        st = getattr(self, "streams", None)
        if st and getattr(self, "ctxs", None):
            for side in st[1:]:  # (a prefetched prefix may still be running: the buffers are freed in lane 0's order)
                st[0].wait_stream(side)
        for c in getattr(self, "ctxs", []):
            c.close()
        self.ctxs = []
        self.fwd_ops, self.bwd_ops, self.graph_ops = [], [], []

    def _enter(self):
        """order the engine's lane 0 after the caller's current stream (no-op when they are the same stream)"""
        cur = torch.cuda.current_stream(self.ctx.device)
        if cur.cuda_stream != self.streams[0].cuda_stream:
            self.streams[0].wait_stream(cur)
        return cur

    def _leave(self, cur):
        if cur.cuda_stream != self.streams[0].cuda_stream:
            cur.wait_stream(self.streams[0])

    # ------------------------------------------------------------------------------------ forward plan
    def _new_act(self, name, shapes, C, ld=None, needs_grad=False, relu=False, t=None, pl=None, planes=False, fmt=0):
        a = Act(name, self.B, shapes, C, ld, t, needs_grad, relu, pl, planes, fmt)
        self.acts[name] = a
        return a

    def _fmt(self, spec_name):
        """plane format / arithmetic of a layer: 0 = bf16 pairs (bf16x3), 1 = P16 (f16c8)"""
        if self.arith == "f16c8":
            return 1
        if self.arith == "mixed":
            return 0 if spec_name.startswith(("conv1", "res")) else 1
        return 0

    def _convert(self, x, fmt):
        """`x` re-encoded into plane format `fmt` (one pointwise pass; cached per tensor): the boundary between the two arithmetics"""
        if x.pl is None or x.fmt == fmt:
            return x
        cache = x.__dict__.setdefault("_converted", {})
        if fmt in cache:
            return cache[fmt]
        out = self._new_act(x.name + ":fmt%d" % fmt, x.shapes, x.C, x.ld, x.needs_grad, False, planes=True, fmt=fmt)
        lane = self._lane
        ctx = self.ctxs[lane]
        self._push(Op(lambda: ops.convert_planes(ctx, x.pl, x.fmt, out.pl, fmt), "pointwise", "convert:" + x.name, lane=lane), (x,), out)
        self.graph_ops.append(dict(kind="convert", y=out, x=x))
        cache[fmt] = out
        return out

    def _on(self, lane):
        """`with self._on(1): ...` builds the enclosed launches on side lane 1 (lane 0 when PP_LANES=1)"""
        eng = self

        class _Lane(object):
            def __enter__(self_):
                self_.prev = eng._lane
                eng._lane = (lane % eng.n_lanes) if eng.trunk_lanes else eng._lane

            def __exit__(self_, *exc):
                eng._lane = self_.prev
        return _Lane()

    def _push(self, op, reads, writes):
        """append a forward launch and record its dataflow (lane dependencies are derived from it, _link_lanes)"""
        op.reads = tuple(a for a in reads if a is not None)
        if writes is not None:
            writes.prod_ops.append(op)
        self.fwd_ops.append(op)
        return op

    def _link_lanes(self):
        """A launch waits for the producers of its inputs that were enqueued on another lane (HIP events, created once)."""
        pos = {id(o): i for i, o in enumerate(self.fwd_ops)}
        for i, op in enumerate(self.fwd_ops):
            waits = []
            for a in op.reads:
                for p in a.prod_ops:
                    if p is op:  # (an in-place companion launch, e.g. the plane split of a tensor, lists itself)
                        continue
                    assert pos[id(p)] < i, "forward plan out of order: %s reads %s" % (op.name, a.name)
                    if p.lane != op.lane:
                        if p.done_ev is None:
                            p.done_ev = torch.cuda.Event()
                        if p.done_ev not in waits:
                            waits.append(p.done_ev)
            op.waits = tuple(waits)

    def _conv(self, spec_name, x, out_name=None, relu=False, residual=None, out_t=None, out_ld=None, out_pl=None, f32_out=False):
        """f32_out: keep the output in float32 even in planes mode (the head outputs: read by the losses and the export)"""
        s = self.params.specs[spec_name]
        k, st = s.k, s.stride
        cin_eff = 4 if s.cin == 3 else s.cin
        assert x.C == cin_eff, (spec_name, x.C, cin_eff)
        out_shapes, pt, pl = [], 0, 0
        for (h, w) in x.shapes:
            if s.pad == "same":
                oh, ow = -(-h // st), -(-w // st)
                pt, pl = tf_same_pad(h, k, st), tf_same_pad(w, k, st)
            else:
                p = int(s.pad)
                oh, ow = (h + 2 * p - k) // st + 1, (w + 2 * p - k) // st + 1
                pt = pl = p
            out_shapes.append((oh, ow))
        # head outputs are padded to 32 channels in bf16x3 mode (the bf16 data-gradient kernel reduces 32 channels per step)
        ld_y = out_ld if out_ld is not None else _ru(s.cout, 32 if self.conv_mode == "bf16x3" else 16)
        fmt = self._fmt(spec_name)
        x = self._convert(x, fmt)
        if residual is not None:
            residual = self._convert(residual, fmt)
        needs_grad = self.train and (s.trainable or x.needs_grad or (residual is not None and residual.needs_grad))
        bf3 = self.conv_mode == "bf16x3" and s.cin % 32 == 0
        y_planes = bf3 and self.po and not f32_out and ld_y % 8 == 0
        y = self._new_act(out_name or spec_name, out_shapes, s.cout, ld_y, needs_grad, relu, out_t, out_pl, y_planes, fmt)
        ek = self.params.entries[spec_name + "/kernel"]
        desc = ops.make_conv_desc(self.B, x.shapes, out_shapes, cin_eff, s.cout, k, st, pt, pl, x.ld, ld_y, ek["ld"])
        w = self.params.view(self.params.w_eff, spec_name + "/kernel")
        b = self.params.view(self.params.w_eff, spec_name + "/bias")
        lane = self._lane
        ctx = self.ctxs[lane].twin(fmt)
        flops = 2.0 * y.rows * k * k * s.cin * s.cout
        pl = None
        x_cap = None
        if bf3:
            pl = self.planes.get(spec_name)
            if pl is None:
                i16 = dict(dtype=torch.int16, device="cuda")
                need_dg = self.train and x.needs_grad
                pl = dict(desc=desc, w=w, fmt=fmt,
                          fwd_hi=torch.zeros((k * k, s.cout, s.cin), **i16), fwd_lo=torch.zeros((k * k, s.cout, s.cin), **i16),
                          dg_hi=torch.zeros((k * k, s.cin, _ru(s.cout, 32)), **i16) if need_dg else None,
                          dg_lo=torch.zeros((k * k, s.cin, _ru(s.cout, 32)), **i16) if need_dg else None)
                self.planes[spec_name] = pl
            fh, fl = pl["fwd_hi"], pl["fwd_lo"]
            y.producer = s
            cap = None
            if self._wants_capture(s, x):
                if getattr(x, "cap_pl", None) is None:  # the first eligible consumer of x fills the planes
                    x.cap_pl = cap = _new_planes(x.rows, x.ld)
                x_cap = x.cap_pl
            # operands in the format they exist in: planes where a tensor has no float32 copy
            x_t, x_pl = (x.t, None) if x.t is not None else (None, x.pl)
            r_t = r_pl = None
            if residual is not None:
                r_t, r_pl = (residual.t, None) if residual.t is not None else (None, residual.pl)
            self._push(Op(lambda: ops.conv_fwd3(ctx, desc, x_t, fh, fl, b, r_t, relu, y.t, x_pl, y.pl, cap, r_pl), "conv_fwd", spec_name,
                          flops, None, lane), (x, residual), y)
        else:
            assert x.t is not None and (residual is None or residual.t is not None), spec_name
            rt = residual.t if residual is not None else None
            self._push(Op(lambda: ops.conv_fwd(ctx, desc, x.t, w, b, rt, relu, y.t), "conv_fwd", spec_name, flops, None, lane),
                       (x, residual), y)
        self.graph_ops.append(dict(kind="conv", spec=s, x=x, y=y, residual=residual, desc=desc, w=w, flops=flops, planes=pl,
                                   x_cap=x_cap))
        return y

    def _wants_capture(self, s, x):
        return (self.train and self.capture and s.trainable and x.needs_grad and s.k == 3 and s.stride == 1 and (s.pad == "same" or str(s.pad) == "1")
                and s.cin % 64 == 0 and s.cout % 32 == 0 and s.cin >= self.capture_min_cin and x.ld % 8 == 0 and x.t is not None
                and not any(s.name.startswith(t) for t in self.capture_skip) and not self._sparse_layer(s))

    def _sparse_layer(self, s):
        return self.train and s.k == 3 and s.stride == 1 and s.cin % 64 == 0 and any(s.name.startswith(t) for t in self.sparse_bwd)

    def _build_forward(self):
        B, H, W = self.B, self.H, self.W
        ctx = self.ctx
        self.x_in = torch.zeros((B, H, W, 3), dtype=torch.float32, device="cuda")
        if self.prefix_lane is not None:
            self._lane = self.prefix_lane
        pctx = self.ctxs[self._lane]  # the context of the frozen prefix's launches (lane 0, or the prefix lane)
        # (the row-as-tap stem needs the buffer-addressed loop: PP_CONV3_FAST=0 also turns it off)
        self.stem3 = (self.conv_mode == "bf16x3" and _os.environ.get("PP_STEM3", "1") != "0" and
                      _os.environ.get("PP_CONV3_FAST", "1") != "0")
        if self.stem3:
            # the stem on the bf16 path: the packed image sits at (3, 3) of a zero frame and every kernel ROW is one tap of
            # a 7x1 conv over 32 overlapping "channels" (pp_stem7x7s2_fwd_bf16x3)
            Hp, Wp = H + 6, (W + 8 + 1) // 2 * 2
            self.stem_frame = (Hp, Wp)
            x4 = self._new_act("input4", [(Hp, Wp)], 4)
            self._push(Op(lambda: ops.pack_rgb_to_4_padded(pctx, self.x_in, x4.t, Hp, Wp), "pointwise", "pack_rgb", lane=self._lane), (), x4)
            y = self._build_stem3(x4)
        else:
            x4 = self._new_act("input4", [(H, W)], 4)
            self._push(Op(lambda: ops.pack_rgb_to_4(pctx, self.x_in, x4.t), "pointwise", "pack_rgb", lane=self._lane), (), x4)
            y = self._conv("conv1", x4, relu=True)
        (h1, w1) = y.shapes[0]
        ph, pw = (h1 + 1) // 2, (w1 + 1) // 2
        pool = self._new_act("pool1", [(ph, pw)], 64)
        c1 = y
        self._push(Op(lambda: ops.maxpool3x3s2(pctx, B, h1, w1, 64, c1.t, ph, pw, pool.t), "pointwise", "pool1", lane=self._lane), (c1,), pool)
        self.graph_ops.append(dict(kind="stop"))
        y = pool
        blocks = arch.BACKBONE_BLOCKS[self.backbone]
        numerical = {"resnet50": [False] * 4, "resnet101": [False, True, True, False], "resnet152": [False, True, True, False]}[self.backbone]
        stage_out = []
        for stage, n_blocks in enumerate(blocks):
            for block in range(n_blocks):
                pre = "res%d%s" % (stage + 2, arch.block_name(stage, block, numerical[stage]))
                a = self._conv(pre + "_branch2a", y, relu=True)
                b = self._conv(pre + "_branch2b", a, relu=True)
                if block == 0:
                    if self._lane == self.prefix_lane:  # (the prefix is one chain on its own stream)
                        sc = self._conv(pre + "_branch1", y)
                    else:
                        with self._on(1):  # the projection shortcut runs beside the 2a -> 2b chain
                            sc = self._conv(pre + "_branch1", y)
                else:
                    sc = y
                if stage == 0 and block == n_blocks - 1 and self.prefix_lane is not None:
                    self.n_prefix_early = len(self.fwd_ops)  # the launches before this one never touch what the rest of the step reads
                    self._lane = 0
                y = self._conv(pre + "_branch2c", b, out_name=pre, relu=True, residual=sc)
            stage_out.append(y)
        C3, C4, C5 = stage_out[1], stage_out[2], stage_out[3]
        self.C3, self.C4, self.C5 = C3, C4, C5
        if self.pyramid == "sparse":
            pyr, P3 = self._build_sparse_fpn(C3, C4, C5)
        else:
            pyr, P3 = self._build_pyramid_features(C3, C4, C5)
        self.pyr, self.P3 = pyr, P3
        # ---- heads (models/retinanet.py:9-131, shared across levels :224-225; mask on P3 only :296)
        def run_head(prefix, feat):
            y = feat
            for i in range(4):
                y = self._conv("%s_conv%d" % (prefix, i), y, relu=True)
            return self._conv(prefix + "_out", y, f32_out=True)
        self.fwd_fork = len(self.fwd_ops)  # everything before this index is the serial trunk (lane 0)
        self._lane = 0
        self.reg_out = run_head("reg", pyr)
        self._lane = 1 % self.n_lanes
        self.cls_out = run_head("cls", pyr)
        # the mask head (P3 only: 0.6 ms) follows the class head on lane 1 when there are two lanes: the 3D-box head (2.3 ms)
        # has lane 0 to itself and both lanes stay busy for longer (PP_MASK_LANE overrides)
        # (with the sparse forward of the 3D-box head lane 0 is the short one: the mask head goes there)
        self._lane = int(_os.environ.get("PP_MASK_LANE", ("0" if self.sparse_fwd else "1") if self.n_lanes == 2 else "2")) % self.n_lanes
        self.mask_out = run_head("mask", P3)
        self._lane = 0
        # interleave the three chains in enqueue order so that every lane has work from the start
        trunk, heads = self.fwd_ops[: self.fwd_fork], self.fwd_ops[self.fwd_fork:]
        by_lane = [[o for o in heads if o.lane == l] for l in range(self.n_lanes)]
        mixed = []
        for i in range(max(len(c) for c in by_lane)):
            for c in by_lane:
                if i < len(c):
                    mixed.append(c[i])
        self.fwd_ops = trunk + mixed
        self._link_lanes()
        # where in the forward plan the next batch's prefix is released (after the launch of that name; default: as soon as this
        # batch's own prefix has been consumed)
        if self.sparse_fwd:
            self._plan_sparse_forward()
        self.pf_trigger = self.n_prefix_early + 1
        after = _os.environ.get("PP_PREFETCH_AFTER")
        if after and self.prefix_lane is not None:
            idx = [i for i, o in enumerate(self.fwd_ops) if o.name == after and o.lane == 0]
            assert idx, "PP_PREFETCH_AFTER=%s: no such lane-0 launch" % after
            self.pf_trigger = max(self.pf_trigger, idx[0] + 1)

    def _plan_sparse_forward(self):
        """Sparse forward of the 3D-box head in a TRAINING step (opt-in, PP_SPARSE_FWD=1): orthogonal_l1 reads that head's output
        at the anchors with state 1 only (losses.py:332-333), so the step needs reg_out on the 32-row blocks that hold a positive
        anchor, reg_conv3 on those within one pixel of them, ... reg_conv0 within four -- the same block sets the sparse
        backward finds in the gradient.  Per step: one scan of the targets + four dilations (train_step), then every conv of
        the head runs the listed-block launch (pp_ctx_set_row_block_out).  The other rows of the head's tensors keep stale values
        that nothing reads: the loss masks them, the backward's row-block skip never goes there.  forward() / predict paths
        are not affected (the hint is set inside train_step only)."""
        names = ["reg_conv0", "reg_conv1", "reg_conv2", "reg_conv3", "reg_out"]
        nb = (self.pyr.rows + 31) // 32
        dev = dict(device="cuda")
        self._sf_flags = [torch.zeros((nb,), dtype=torch.uint8, **dev) for _ in names]
        self._sf_lists = [torch.zeros((nb + 1,), dtype=torch.int32, **dev) for _ in names]
        self._sf_desc = next(g["desc"] for g in self.graph_ops if g.get("kind") == "conv" and g["spec"].name == "reg_conv1")
        self._sparse_fwd_now = False
        by_name = {o.name: o for o in self.fwd_ops}
        for k, name in enumerate(names):
            op = by_name[name]
            octx = self.ctxs[op.lane].twin(self._fmt(op.name))

            def fn(inner=op.fn, k=k, octx=octx):
                if self._sparse_fwd_now:
                    ops.set_row_block_out(octx, self._sf_flags[k], self._sf_lists[k])
                inner()
            op.fn = fn

    def _sparse_forward_lists(self):
        """the block sets of this step's targets (lane 0; ~25 us)"""
        F = self._sf_flags
        ops.positive_row_blocks(self.ctx, self.pyr.rowspace(), self.A, self.y_box, F[4])
        for k in (3, 2, 1, 0):
            ops.row_block_dilate(self.ctx, self._sf_desc, F[k + 1], F[k])

    def _build_stem3(self, x4):
        B, H, W = self.B, self.H, self.W
        Hp, Wp = self.stem_frame
        s = self.params.specs["conv1"]
        assert (s.k, s.stride, s.cin, int(s.pad)) == (7, 2, 3, 3)
        oh, ow = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
        y = self._new_act("conv1", [(oh, ow)], s.cout, _ru(s.cout, 32), False, True)
        i16 = dict(dtype=torch.int16, device="cuda")
        d = ops.make_conv_desc(B, [(Hp, Wp)], [(oh, ow)], 32, s.cout, 7, 2, 0, 0, 32, y.ld, _ru(s.cout, 16))
        d.kw = 1  # 7 kernel rows x (7 taps x 4 channels -> 32)
        self._stem = dict(desc=d, w=torch.zeros((7 * 32, _ru(s.cout, 16)), dtype=torch.float32, device="cuda"),
                          hi=torch.zeros((7, s.cout, 32), **i16), lo=torch.zeros((7, s.cout, 32), **i16))
        ctx = self.ctxs[self._lane].twin(self._fmt("conv1"))
        bias = self.params.view(self.params.w_eff, "conv1/bias")
        st = self._stem
        flops = 2.0 * y.rows * 49 * 3 * s.cout
        self._push(Op(lambda: ops.stem7x7s2_fwd3(ctx, B, H, W, Hp, Wp, x4.t, st["hi"], st["lo"], s.cout, bias, True, y.t), "conv_fwd", "conv1",
                      flops, None, self._lane), (x4,), y)
        return y

    def _refresh_stem(self):
        """conv1 (frozen) in the row-as-tap layout: plane row ty, column tx * 4 + c  <-  HWIO row (ty * 7 + tx) * 4 + c"""
        st = self._stem
        w = self.params.view(self.params.w_eff, "conv1/kernel")
        st["w"].zero_()
        for ty in range(7):
            st["w"][ty * 32: ty * 32 + 28] = w[ty * 28: (ty + 1) * 28]
        ops.conv_split_weights3(self.ctx.twin(self._fmt("conv1")), st["desc"], st["w"], st["hi"], st["lo"], None, None)

    def _pyramid_buffer(self, level_shapes):
        """P3 | P4 | ... rows in one buffer, so that the shared heads run as ONE multi-level launch"""
        B = self.B
        rows = [B * h * w for h, w in level_shapes]
        pyr_t = None if self.po else torch.empty((sum(rows), 256), dtype=torch.float32, device="cuda")
        cuts, r0 = [], 0
        for n in rows:
            cuts.append((r0, r0 + n))
            r0 += n
        sl = [pyr_t[a:b] if pyr_t is not None else None for a, b in cuts]
        pyr_pl = _new_planes(sum(rows), 256) if self.po else None
        spl = [(pyr_pl[0][a:b], pyr_pl[1][a:b]) if pyr_pl else None for a, b in cuts]
        return pyr_t, pyr_pl, rows, sl, spl

    def _build_sparse_fpn(self, C3, C4, C5):
        """__create_sparceFPN (models/retinanet.py:180-214)"""
        # two chains: level 3 on lane 0, levels 4/5 on lane 1 (their launches are small: 150-600 workgroups each)
        L3 = self._conv("fpn_lat3", C3)
        with self._on(1):
            L4 = self._conv("fpn_lat4", C4)
            L5 = self._conv("fpn_lat5", C5)
            S4 = self._upadd("fpn_sum4", L5, L4)
            M4 = self._conv("fpn_mid4", S4)
        S3 = self._upadd("fpn_sum3", L4, L3)
        M3 = self._conv("fpn_mid3", S3)
        D3 = self._conv("fpn_down3", M3)
        F3 = self._add("fpn_fin3", [M3, L3])
        lv = [L3.shapes[0], L4.shapes[0], L5.shapes[0]]
        pyr_t, pyr_pl, rows, sl, spl = self._pyramid_buffer(lv)
        P3 = self._conv("P3", F3, out_t=sl[0], out_pl=spl[0])
        with self._on(1):
            F4 = self._add("fpn_fin4", [D3, M4, L4])
            D4 = self._conv("fpn_down4", M4)
            P4 = self._conv("P4", F4, out_t=sl[1], out_pl=spl[1])
            F5 = self._add("fpn_fin5", [D4, L5])
            P5 = self._conv("P5", F5, out_t=sl[2], out_pl=spl[2])
        needs = self.train
        pyr = self._new_act("pyramid", lv, 256, 256, needs, False, pyr_t, pyr_pl, pyr_pl is not None, fmt=P3.fmt)
        pyr.producer = getattr(P3, "producer", None)
        pyr.prod_ops = [o for part in (P3, P4, P5) for o in part.prod_ops]
        self.graph_ops.append(dict(kind="alias", y=pyr, parts=[P3, P4, P5], rows=rows))
        return pyr, P3

    def _build_pyramid_features(self, C3, C4, C5):
        """__create_pyramid_features (models/retinanet.py:134-157; P3..P7) / __create_FPN (:160-177; P3..P5)"""
        R5 = self._conv("C5_reduced", C5)
        R4 = self._conv("C4_reduced", C4)
        R3 = self._conv("C3_reduced", C3)
        S4 = self._upadd("td_sum4", R5, R4)   # P5_upsampled + P4   (:136,141)
        S3 = self._upadd("td_sum3", S4, R3)   # P4_upsampled + P3   (:142,147)
        lv = [R3.shapes[0], R4.shapes[0], R5.shapes[0]]
        if self.pyramid == "p3p7":
            s6 = (-(-R5.shapes[0][0] // 2), -(-R5.shapes[0][1] // 2))
            lv += [s6, (-(-s6[0] // 2), -(-s6[1] // 2))]
        pyr_t, pyr_pl, rows, sl, spl = self._pyramid_buffer(lv)
        P3 = self._conv("P3_con", S3, out_t=sl[0], out_pl=spl[0])
        P4 = self._conv("P4_con", S4, out_t=sl[1], out_pl=spl[1])
        P5 = self._conv("P5_con", R5, out_t=sl[2], out_pl=spl[2])
        parts = [P3, P4, P5]
        if self.pyramid == "p3p7":
            P6 = self._conv("P6_con", C5, out_t=sl[3], out_pl=spl[3])           # "3x3 stride-2 conv on C5" (:151)
            R6 = self._relu("P6_relu", P6)                                      # :154
            P7 = self._conv("P7_con", R6, out_t=sl[4], out_pl=spl[4])           # :155
            parts += [P6, P7]
        pyr = self._new_act("pyramid", lv, 256, 256, self.train, False, pyr_t, pyr_pl, pyr_pl is not None, fmt=P3.fmt)
        pyr.producer = getattr(P3, "producer", None)
        pyr.prod_ops = [o for part in parts for o in part.prod_ops]
        self.graph_ops.append(dict(kind="alias", y=pyr, parts=parts, rows=rows))
        return pyr, P3

    def _relu(self, name, x):
        out = self._new_act(name, x.shapes, x.C, x.ld, x.needs_grad, True, planes=self.po, fmt=x.fmt)
        lane = self._lane
        ctx = self.ctxs[lane].twin(x.fmt)
        vx, vo = x.view(), out.view()
        self._push(Op(lambda: ops.relu_fwd_v(ctx, vx, vo), "pointwise", name, lane=lane), (x,), out)
        self.graph_ops.append(dict(kind="relu", y=out, x=x))
        return out

    def _upadd(self, name, src, other):
        (sh, sw), (th, tw) = src.shapes[0], other.shapes[0]
        other = self._convert(other, src.fmt)
        out = self._new_act(name, [(th, tw)], src.C, src.ld, src.needs_grad or other.needs_grad, planes=self.po, fmt=src.fmt)
        lane, B = self._lane, self.B
        ctx = self.ctxs[lane].twin(src.fmt)
        vs, vt, vo = src.view(), other.view(), out.view()
        self._push(Op(lambda: ops.upsample_add_fwd_v(ctx, B, sh, sw, th, tw, src.C, vs, vt, vo), "pointwise", name, lane=lane),
                   (src, other), out)
        self.graph_ops.append(dict(kind="upadd", y=out, src=src, other=other))
        return out

    def _add(self, name, ins):
        ins = [ins[0]] + [self._convert(i, ins[0].fmt) for i in ins[1:]]
        out = self._new_act(name, ins[0].shapes, ins[0].C, ins[0].ld, any(i.needs_grad for i in ins), planes=self.po, fmt=ins[0].fmt)
        lane = self._lane
        ctx = self.ctxs[lane].twin(ins[0].fmt)
        va, vb, vc = ins[0].view(), ins[1].view(), (ins[2].view() if len(ins) > 2 else None)
        vo = out.view()
        self._push(Op(lambda: ops.add_n_v(ctx, va, vb, vc, vo), "pointwise", name, lane=lane), ins, out)
        self.graph_ops.append(dict(kind="add", y=out, ins=ins))
        return out

    # ------------------------------------------------------------------------------------ backward plan
    def _new_grad(self, rows, ld, fmt=0):
        """an uninitialised gradient matrix in the mode's storage format (fmt: the plane format of the tensor it belongs to)"""
        if self.po and ld % 8 == 0:
            g = Grad(None, _new_planes(rows, ld), fmt)
            self._plane_grads.append(g)  # (p16_stats() looks at the P16 ones after a step)
            return g
        return Grad(torch.empty((rows, ld), dtype=torch.float32, device="cuda"))

    def _finalize(self, act, producer=None):
        """Sum the gradient contributions of `act`; the ReLU mask (act > 0) is folded into the last
        data-gradient launch.  Returns the gradient w.r.t. the pre-activation (a Grad), or None.
        producer: the op whose backward will read that gradient (decides whether a sparse launch may leave it unfilled)."""
        ctx = self.ctx.twin(act.fmt)  # (every contribution to a tensor is in that tensor's plane format: its consumers read it so)
        grads = [c[1] for c in act.contribs if c[0] in ("tensor", "masked")]  # masked: the ReLU of `act` is already applied
        dgrads = [c for c in act.contribs if c[0] == "dgrad"]
        if any(c[0] == "masked" for c in act.contribs):
            assert len(act.contribs) == 1, act.name
            return grads[0]
        if not grads and not dgrads:
            return None
        acc = None
        if len(grads) == 1:
            acc = grads[0]
        elif len(grads) > 1:
            acc = self._new_grad(act.rows, act.ld, act.fmt)
            srcs, rest = grads[:3], grads[3:]
            while True:
                va, vb, vc = srcs[0].view(), (srcs[1].view() if len(srcs) > 1 else None), (srcs[2].view() if len(srcs) > 2 else None)
                vo = ops.tview(acc.t, acc.pl)
                self.bwd_ops.append(Op(lambda va=va, vb=vb, vc=vc, vo=vo: ops.add_n_v(ctx, va, vb, vc, vo), "pointwise", "add:" + act.name))
                if not rest:
                    break
                srcs, rest = [acc] + rest[:2], rest[2:]  # in-place accumulate (pointwise: safe)
        if act.relu and not dgrads:
            raise NotImplementedError("relu output %s without a data-gradient consumer" % act.name)
        if not dgrads:
            return acc
        for i, (_, op, gy) in enumerate(dgrads):
            last = i == len(dgrads) - 1
            mask = act if (act.relu and last) else None
            pl = op.get("planes")
            if pl is not None and pl["dg_hi"] is not None:
                gcap = op.get("g_cap")
                sk = op.get("skip")
                # a sparse data gradient (row-block skip) that only adds to the sum of the earlier ones works in place on that sum:
                # the rows no non-zero reaches keep their value and nothing else is moved (`acc` of i > 0 is private to this loop)
                in_place = (sk is not None and i > 0 and mask is None and acc is not None and gcap is None
                            and _os.environ.get("PP_SPARSE_INPLACE", "1") != "0")
                out = acc if in_place else self._new_grad(act.rows, act.ld, act.fmt)
                # Lazy: the only readers of this gradient are the backward launches of a sparse layer (its scan restricted to the
                # blocks this launch computes, its listed-block weight gradient, its listed-block data gradient) -- the rows outside
                # those blocks are never looked at, so the 103 MB fill pass of zeros is not made (3D-box head: four per step)
                lazy_out = (sk is not None and len(dgrads) == 1 and acc is None and gcap is None and gy.pl is not None and out.pl is not None
                            and self._lazy_reader(producer) and _os.environ.get("PP_SPARSE_WITHIN", "1") != "0"
                            and _os.environ.get("PP_SPARSE_LAZY", "1") != "0" and _os.environ.get("PP_SPARSE_DGRAD", "22") != "0")
                lazy_in = bool(gy.lazy)
                # every operand in the format it exists in (planes where there is no float32 copy)
                dy_t, dy_pl = (None, gy.pl) if gy.pl is not None else (gy.t, None)
                a_t = a_pl = m_t = m_hi = None
                if acc is not None:
                    a_t, a_pl = (acc.t, None) if acc.t is not None else (None, acc.pl)
                if mask is not None:
                    m_t, m_hi = (mask.t, None) if mask.t is not None else (None, mask.pl[0])
                if gcap is not None:
                    assert dy_t is not None
                self.bwd_ops.append(Op(lambda d=op["desc"], dy_t=dy_t, dy_pl=dy_pl, dh=pl["dg_hi"], dl=pl["dg_lo"], a_t=a_t, a_pl=a_pl, m_t=m_t,
                                       m_hi=m_hi, out=out, gcap=gcap, sk=sk, lazy_out=lazy_out, lazy_in=lazy_in:
                                       ops.conv_bwd_data3(ctx, d, dy_t, dh, dl, a_t, m_t, out.t, dy_pl, out.pl, gcap, sk, a_pl, m_hi,
                                                          lazy_out=lazy_out, lazy_in=lazy_in), "conv_dgrad",
                                       op["spec"].name, op["flops"]))
                # a row-block-skip launch without addend leaves zeros (lazy: nothing at all) outside the blocks it flags in the second
                # half of its scratch
                out.within = sk[0][sk[0].numel() // 2:] if (sk is not None and acc is None and _os.environ.get("PP_SPARSE_WITHIN", "1") != "0") else None
                out.lazy = lazy_out
                assert not lazy_out or out.within is not None
                pw = op.pop("pending_wgrad", None)
                if pw is not None:  # the layer's weight gradient reads the planes this launch has just written
                    self.bwd_ops.append(pw)
            else:
                assert gy.t is not None and (acc is None or acc.t is not None) and (mask is None or mask.t is not None), act.name
                out = Grad(torch.empty((act.rows, act.ld), dtype=torch.float32, device="cuda"))
                self.bwd_ops.append(Op(lambda d=op["desc"], gy=gy, w=op["w"], a=(acc.t if acc is not None else None),
                                       m=(mask.t if mask is not None else None), out=out:
                                       ops.conv_bwd_data(ctx, d, gy.t, w, a, m, out.t), "conv_dgrad", op["spec"].name, op["flops"]))
            acc = out
        return acc

    def _lazy_reader(self, producer):
        """is `producer` (the op that made the activation whose gradient is being formed) a layer whose whole backward goes by
        row-block flags?  a sparse 3x3 stride-1 conv without residual, planes on both sides, not the first layer of the graph"""
        if producer is None or producer.get("kind") != "conv" or producer.get("residual") is not None:
            return False
        s_, pl_, x_, y_ = producer["spec"], producer.get("planes"), producer["x"], producer["y"]
        # (the listed-block launches are forms of the buffer-addressed loops: their conditions, csrc/conv3.hip igemm3_fast_ok /
        # launch_wgrad3 -- the C side refuses a lazy operand where they do not hold, this keeps the plan away from that)
        geo = (_os.environ.get("PP_CONV3_FAST", "1") != "0" and min(h * w for h, w in x_.shapes) >= 32 and x_.rows < (1 << 24)
               and x_.rows * x_.ld * 4 < (1 << 31) and y_.rows * y_.ld * 4 < (1 << 31) and list(x_.shapes) == list(y_.shapes))
        return self._sparse_layer(s_) and pl_ is not None and pl_["dg_hi"] is not None and x_.pl is not None and y_.pl is not None and geo

    def _build_backward(self):
        ctx, P = self.ctx, self.params
        f32 = dict(dtype=torch.float32, device="cuda")
        self.g_reg = torch.zeros((self.reg_out.rows, self.reg_out.ld), **f32)
        self.g_cls = torch.zeros((self.cls_out.rows, self.cls_out.ld), **f32)
        self.g_mask = torch.zeros((self.mask_out.rows, self.mask_out.ld), **f32)
        for out, gt in ((self.reg_out, self.g_reg), (self.cls_out, self.g_cls), (self.mask_out, self.g_mask)):
            g = Grad(gt, None, out.fmt)
            if self.po and out.ld % 8 == 0:
                # the loss kernels write float32; the head's last conv takes its dy (bwd-data, bwd-weight) from planes -- in the
                # format of that conv, multiplied by the gradient scale where that format is P16
                g.pl = _new_planes(out.rows, out.ld)
                sctx, sc = ctx.twin(out.fmt), (self.gscale if out.fmt == 1 else None)
                self.bwd_ops.append(Op(lambda gt=gt, pl=g.pl, sctx=sctx, sc=sc: ops.split_planes3(sctx, gt, pl[0], pl[1], sc), "pointwise",
                                       "split:" + out.name))
            out.contribs.append(("tensor", g))
        for op in reversed(self.graph_ops):
            kind = op["kind"]
            if kind == "stop":
                break
            y = op["y"]
            if not y.needs_grad:
                continue
            g = self._finalize(y, op)
            if g is None:
                continue
            if kind == "convert":
                # the boundary of the two arithmetics, backwards: the gradient of the re-encoded copy (P16: carries 2^G) returns to
                # the original tensor's format (bf16 pairs: unscaled), or the other way round
                x = op["x"]
                if x.needs_grad:
                    gx = self._new_grad(x.rows, x.ld, x.fmt)
                    assert g.pl is not None and gx.pl is not None, x.name
                    sc, idx = (self.gscale, 1) if (g.fmt == 1 and x.fmt == 0) else ((self.gscale, 0) if (g.fmt == 0 and x.fmt == 1) else (None, 0))
                    # a ReLU output that only this copy consumes (C5) has no data-gradient launch to carry its mask: it goes here
                    lone = x.relu and not any(o is not op and o.get("kind") == "conv" and o.get("x") is x for o in self.graph_ops)
                    mhi = x.pl[0] if lone else None
                    self.bwd_ops.append(Op(lambda g=g, gx=gx, sc=sc, idx=idx, mhi=mhi: ops.convert_planes(ctx, g.pl, g.fmt, gx.pl, gx.fmt, sc, idx, mhi),
                                           "pointwise", "convert_bwd:" + x.name))
                    x.contribs.append(("masked" if lone else "tensor", gx))
                continue
            if kind == "conv":
                s, x = op["spec"], op["x"]
                lctx = ctx.twin(y.fmt)  # the layer's format = its output's
                if self._sparse_layer(s) and g.contiguous() and len(g.shape()) == 2:
                    nb = (g.shape()[0] + 31) // 32
                    skip = op["skip"] = (torch.zeros((2 * nb,), dtype=torch.uint8, device="cuda"),   # flags | bwd-data scratch
                                         torch.zeros((2 * (nb + 1),), dtype=torch.int32, device="cuda"))
                    cols = min((y.C + 3) // 4 * 4, g.shape()[1])
                    if g.t is not None:
                        self.bwd_ops.append(Op(lambda gt=g.t, cols=cols, skip=skip, lctx=lctx: ops.row_block_list(lctx, gt, cols, skip[0], skip[1]),
                                               "pointwise", "rowblocks:" + s.name))
                    else:
                        self.bwd_ops.append(Op(lambda gp=g.pl, cols=cols, skip=skip, within=g.within, lctx=lctx:
                                               ops.row_block_list_planes(lctx, gp, cols, skip[0], skip[1], within),
                                               "pointwise", "rowblocks:" + s.name))
                if s.trainable:
                    dw = P.view(P.grad, s.name + "/kernel")
                    db = P.view(P.grad, s.name + "/bias") if s.bias else None
                    ek = P.entries[s.name + "/kernel"]
                    eb = P.entries[s.name + "/bias"]
                    wr = (ek["offset"], eb["offset"] + eb["count"])
                    wl = 1 % self.n_lanes
                    wctx = self.ctxs[wl].twin(y.fmt)
                    pl_ = op.get("planes")
                    if (op.get("x_cap") is not None and x.needs_grad and pl_ is not None and pl_["dg_hi"] is not None
                            and g.pl is None and g.t.shape == (y.rows, y.ld) and g.t.is_contiguous()):
                        gc = op["g_cap"] = _new_planes(y.rows, y.ld)
                        fn = lambda d=op["desc"], x=x, g=g, dw=dw, db=db, wctx=wctx, xp=op["x_cap"], gp=gc: \
                            ops.conv_bwd_weight3(wctx, d, x.t, g.t, dw, db, xp, gp)
                        op["pending_wgrad"] = Op(fn, "conv_wgrad", s.name, op["flops"], wr, wl)  # enqueued after the layer's dgrad
                        fn = None
                    elif self.conv_mode == "bf16x3" and s.cin % 64 == 0:
                        # both operands as planes where both exist as planes (no conversion in the loop), else both as float32
                        both = x.pl is not None and g.pl is not None
                        assert both or (x.t is not None and g.t is not None), "weight gradient of %s: operands in different formats" % s.name
                        sk = op.get("skip")
                        assert not g.lazy or (both and sk is not None), s.name
                        fn = lambda d=op["desc"], xt=(None if both else x.t), gt=(None if both else g.t), dw=dw, db=db, wctx=wctx, \
                            xp=(x.pl if both else None), gp=(g.pl if both else None), sk=sk, lz=g.lazy: \
                            ops.conv_bwd_weight3(wctx, d, xt, gt, dw, db, xp, gp, sk, lazy_in=lz)
                    else:
                        assert x.t is not None and g.t is not None, s.name
                        fn = lambda d=op["desc"], xt=x.t, g=g, dw=dw, db=db, wctx=wctx: ops.conv_bwd_weight(wctx, d, xt, g.t, dw, db)
                    if fn is not None:
                        self.bwd_ops.append(Op(fn, "conv_wgrad", s.name, op["flops"], wr, wl))
                if x.needs_grad:
                    x.contribs.append(("dgrad", op, g))
                r = op["residual"]
                if r is not None and r.needs_grad:
                    r.contribs.append(("tensor", g))
            elif kind == "add":
                for i in op["ins"]:
                    if i.needs_grad:
                        i.contribs.append(("tensor", g))
            elif kind == "upadd":
                src, other = op["src"], op["other"]
                if other.needs_grad:
                    other.contribs.append(("tensor", g))
                if src.needs_grad:
                    gs = self._new_grad(src.rows, src.ld, src.fmt)
                    (sh, sw), (th, tw) = src.shapes[0], other.shapes[0]
                    vg, vs = g.view(), ops.tview(gs.t, gs.pl)
                    uctx = ctx.twin(src.fmt)
                    self.bwd_ops.append(Op(lambda vg=vg, vs=vs, sh=sh, sw=sw, th=th, tw=tw, c=src.C, uctx=uctx:
                                           ops.upsample_add_bwd_v(uctx, self.B, sh, sw, th, tw, c, vg, None, vs), "pointwise", "upbwd:" + src.name))
                    src.contribs.append(("tensor", gs))
            elif kind == "relu":
                # `g` already carries the mask (y > 0): _finalize folded it into the data-gradient launch that produced it
                if op["x"].needs_grad:
                    op["x"].contribs.append(("tensor", g))
            elif kind == "alias":
                r0 = 0
                for part, n in zip(op["parts"], op["rows"]):
                    part.contribs.append(("tensor", g.rows(r0, r0 + n)))
                    r0 += n
        left = [op["spec"].name for op in self.graph_ops if op.get("pending_wgrad") is not None]
        assert not left, "weight gradients waiting for a bwd-data launch that never came: %s" % left

    # ------------------------------------------------------------------------------------ execution
    def _fork(self, lanes):
        """side lanes wait for everything enqueued on lane 0 so far"""
        if self.n_lanes == 1:
            return
        ev = torch.cuda.Event()
        ev.record(self.streams[0])
        for l in lanes:
            self.streams[l].wait_event(ev)

    def _join(self, lanes):
        """lane 0 waits for everything enqueued on the side lanes so far"""
        for l in lanes:
            ev = torch.cuda.Event()
            ev.record(self.streams[l])
            self.streams[0].wait_event(ev)

    RESIDENT = "resident"  # next_x: the next batch is what x_in holds now (bench.py: the synthetic batch stays in HBM)

    def _load_input(self, inp, lane):
        """the batch into the stem's input on `lane`'s stream: ("f32", x or None) = NHWC float32 into x_in (pack_rgb follows),
        ("u8", images, sizes) = the uint8 kernel (mean subtraction + padding + packing in one launch, no pack_rgb)"""
        if inp[0] == "f32":
            if inp[1] is not None:
                with torch.cuda.stream(self.streams[lane]):
                    self.x_in.copy_(inp[1])
            return
        x4 = self.acts["input4"]
        xd = inp[1]
        with torch.cuda.stream(self.streams[lane]):  # (temporaries belong to the stream whose kernels use them)
            if len(inp) > 3 and inp[3] is not None:  # augmentation: warp on the device (utils/image.py:207-214)
                xd = ops.warp_affine_u8(self.ctxs[lane], xd, inp[3], "linear", inp[4], inp[5])
            if self.stem3:
                ops.preprocess_caffe_u8_padded(self.ctxs[lane], xd, inp[2], x4.t, *self.stem_frame)
            else:
                ops.preprocess_caffe_u8(self.ctxs[lane], xd, inp[2], x4.t)

    @staticmethod
    def _input_key(inp):
        """(kind, the batch tensor itself -- held, so that its identity cannot be recycled -- or RESIDENT)"""
        return (inp[0], Engine.RESIDENT if inp[1] is None else inp[1])

    @staticmethod
    def _same_input(a, b):
        return a is not None and a[0] == b[0] and a[1] is b[1]

    def _run_fwd_ops(self, lo, hi, skip_pack):
        streams = self.streams
        for i in range(lo, hi):
            op = self.fwd_ops[i]
            st = streams[op.lane]
            for ev in op.waits:
                st.wait_event(ev)
            if not (skip_pack and i == 0):
                op.fn()
            if op.done_ev is not None:
                op.done_ev.record(st)

    def _forward(self, inp, nxt):
        """inp / nxt: this batch and (optionally) the next one, as _load_input takes them.  With a prefix lane the early prefix
        of `nxt` is enqueued on the prefix stream as soon as this batch's prefix has been consumed; the next call that
        brings the same batch finds it done (or running) and starts behind it."""
        cur = self._enter()
        lp, ne = self.prefix_lane, self.n_prefix_early
        assert self.fwd_ops[0].name == "pack_rgb"
        pre, self._prefetched = self._prefetched, None
        start = 0
        if lp is not None and self._same_input(pre, self._input_key(inp)):
            start = ne  # the early prefix of this batch is already on the prefix stream
        else:
            if lp is not None:
                self._fork([lp])  # (also orders this batch's prefix behind a prefetched one that is not used)
            self._load_input(inp, lp if lp is not None else 0)
        trig = self.pf_trigger if (lp is not None and nxt is not None) else ne + 1
        self._run_fwd_ops(start, trig if lp is not None else len(self.fwd_ops), inp[0] == "u8")
        if lp is not None:
            if nxt is not None:
                # the last prefix launch (lane 0) has read the prefix lane's buffers: the next batch may overwrite them
                ev = torch.cuda.Event()
                ev.record(self.streams[0])
                self.streams[lp].wait_event(ev)
                nxt = ("f32", None) if nxt[1] is Engine.RESIDENT else nxt
                self._load_input(nxt, lp)
                self._run_fwd_ops(0, ne, nxt[0] == "u8")
                self._prefetched = self._input_key(nxt)
            self._run_fwd_ops(trig, len(self.fwd_ops), False)
        self._join(list(range(1, self.n_lanes)))
        self._leave(cur)

    def forward(self, x=None, next_x=None):
        """x: NHWC float32 batch (None: what x_in holds).  next_x: the batch of the NEXT call (the same tensor object, unchanged
        until then; Engine.RESIDENT = x_in as it is) -- its frozen prefix runs beside this batch's trunk."""
        self._forward(("f32", x), None if next_x is None else ("f32", next_x))

    def forward_u8(self, images_u8, sizes_hw=None, transforms=None, border="replicate", cval=0, next_batch=None):
        """Forward from a uint8 BGR batch [B,H,W,3] on the device: mean subtraction, zero padding and channel packing run
        in one kernel in place of the host-side preprocess_image / compute_inputs (4x less host->device traffic).
        transforms: one augmentation matrix per image (the batch is warped on the device first).
        next_batch: dict(images_u8=, sizes_hw=, transforms=, border=, cval=) of the NEXT call (same tensor object) -> its prefix is prefetched."""
        full = [(self.H, self.W)] * self.B
        nxt = None
        if next_batch is not None:
            nb = next_batch
            nxt = ("u8", nb["images_u8"], nb.get("sizes_hw") or full, nb.get("transforms"), nb.get("border", "replicate"), nb.get("cval", 0))
        self._forward(("u8", images_u8, sizes_hw or full, transforms, border, cval), nxt)

    def export_outputs(self):
        """Keras prediction-model outputs (models/retinanet.py:302-335): [boxes3D, cls probs, mask probs]."""
        ctx = self.ctx
        cur = self._enter()
        ops.export_head(ctx, self.pyr.rowspace(), self.A, 16, self.reg_out.t, False, self.out_box)
        ops.export_head(ctx, self.pyr.rowspace(), self.A, self.C, self.cls_out.t, True, self.out_cls)
        ops.export_head(ctx, self.P3.rowspace(), 1, self.C, self.mask_out.t, True, self.out_mask)
        self._leave(cur)
        return self.out_box, self.out_cls, self.out_mask

    def anchors_device_f32(self):
        if self.anchors_f32 is None:
            p = self.anchor_params
            base = np.stack([ops.generate_base_anchors(sz, p.ratios, p.scales) for sz in p.sizes[: len(self.levels)]])
            self.anchors_f32 = ops.anchors_shift(self.ctx, self.levels, p.strides[: len(self.levels)], base, torch.float32)
        return self.anchors_f32

    def predict_on_batch(self, x):
        """-> (boxes3D (B,N,16), scores (B,N,C), mask (B,HW/64,C)) device tensors."""
        self.forward(x)
        reg, cls, mask = self.export_outputs()
        cur = self._enter()
        with torch.cuda.stream(self.streams[0]):  # (the result tensor is allocated for the stream that writes it)
            boxes3d = ops.box3d_decode(self.ctx, self.anchors_device_f32(), reg)
        self._leave(cur)
        return boxes3d, cls, mask

    def set_targets(self, y_box, y_cls, y_mask):
        cur = self._enter()
        with torch.cuda.stream(self.streams[0]):
            self.y_box.copy_(y_box)
            self.y_cls.copy_(y_cls)
            self.y_mask.copy_(y_mask)
        self._leave(cur)

    def loss_and_backward(self):
        """Counts -> (DP: global counts) -> fused loss fwd+bwd -> backward plan.  Leaves gradients in params.grad."""
        ctx, P = self.ctx, self.params
        cur = self._enter()
        with torch.cuda.stream(self.streams[0]):
            P.grad.zero_()
            self.counts.zero_()
            self.loss_sums.zero_()
            ops.count_positives(ctx, self.y_box, self.y_cls, self.y_mask, self.counts)
            if self.grad_sync is not None:
                self.grad_sync.reduce_counts(self.counts)
            if self.gscale is not None:
                ops.grad_scale_from_counts(ctx, self.counts[0:3], self.gscale, self.gscale_adjust)
        (bw, bs), (ca, cg), (ma, mg) = self.loss_params["box"], self.loss_params["cls"], self.loss_params["mask"]
        ops.orth_l1(ctx, self.pyr.rowspace(), self.A, self.reg_out.t, self.y_box, bw, bs, self.counts[0:1], 1.0,
                    self.loss_sums[0:1], self.g_reg)
        ops.focal(ctx, self.pyr.rowspace(), self.A, self.C, self.cls_out.t, self.y_cls, ca, cg, self.counts[1:2], 1.0,
                  self.loss_sums[1:2], self.g_cls)
        ops.focal(ctx, self.P3.rowspace(), 1, self.C, self.mask_out.t, self.y_mask, ma, mg, self.counts[2:3], 1.0,
                  self.loss_sums[2:3], self.g_mask)
        sync = self.grad_sync
        for i, op in enumerate(self.bwd_ops):
            if op.lane:
                self._fork([op.lane])  # the weight gradient needs the gradient tensor lane 0 has just produced
            op()
            if sync is not None:
                sync.after_bwd_op(i, self.streams[op.lane])
        self._join(list(range(1, self.n_lanes)))
        self._leave(cur)

    def refresh_planes(self, only_trainable=False):
        """Re-split the effective weights into the bf16 (hi, lo) planes of the bf16x3 kernels (one launch)."""
        key = "_split_trainable" if only_trainable else "_split_all"
        if not only_trainable and getattr(self, "stem3", False):
            self._refresh_stem()
        batches = getattr(self, key, None)
        if batches is None:  # one job table per plane format (the split kernel encodes in its context's format)
            batches = []
            for fmt in (0, 1):
                jobs = [(pl["desc"], pl["w"], pl["fwd_hi"], pl["fwd_lo"], pl["dg_hi"], pl["dg_lo"]) for name, pl in self.planes.items()
                        if pl.get("fmt", 0) == fmt and (not only_trainable or self.params.specs[name].trainable)]
                if jobs:
                    batches.append((fmt, ops.SplitWeightsBatch(jobs)))
            setattr(self, key, batches)
        for fmt, batch in batches:
            batch.run(self.ctx.twin(fmt))

    def optimizer_step(self):
        P = self.params
        cur = self._enter()
        self.step_count += 1
        self.opt.grad_norm(P.w_master, P.grad, P.scales, self.gnorm_sq, self.loss_sums[3:4])
        self.opt.adam_step(P.w_master, P.w_eff, P.grad, P.scales, P.m, P.v, self.gnorm_sq, self.lr, self.beta1, self.beta2,
                           self.eps, self.clipnorm, self.step_count)
        self.refresh_planes(only_trainable=True)
        self._leave(cur)

    def train_step(self, x=None, targets=None, next_x=None):
        """One optimisation step (Keras train_on_batch): fwd + losses + bwd + clipnorm-Adam.  Returns nothing;
        read `losses()` afterwards (a device->host copy) when the values are wanted.  next_x: see forward()."""
        if targets is not None:
            self.set_targets(*targets)
        self._train_forward(lambda: self.forward(x, next_x))
        self.loss_and_backward()
        if self.grad_sync is not None:
            self.grad_sync.finish()
        self.optimizer_step()

    def _train_forward(self, run_forward):
        """the forward pass of a training step: with PP_SPARSE_FWD=1 the 3D-box head computes the blocks its loss reads only"""
        if not self.sparse_fwd:
            return run_forward()
        cur = self._enter()
        self._sparse_forward_lists()
        self._leave(cur)
        self._sparse_fwd_now = True
        try:
            run_forward()
        finally:
            self._sparse_fwd_now = False

    def train_step_from_annotations(self, images_u8, annotations, image_group=None, transforms=None, border="replicate", cval=0,
                                    next_batch=None):
        """The lean feed of one optimisation step: a uint8 BGR batch [B,H,W,3] (cuda, or pinned host memory) and the raw
        annotation dicts of preprocessing/generator.py:142-226 (bboxes, labels, poses, segmentations, cam_params, mask,
        mask_ids).  Mean subtraction + packing (image.py:58-60, generator.py:320-336) and target assignment
        (utils/anchors.py:72-287) run on the device: 7.4 MB cross PCIe per batch of 8 instead of 87.9 MB.
        transforms: one 2x3 / 3x3 augmentation matrix per image (utils/transform.py:random_transform): the image is warped on the
        device like apply_transform (utils/image.py:207-214: bilinear, border = TransformParameters.fill_mode: 'replicate' for
        the default 'nearest', or 'constant' with cval) and the id mask like apply_transform2mask; the caller has already
        moved boxes / poses (generator.py:252-286 does that on the host: a few numbers per object).
        next_batch: dict(images_u8=, image_group=, transforms=, border=, cval=) of the NEXT call (the generator's look-ahead; the
        same images_u8 object must come back): its upload, warp and frozen prefix run beside this step (forward_u8)."""
        from .utils import anchors as UA
        if getattr(self, "_anchors_f64", None) is None:
            self._anchors_f64 = UA.anchors_for_shape_device((self.H, self.W), pyramid_levels=list(arch.PYRAMID_LEVELS[self.pyramid]),
                                                            anchor_params=self.anchor_params)

        def sizes(group):
            return [(self.H, self.W)] * self.B if group is None else [(int(im.shape[0]), int(im.shape[1])) for im in group]

        held, self._next_dev = getattr(self, "_next_dev", None), None
        if held is not None and held[0] is images_u8:  # the host tensor itself is held: an id() could be reused by a new batch
            xd = held[1]  # uploaded by the previous call's look-ahead
        else:
            xd = images_u8 if images_u8.is_cuda else images_u8.cuda(non_blocking=True)
        nb = None
        if next_batch is not None and self.prefix_lane is not None:
            nimg = next_batch["images_u8"]
            nxd = nimg if nimg.is_cuda else nimg.cuda(non_blocking=True)
            self._next_dev = (nimg, nxd)
            nb = dict(images_u8=nxd, sizes_hw=sizes(next_batch.get("image_group")), transforms=next_batch.get("transforms"),
                      border=next_batch.get("border", "replicate"), cval=next_batch.get("cval", 0))
        if image_group is None:
            image_group = [np.empty((self.H, self.W, 3), np.uint8)] * self.B  # only the shapes are read
        self.set_targets(*UA.anchor_targets_bbox_device(self._anchors_f64, image_group, annotations, self.C, mask_transforms=transforms))
        self._train_forward(lambda: self.forward_u8(xd, sizes(image_group), transforms, border, cval, next_batch=nb))
        self.loss_and_backward()
        if self.grad_sync is not None:
            self.grad_sync.finish()
        self.optimizer_step()

    def export_optimizer_state(self):
        """{'iterations': steps taken, 'm': {key: array}, 'v': {key: array}} -- Adam's state in the Keras layout of the weights"""
        assert self.train
        torch.cuda.synchronize()
        return dict(iterations=int(self.step_count), m=self.params.export_trainable(self.params.m), v=self.params.export_trainable(self.params.v))

    def load_optimizer_state(self, state):
        assert self.train
        torch.cuda.synchronize()  # (host -> device copies on the caller's stream: nothing of a running step may still read m / v)
        self.params.import_trainable(self.params.m, state["m"])
        self.params.import_trainable(self.params.v, state["v"])
        self.step_count = int(state["iterations"])
        torch.cuda.synchronize()

    def p16_stats(self):
        """Where the halves of this plan's P16 tensors sit in the format's range, after the last forward / train step
        (pp_planes_stats): {'activations' | 'gradients': {'elements', 'nonzero', 'clamped', 'subnormal', 'max_abs'}} plus
        'grad_scale_log2'.
        `clamped` counts halves AT the encode's clamp (|x| >= 28 672: the P16 encode saturates silently), `subnormal` non-zero
        halves below 2^-14 (the element has lost significand bits: harmless for the many near-zero elements of a gradient -- their
        absolute error stays 2^-25 -- and a warning only when `max_abs`, the largest |half| of the group, is small itself: then the
        power-of-two gradient scale 2^G is too small).  Lazily
        filled sparse gradients are read inside their flagged blocks only.  An audit pass over HBM (~0.3 ms per GB): never timed."""
        torch.cuda.synchronize()
        ctx1 = self.ctx.twin(1)
        out = {}
        for kind, items in (("activations", [(a.pl, None) for a in self.acts.values() if a.pl is not None and a.fmt == 1]),
                            ("gradients", [(g.pl, (g.within if g.lazy else None)) for g in self._plane_grads if g.pl is not None and g.fmt == 1])):
            st = torch.zeros((5,), dtype=torch.int64, device="cuda")
            spans = []  # byte ranges already counted (the pyramid levels are slices of one buffer: largest tensors first)
            for pl, within in sorted(items, key=lambda it: -it[0][0].shape[0]):
                ld = ops.planes_ld(pl)
                a = pl[0].data_ptr()
                b = a + pl[0].shape[0] * ld * 4
                if pl[0].shape[0] == 0 or any(a >= lo and b <= hi for lo, hi in spans):
                    continue
                spans.append((a, b))
                ops.planes_stats(ctx1, pl, ld, st, within)
            v = st.cpu().numpy()
            out[kind] = dict(elements=int(v[0]), nonzero=int(v[1]), clamped=int(v[2]), subnormal=int(v[3]),
                             max_abs=float(np.array([int(v[4])], np.uint16).view(np.float16)[0]))
        if self.gscale is not None:
            out["grad_scale_log2"] = float(torch.log2(self.gscale[0]).cpu())
        return out

    def losses(self):
        v = self.loss_sums.detach().cpu().numpy()
        return {"3Dbox": float(v[0]), "cls": float(v[1]), "mask": float(v[2]), "l2": float(v[3]), "total": float(v.sum())}
