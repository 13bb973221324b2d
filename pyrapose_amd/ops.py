"""Thin torch-tensor wrappers over the C ABI (include/pyrapose_hip.h).

Every function takes CUDA(ROCm) float32/float64/int tensors, hands raw device pointers to the HIP
library on the ctx stream, and raises on a non-zero status.  torch is used only for memory and streams.
"""
import ctypes as C

import numpy as np
import torch

from ._lib import ConvDesc, ParamDesc, RowSpace, TView, check, lib


def _ptr(t):
    if t is None:
        return None
    assert t.is_cuda, "pyrapose_amd ops need device tensors"
    return C.c_void_p(t.data_ptr())


class Context:
    """One per (process, device, stream): wraps pp_ctx."""

    def __init__(self, device=0, stream=None):
        if not torch.cuda.is_available():
            raise RuntimeError("pyrapose_amd: no GPU visible (torch.cuda.is_available() is False); "
                               "the HIP path has no CPU fallback")
        self.device = int(device)
        torch.cuda.set_device(self.device)
        self.handle = C.c_void_p()
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        self.stream = s
        check(lib.pp_ctx_create(C.byref(self.handle), self.device, C.c_void_p(s.cuda_stream)), None, "pp_ctx_create")

        self.planes_fmt = 0
        self._twin = None

    def use_stream(self, stream):
        # (the twin points back at this context: bind both handles here instead of forwarding, which would never return)
        for c in (self, self._twin):
            if c is not None:
                c.stream = stream
                check(lib.pp_ctx_set_stream(c.handle, C.c_void_p(stream.cuda_stream)), c.handle, "pp_ctx_set_stream")

    def twin(self, fmt):
        """This context for plane format `fmt` (0 bf16 pairs / bf16x3, 1 P16 / f16c8): itself, or -- created on first use -- a
        second context on the same stream, with the same split-K scratch, whose launches read and write the other format."""
        if int(fmt) == self.planes_fmt:
            return self
        if self._twin is None:
            t = Context(self.device, self.stream)
            set_planes_format(t, fmt)
            t._twin = self
            ws = getattr(self, "workspace", None)
            t.workspace = ws
            if ws is not None:
                check(lib.pp_ctx_set_workspace(t.handle, _ptr(ws), ws.numel() * 4), t.handle, "pp_ctx_set_workspace")
            self._twin = t
        assert self._twin.planes_fmt == int(fmt)
        return self._twin

    def set_workspace(self, nbytes):
        """Scratch for split-K convolutions (pp_ctx_set_workspace); 0 removes it."""
        self.workspace = torch.empty((int(nbytes) // 4,), dtype=torch.float32, device="cuda:%d" % self.device) if nbytes else None
        for c in (self, self._twin):
            if c is not None:
                c.workspace = self.workspace
                check(lib.pp_ctx_set_workspace(c.handle, _ptr(self.workspace), (int(nbytes) // 4) * 4 if nbytes else 0), c.handle,
                      "pp_ctx_set_workspace")

    def device_info(self):
        n = C.c_int(0)
        buf = C.create_string_buffer(128)
        check(lib.pp_device_info(self.handle, C.byref(n), buf, 128), self.handle)
        return n.value, buf.value.decode()

    def close(self):
        t, self._twin = self._twin, None
        if t is not None and t.handle:
            t._twin = None
            t.close()
        if self.handle:
            lib.pp_ctx_destroy(self.handle)
            self.handle = C.c_void_p()


def make_conv_desc(n_img, in_shapes, out_shapes, cin, cout, k, stride, pad_t, pad_l, ld_x, ld_y, ld_w):
    d = ConvDesc()
    d.in_ = RowSpace.make(n_img, in_shapes)
    d.out = RowSpace.make(n_img, out_shapes)
    d.cin, d.cout, d.kh, d.kw, d.stride = cin, cout, k, k, stride
    d.pad_t, d.pad_l, d.ld_x, d.ld_y, d.ld_w = pad_t, pad_l, ld_x, ld_y, ld_w
    return d


def conv_fwd(ctx, d, x, w, bias, residual, relu, y):
    ld_res = residual.stride(0) if residual is not None else 0
    check(lib.pp_conv2d_nhwc_fwd(ctx.handle, C.byref(d), _ptr(x), _ptr(w), _ptr(bias), _ptr(residual), ld_res,
                                 int(bool(relu)), _ptr(y)), ctx.handle, "pp_conv2d_nhwc_fwd")


def conv_bwd_data(ctx, d, dy, w, addend, relu_src, dx):
    ld_add = addend.stride(0) if addend is not None else 0
    ld_rs = relu_src.stride(0) if relu_src is not None else 0
    check(lib.pp_conv2d_nhwc_bwd_data(ctx.handle, C.byref(d), _ptr(dy), _ptr(w), _ptr(addend), ld_add, _ptr(relu_src),
                                      ld_rs, _ptr(dx)), ctx.handle, "pp_conv2d_nhwc_bwd_data")


def conv_bwd_weight(ctx, d, x, dy, dw, dbias):
    check(lib.pp_conv2d_nhwc_bwd_weight(ctx.handle, C.byref(d), _ptr(x), _ptr(dy), _ptr(dw), _ptr(dbias)), ctx.handle,
          "pp_conv2d_nhwc_bwd_weight")


def conv_split_weights3(ctx, d, w, fwd_hi, fwd_lo, dg_hi, dg_lo):
    check(lib.pp_conv_split_weights_bf16x3(ctx.handle, C.byref(d), _ptr(w), _ptr(fwd_hi), _ptr(fwd_lo), _ptr(dg_hi), _ptr(dg_lo)),
          ctx.handle, "pp_conv_split_weights_bf16x3")


class SplitWeightsBatch(object):
    """Job table (device resident) for pp_conv_split_weights_bf16x3_batch: every listed tensor is re-split by one launch."""

    def __init__(self, jobs):
        """jobs: iterable of (desc, w, fwd_hi, fwd_lo, dg_hi, dg_lo) as for conv_split_weights3."""
        from ._lib import SplitJob
        jobs = list(jobs)
        arr = (SplitJob * max(len(jobs), 1))()
        tiles = 0
        self._keep = jobs  # the table holds raw pointers into these tensors
        for i, (d, w, fh, fl, dh, dl) in enumerate(jobs):
            if d.cin % 32:
                raise ValueError("SplitWeightsBatch: cin %d must be a multiple of 32" % d.cin)
            j = arr[i]
            j.w, j.fwd_hi, j.fwd_lo, j.dg_hi, j.dg_lo = [(t.data_ptr() if t is not None else None) for t in (w, fh, fl, dh, dl)]
            j.taps, j.cin, j.cout, j.ld_w, j.tile_begin = d.kh * d.kw, d.cin, d.cout, d.ld_w, tiles
            tiles += j.taps * (d.cin // 32) * ((d.cout + 31) // 32)
        self.n, self.tiles = len(jobs), tiles
        host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        self.table = host.cuda()

    def run(self, ctx):
        check(lib.pp_conv_split_weights_bf16x3_batch(ctx.handle, self.n, _ptr(self.table), self.tiles), ctx.handle,
              "pp_conv_split_weights_bf16x3_batch")


def new_planes(rows, ld, device="cuda", fill=0):
    """A tensor [rows][ld] (ld % 8 == 0) as PACKED bf16 planes: one buffer of rows * ld * 4 bytes cut into 32-byte groups of 8
    channels -- 16 bytes of hi, 16 bytes of lo (value = hi + lo).  Returns the (hi, lo) pair the C ABI takes: int16 views
    [rows, ld / 8, 8] of that buffer with lo.data_ptr() == hi.data_ptr() + 16; row slices of both stay valid pairs."""
    assert ld % 8 == 0, ld
    base = torch.full((rows, ld // 8, 2, 8), fill, dtype=torch.int16, device=device)
    return base[:, :, 0, :], base[:, :, 1, :]


def planes_ld(planes):
    """leading dimension (elements per row) of a packed plane pair"""
    return planes[0].stride(0) // 2


def set_planes_format(ctx, fmt):
    """0: bf16 pairs (bf16x3 arithmetic); 1: P16 (IEEE half + two e5m2 bytes; f16c8 arithmetic).  See pp_ctx_set_planes_format."""
    check(lib.pp_ctx_set_planes_format(ctx.handle, int(fmt)), ctx.handle, "pp_ctx_set_planes_format")
    ctx.planes_fmt = int(fmt)


def convert_planes(ctx, src, src_fmt, dst, dst_fmt, scale2=None, scale_index=0, relu_src_hi=None):
    """re-encode a plane pair into the other format (dst = src * scale2[scale_index] when scale2 is given; zero where the tensor
    with hi plane relu_src_hi is not positive)"""
    n = src[0].numel()
    check(lib.pp_convert_planes(ctx.handle, n, _ptr(src[0]), _ptr(src[1]), int(src_fmt), _ptr(dst[0]), _ptr(dst[1]), int(dst_fmt), _ptr(scale2),
                                int(scale_index), _ptr(relu_src_hi)), ctx.handle, "pp_convert_planes")


def planes_to_f32(planes, fmt=0):
    """the values of a plane pair as a float32 tensor [rows, ld] (torch arithmetic: tests / inspection).  fmt 0: bf16 pairs, hi + lo;
    fmt 1 (P16): hi = IEEE halves, the lo unit of a gathered-operand tensor is [e5m2(x) | e5m2(remainder * 2^12) << 8]."""
    hi, lo = planes
    rows = hi.shape[0]
    if fmt == 0:
        return (hi.contiguous().view(torch.bfloat16).float() + lo.contiguous().view(torch.bfloat16).float()).reshape(rows, -1)
    rem = (lo.contiguous().to(torch.int32) & 0xff00).to(torch.int16).view(torch.float16).float() / 4096.0
    return (hi.contiguous().view(torch.float16).float() + rem).reshape(rows, -1)


def weight_planes_to_f32(hi, lo, fmt=0):
    """the same for weight planes (P16: the lo unit's bytes are swapped, the remainder is its LOW byte)"""
    if fmt == 0:
        return hi.contiguous().view(torch.bfloat16).float() + lo.contiguous().view(torch.bfloat16).float()
    rem = ((lo.contiguous().to(torch.int32) & 0xff) << 8).to(torch.int16).view(torch.float16).float() / 4096.0
    return hi.contiguous().view(torch.float16).float() + rem


def split_planes3(ctx, src, hi, lo, scale=None):
    """scale: device float32 [2] = {2^G, 2^-G} (grad_scale_from_counts): the planes hold src * 2^G"""
    if scale is None:
        check(lib.pp_split_planes_bf16x3(ctx.handle, src.numel(), _ptr(src), _ptr(hi), _ptr(lo)), ctx.handle, "pp_split_planes_bf16x3")
    else:
        check(lib.pp_split_planes_scaled_bf16x3(ctx.handle, src.numel(), _ptr(src), _ptr(hi), _ptr(lo), _ptr(scale)), ctx.handle,
              "pp_split_planes_scaled_bf16x3")


def grad_scale_from_counts(ctx, counts, scale2, log2_adjust=0):
    check(lib.pp_grad_scale_from_counts_adj(ctx.handle, _ptr(counts), int(counts.numel()), _ptr(scale2), int(log2_adjust)), ctx.handle,
          "pp_grad_scale_from_counts")


def set_grad_scale(ctx, scale2):
    """persistent: the weight gradients of this context divide the gradient scale out (None: gradients are unscaled)"""
    ctx._grad_scale_keep = scale2
    check(lib.pp_ctx_set_grad_scale(ctx.handle, _ptr(scale2)), ctx.handle, "pp_ctx_set_grad_scale")


def _set_capture(ctx, planes):
    if planes is not None:
        check(lib.pp_ctx_set_split_capture(ctx.handle, _ptr(planes[0]), _ptr(planes[1])), ctx.handle, "pp_ctx_set_split_capture")


def row_block_list(ctx, x, cols, flags=None, blocks=None):
    """pp_row_block_list: x float32 [rows, ld] -> (flags uint8 [2 nb], list int32 [2 (1 + nb)]) with nb = ceil(rows / 32);
    the first halves hold the result, the second halves are scratch of the bwd-data launch that takes the hint."""
    rows, ld = x.shape
    nb = (rows + 31) // 32
    if flags is None:
        flags = torch.zeros((2 * nb,), dtype=torch.uint8, device=x.device)
    if blocks is None:
        blocks = torch.zeros((2 * (nb + 1),), dtype=torch.int32, device=x.device)
    check(lib.pp_row_block_list(ctx.handle, _ptr(x), rows, x.stride(0), int(cols), _ptr(flags), _ptr(blocks)), ctx.handle, "pp_row_block_list")
    return flags, blocks


def row_block_list_planes(ctx, planes, cols, flags, blocks, within=None):
    """pp_row_block_list_planes(_within): the same scan of a tensor stored as bf16 (hi, lo) planes [rows, ld]; within = uint8 flags of
    the only blocks that can hold a non-zero (the others are not read)"""
    hi, lo = planes
    rows = hi.shape[0]
    if within is not None:
        assert within.dtype == torch.uint8 and within.is_contiguous() and within.numel() >= (rows + 31) // 32
        check(lib.pp_row_block_list_planes_within(ctx.handle, _ptr(hi), _ptr(lo), rows, planes_ld(planes), int(cols), _ptr(within), _ptr(flags),
                                                  _ptr(blocks)), ctx.handle, "pp_row_block_list_planes_within")
    else:
        check(lib.pp_row_block_list_planes(ctx.handle, _ptr(hi), _ptr(lo), rows, planes_ld(planes), int(cols), _ptr(flags), _ptr(blocks)),
              ctx.handle, "pp_row_block_list_planes")
    return flags, blocks


def _set_epilogue_planes(ctx, add_planes, mask_hi):
    if add_planes is not None or mask_hi is not None:
        ah, al = add_planes if add_planes is not None else (None, None)
        check(lib.pp_ctx_set_epilogue_planes(ctx.handle, _ptr(ah), _ptr(al), _ptr(mask_hi)), ctx.handle, "pp_ctx_set_epilogue_planes")


def tview(t=None, planes=None):
    """pp_tview of a float32 tensor and / or a (hi, lo) plane pair; None, None = the NULL view"""
    v = TView()
    v.f32 = t.data_ptr() if t is not None else None
    v.hi = planes[0].data_ptr() if planes is not None else None
    v.lo = planes[1].data_ptr() if planes is not None else None
    v._keep = (t, planes)
    return v


def _numel(v):
    t, planes = v._keep
    return (t if t is not None else planes[0]).numel()


def add_n_v(ctx, a, b, c, out):
    null = TView()
    check(lib.pp_add_n_v(ctx.handle, _numel(a), C.byref(a), C.byref(b if b is not None else null), C.byref(c if c is not None else null),
                         C.byref(out)), ctx.handle, "pp_add_n_v")


def relu_fwd_v(ctx, x, y):
    check(lib.pp_relu_fwd_v(ctx.handle, _numel(x), C.byref(x), C.byref(y)), ctx.handle, "pp_relu_fwd_v")


def upsample_add_fwd_v(ctx, n_img, sh, sw, th, tw, c, src, other, out):
    null = TView()
    check(lib.pp_upsample_nearest_add_fwd_v(ctx.handle, n_img, sh, sw, th, tw, c, C.byref(src), C.byref(other if other is not None else null),
                                            C.byref(out)), ctx.handle, "pp_upsample_nearest_add_fwd_v")


def upsample_add_bwd_v(ctx, n_img, sh, sw, th, tw, c, dtarget, base, dsrc):
    null = TView()
    check(lib.pp_upsample_nearest_add_bwd_v(ctx.handle, n_img, sh, sw, th, tw, c, C.byref(dtarget), C.byref(base if base is not None else null),
                                            C.byref(dsrc)), ctx.handle, "pp_upsample_nearest_add_bwd_v")


def merge_planes3(ctx, planes, dst):
    check(lib.pp_merge_planes_bf16x3(ctx.handle, dst.numel(), _ptr(planes[0]), _ptr(planes[1]), _ptr(dst)), ctx.handle, "pp_merge_planes_bf16x3")


def planes_stats(ctx, planes, cols, stats, within=None):
    """pp_planes_stats: adds (elements, non-zero, at the clamp, subnormal) of a P16 tensor's halves to stats[0..3] and keeps the largest
    |half| (15 bits) in stats[4] (int64 [5], device);
    ctx must be the P16 twin; within = uint8 flags of the 32-row blocks to look at"""
    hi, lo = planes
    check(lib.pp_planes_stats(ctx.handle, _ptr(hi), _ptr(lo), int(hi.shape[0]), planes_ld(planes), int(cols), _ptr(within), _ptr(stats)), ctx.handle,
          "pp_planes_stats")
    return stats


def positive_row_blocks(ctx, rs, n_anchor, y_true, flags):
    """pp_positive_row_blocks: flags[b] = 1 where the 32-row block b of the row space rs holds an anchor with state 1
    (y_true [B, N, stride], state = last column)"""
    check(lib.pp_positive_row_blocks(ctx.handle, C.byref(rs), int(n_anchor), int(y_true.shape[-1]), _ptr(y_true), _ptr(flags)), ctx.handle,
          "pp_positive_row_blocks")
    return flags


def row_block_dilate(ctx, d, in_flags, out_flags):
    """pp_row_block_dilate: the blocks within one pixel (2-D) of a flagged block, on the grid of the 3x3 stride-1 conv d"""
    check(lib.pp_row_block_dilate(ctx.handle, C.byref(d), _ptr(in_flags), _ptr(out_flags)), ctx.handle, "pp_row_block_dilate")
    return out_flags


def set_row_block_out(ctx, flags, blocks):
    """pp_ctx_set_row_block_out (one-shot): the next conv_fwd3 on ctx computes the flagged 32-row output blocks only"""
    check(lib.pp_ctx_set_row_block_out(ctx.handle, _ptr(flags), _ptr(blocks)), ctx.handle, "pp_ctx_set_row_block_out")


def _set_skip(ctx, skip, lazy_out=False, lazy_in=False):
    if skip is not None:
        check(lib.pp_ctx_set_row_block_skip(ctx.handle, _ptr(skip[0]), _ptr(skip[1])), ctx.handle, "pp_ctx_set_row_block_skip")
    if lazy_out or lazy_in:
        if skip is None:
            raise ValueError("lazy sparse gradients go with a row-block skip hint (dy_skip)")
        check(lib.pp_ctx_set_row_block_lazy(ctx.handle, int(bool(lazy_out)), int(bool(lazy_in))), ctx.handle, "pp_ctx_set_row_block_lazy")


def conv_fwd3(ctx, d, x, w_hi, w_lo, bias, residual, relu, y, x_planes=None, y_planes=None, x_capture=None, res_planes=None):
    """x_capture = (hi, lo): the launch also writes the bf16 split of x (pp_ctx_set_split_capture).
    res_planes = (hi, lo): the residual as planes (pp_ctx_set_epilogue_planes; `residual` must then be None)."""
    _set_capture(ctx, x_capture)
    _set_epilogue_planes(ctx, res_planes, None)
    ld_res = residual.stride(0) if residual is not None else (planes_ld(res_planes) if res_planes is not None else 0)
    xh, xl = x_planes if x_planes is not None else (None, None)
    yh, yl = y_planes if y_planes is not None else (None, None)
    check(lib.pp_conv2d_nhwc_fwd_bf16x3(ctx.handle, C.byref(d), _ptr(x), _ptr(xh), _ptr(xl), _ptr(w_hi), _ptr(w_lo), _ptr(bias),
                                        _ptr(residual), ld_res, int(bool(relu)), _ptr(y), _ptr(yh), _ptr(yl)), ctx.handle,
          "pp_conv2d_nhwc_fwd_bf16x3")


def conv_bwd_data3(ctx, d, dy, w_hi, w_lo, addend, relu_src, dx, dy_planes=None, dx_planes=None, dy_capture=None, dy_skip=None,
                   addend_planes=None, relu_src_hi=None, lazy_out=False, lazy_in=False):
    """dy_capture = (hi, lo): the launch also writes the bf16 split of dy (pp_ctx_set_split_capture).
    dy_skip = (flags, list) from row_block_list(dy): tiles that only see zero blocks of dy skip their reduction.
    addend_planes = (hi, lo) / relu_src_hi = hi plane: those epilogue operands as planes (pp_ctx_set_epilogue_planes).
    lazy_out / lazy_in: pp_ctx_set_row_block_lazy (dx is left unwritten outside the computed blocks / dy is such a tensor)."""
    _set_capture(ctx, dy_capture)
    _set_skip(ctx, dy_skip, lazy_out, lazy_in)
    _set_epilogue_planes(ctx, addend_planes, relu_src_hi)
    ld_add = addend.stride(0) if addend is not None else (planes_ld(addend_planes) if addend_planes is not None else 0)
    ld_rs = relu_src.stride(0) if relu_src is not None else (relu_src_hi.stride(0) // 2 if relu_src_hi is not None else 0)
    dh, dl = dy_planes if dy_planes is not None else (None, None)
    xh, xl = dx_planes if dx_planes is not None else (None, None)
    check(lib.pp_conv2d_nhwc_bwd_data_bf16x3(ctx.handle, C.byref(d), _ptr(dy), _ptr(dh), _ptr(dl), _ptr(w_hi), _ptr(w_lo), _ptr(addend),
                                             ld_add, _ptr(relu_src), ld_rs, _ptr(dx), _ptr(xh), _ptr(xl)), ctx.handle,
          "pp_conv2d_nhwc_bwd_data_bf16x3")


def conv_bwd_weight3(ctx, d, x, dy, dw, dbias, x_planes=None, dy_planes=None, dy_skip=None, lazy_in=False):
    """dy_skip = (flags, list) from row_block_list(dy): the reduction walks the listed 32-row blocks only.
    lazy_in: dy is unwritten outside those blocks (pp_ctx_set_row_block_lazy): fails unless the listed-block reduction runs."""
    _set_skip(ctx, dy_skip, False, lazy_in)
    xh, xl = x_planes if x_planes is not None else (None, None)
    dh, dl = dy_planes if dy_planes is not None else (None, None)
    check(lib.pp_conv2d_nhwc_bwd_weight_bf16x3(ctx.handle, C.byref(d), _ptr(x), _ptr(dy), _ptr(xh), _ptr(xl), _ptr(dh), _ptr(dl),
                                               _ptr(dw), _ptr(dbias)), ctx.handle, "pp_conv2d_nhwc_bwd_weight_bf16x3")


def maxpool3x3s2(ctx, n_img, h, w, c, x, oh, ow, y):
    check(lib.pp_maxpool3x3s2_fwd(ctx.handle, n_img, h, w, c, _ptr(x), oh, ow, _ptr(y)), ctx.handle, "pp_maxpool3x3s2_fwd")


def upsample_add_fwd(ctx, n_img, sh, sw, th, tw, c, src, other, out):
    check(lib.pp_upsample_nearest_add_fwd(ctx.handle, n_img, sh, sw, th, tw, c, _ptr(src), _ptr(other), _ptr(out)),
          ctx.handle, "pp_upsample_nearest_add_fwd")


def upsample_add_bwd(ctx, n_img, sh, sw, th, tw, c, dtarget, base, dsrc):
    check(lib.pp_upsample_nearest_add_bwd(ctx.handle, n_img, sh, sw, th, tw, c, _ptr(dtarget), _ptr(base), _ptr(dsrc)),
          ctx.handle, "pp_upsample_nearest_add_bwd")


def add_n(ctx, a, b, c, out):
    check(lib.pp_add_n(ctx.handle, a.numel(), _ptr(a), _ptr(b), _ptr(c), _ptr(out)), ctx.handle, "pp_add_n")


def relu_fwd(ctx, x, y):
    check(lib.pp_relu_fwd(ctx.handle, x.numel(), _ptr(x), _ptr(y)), ctx.handle, "pp_relu_fwd")


def warp_affine_u8(ctx, images_u8, matrices, interpolation="linear", border="replicate", cval=0, out=None):
    """cv2.warpAffine per image of a uint8 batch [B,H,W,3] or [B,H,W] (utils/image.py:207-214, :222-229): matrices = B forward
    2x3 (or 3x3) matrices on the host; interpolation 'linear' / 'nearest'; border 'replicate' ('nearest' fill mode) / 'constant'."""
    B, H, W = images_u8.shape[:3]
    ch = images_u8.shape[3] if images_u8.dim() == 4 else 1
    m = np.ascontiguousarray(np.asarray(matrices, np.float64).reshape(B, -1)[:, :6])
    if out is None:
        out = torch.empty_like(images_u8)
    check(lib.pp_warp_affine_u8(ctx.handle, B, H, W, ch, m.ctypes.data_as(C.POINTER(C.c_double)), {"nearest": 0, "linear": 1}[interpolation],
                                {"constant": 0, "replicate": 1}[border], int(cval), _ptr(images_u8), _ptr(out)), ctx.handle, "pp_warp_affine_u8")
    return out


def resize_scale(rows, cols, min_side=480, max_side=640):
    s = C.c_double(0)
    check(lib.pp_resize_scale(int(rows), int(cols), int(min_side), int(max_side), C.byref(s)), None, "pp_resize_scale")
    return s.value


def resize_linear_u8(ctx, images_u8, scale):
    """cv2.resize(img, None, fx=scale, fy=scale) of a uint8 batch [B,H,W,3] or [B,H,W] (utils/image.py:307-323)."""
    B, H, W = images_u8.shape[:3]
    ch = images_u8.shape[3] if images_u8.dim() == 4 else 1
    dh, dw = int(np.rint(H * scale)), int(np.rint(W * scale))
    out = torch.empty((B, dh, dw) + ((ch,) if images_u8.dim() == 4 else ()), dtype=torch.uint8, device=images_u8.device)
    check(lib.pp_resize_linear_u8(ctx.handle, B, H, W, ch, float(scale), dh, dw, _ptr(images_u8), _ptr(out)), ctx.handle, "pp_resize_linear_u8")
    return out


def preprocess_caffe_u8(ctx, images_u8, sizes_hw, x4):
    """images_u8: cuda uint8 [B,H,W,3]; sizes_hw: B (h, w) pairs; x4: cuda float32 [B*H*W, 4] (the engine's stem input)."""
    Bn, H, W, _ = images_u8.shape
    assert images_u8.dtype == torch.uint8 and images_u8.is_contiguous()
    arr = (C.c_int * (2 * Bn))(*[int(v) for hw in sizes_hw for v in hw])
    check(lib.pp_preprocess_caffe_u8(ctx.handle, Bn, H, W, arr, _ptr(images_u8), _ptr(x4)), ctx.handle, "pp_preprocess_caffe_u8")


def pack_rgb_to_4_padded(ctx, x3, x4p, Hp, Wp, pad=3):
    Bn, H, W, _ = x3.shape
    check(lib.pp_pack_rgb_to_4_padded(ctx.handle, Bn, H, W, Hp, Wp, pad, _ptr(x3), _ptr(x4p)), ctx.handle, "pp_pack_rgb_to_4_padded")


def preprocess_caffe_u8_padded(ctx, images_u8, sizes_hw, x4p, Hp, Wp, pad=3):
    Bn, H, W, _ = images_u8.shape
    assert images_u8.dtype == torch.uint8 and images_u8.is_contiguous()
    arr = (C.c_int * (2 * Bn))(*[int(v) for hw in sizes_hw for v in hw])
    check(lib.pp_preprocess_caffe_u8_padded(ctx.handle, Bn, H, W, Hp, Wp, pad, arr, _ptr(images_u8), _ptr(x4p)), ctx.handle,
          "pp_preprocess_caffe_u8_padded")


def stem7x7s2_fwd3(ctx, n_img, H, W, Hp, Wp, x4p, w_hi, w_lo, cout, bias, relu, y):
    check(lib.pp_stem7x7s2_fwd_bf16x3(ctx.handle, n_img, H, W, Hp, Wp, _ptr(x4p), _ptr(w_hi), _ptr(w_lo), cout, _ptr(bias), int(bool(relu)),
                                      _ptr(y), y.stride(0)), ctx.handle, "pp_stem7x7s2_fwd_bf16x3")


def pack_rgb_to_4(ctx, x3, x4):
    check(lib.pp_pack_rgb_to_4(ctx.handle, x3.numel() // 3, _ptr(x3), _ptr(x4)), ctx.handle, "pp_pack_rgb_to_4")


def export_head(ctx, rs, n_anchor, n_val, src, apply_sigmoid, out):
    check(lib.pp_export_head(ctx.handle, C.byref(rs), n_anchor, n_val, _ptr(src), src.stride(0), int(apply_sigmoid), _ptr(out)),
          ctx.handle, "pp_export_head")


def count_positives(ctx, y_box, y_cls, y_mask, counts):
    rb = y_box.shape[0] * y_box.shape[1] if y_box is not None else 0
    rc = y_cls.shape[0] * y_cls.shape[1] if y_cls is not None else 0
    rm = y_mask.shape[0] * y_mask.shape[1] if y_mask is not None else 0
    cc = y_cls.shape[2] - 1 if y_cls is not None else 0
    cm = y_mask.shape[2] - 1 if y_mask is not None else 0
    check(lib.pp_count_positives(ctx.handle, rb, _ptr(y_box), rc, cc, _ptr(y_cls), rm, cm, _ptr(y_mask), _ptr(counts)),
          ctx.handle, "pp_count_positives")


def focal(ctx, rs, n_anchor, n_class, logits, y_true, alpha, gamma, count, loss_weight, loss_sum, dlogits):
    check(lib.pp_sigmoid_focal_fwd_bwd(ctx.handle, C.byref(rs), n_anchor, n_class, _ptr(logits), logits.stride(0),
                                       _ptr(y_true), alpha, gamma, _ptr(count), loss_weight, _ptr(loss_sum), _ptr(dlogits)),
          ctx.handle, "pp_sigmoid_focal_fwd_bwd")


def orth_l1(ctx, rs, n_anchor, pred, y_true, weight, sigma, count, loss_weight, loss_sum, dpred):
    check(lib.pp_orth_smoothl1_fwd_bwd(ctx.handle, C.byref(rs), n_anchor, _ptr(pred), pred.stride(0), _ptr(y_true), weight,
                                       sigma, _ptr(count), loss_weight, _ptr(loss_sum), _ptr(dpred)),
          ctx.handle, "pp_orth_smoothl1_fwd_bwd")


class Optimizer:
    def __init__(self, ctx, descs, total):
        arr = (ParamDesc * len(descs))(*descs)
        self.handle = C.c_void_p()
        self.ctx = ctx
        check(lib.pp_optimizer_create(ctx.handle, C.byref(self.handle), arr, len(descs), total), ctx.handle,
              "pp_optimizer_create")

    def grad_norm(self, w_master, g_eff, scales, gnorm_sq, l2_loss=None):
        check(lib.pp_grad_global_norm(self.ctx.handle, self.handle, _ptr(w_master), _ptr(g_eff), _ptr(scales),
                                      _ptr(gnorm_sq), _ptr(l2_loss)), self.ctx.handle, "pp_grad_global_norm")

    def adam_step(self, w_master, w_eff, g_eff, scales, m, v, gnorm_sq, lr, beta1, beta2, eps, clipnorm, step):
        check(lib.pp_adam_step_clipnorm(self.ctx.handle, self.handle, _ptr(w_master), _ptr(w_eff), _ptr(g_eff), _ptr(scales),
                                        _ptr(m), _ptr(v), _ptr(gnorm_sq), lr, beta1, beta2, eps, clipnorm, int(step)),
              self.ctx.handle, "pp_adam_step_clipnorm")

    def close(self):
        if self.handle:
            lib.pp_optimizer_destroy(self.handle)
            self.handle = C.c_void_p()


# ---- anchors / targets / decode -------------------------------------------------------------------
def _iarr(v):
    return (C.c_int * len(v))(*[int(x) for x in v])


def generate_base_anchors(base_size, ratios, scales):
    r = np.ascontiguousarray(ratios, np.float32)
    s = np.ascontiguousarray(scales, np.float32)
    out = np.empty((len(r) * len(s), 4), np.float64)
    check(lib.pp_generate_base_anchors_host(int(base_size), r.ctypes.data_as(C.POINTER(C.c_float)), len(r),
                                            s.ctypes.data_as(C.POINTER(C.c_float)), len(s),
                                            out.ctypes.data_as(C.POINTER(C.c_double))), None, "pp_generate_base_anchors_host")
    return out


def anchors_shift(ctx, feat_shapes, strides, base_anchors, dtype=torch.float64):
    """feat_shapes: [(h, w)] per level; base_anchors: float64 [L, A, 4] (numpy)."""
    base = np.ascontiguousarray(base_anchors, np.float64)
    L, A = base.shape[0], base.shape[1]
    n = sum(h * w for h, w in feat_shapes) * A
    out = torch.empty((n, 4), dtype=dtype, device="cuda")
    fn = lib.pp_anchors_shift_f64 if dtype == torch.float64 else lib.pp_anchors_shift_f32
    check(fn(ctx.handle, L, _iarr([h for h, _ in feat_shapes]), _iarr([w for _, w in feat_shapes]), _iarr(strides), A,
             base.ctypes.data_as(C.POINTER(C.c_double)), _ptr(out)), ctx.handle, "pp_anchors_shift")
    return out


def compute_overlap(ctx, boxes, query):
    n, k = boxes.shape[0], query.shape[0]
    out = torch.zeros((n, k), dtype=torch.float64, device="cuda")
    check(lib.pp_compute_overlap_f64(ctx.handle, n, _ptr(boxes), k, _ptr(query), _ptr(out)), ctx.handle, "pp_compute_overlap_f64")
    return out


def compute_gt_annotations(ctx, anchors, gt, neg=0.4, pos=0.5):
    n, k = anchors.shape[0], gt.shape[0]
    argmax = torch.empty((n,), dtype=torch.int32, device="cuda")
    state = torch.empty((n,), dtype=torch.int8, device="cuda")
    check(lib.pp_compute_gt_annotations(ctx.handle, n, _ptr(anchors), k, _ptr(gt), neg, pos, _ptr(argmax), _ptr(state)),
          ctx.handle, "pp_compute_gt_annotations")
    return argmax, state


def project_box3d(pose7, box8x3, cam4):
    p = np.ascontiguousarray(pose7, np.float64)
    b = np.ascontiguousarray(box8x3, np.float64)
    c = np.ascontiguousarray(cam4, np.float64)
    out = np.empty(16, np.float64)
    dp = C.POINTER(C.c_double)
    check(lib.pp_project_box3d_host(p.ctypes.data_as(dp), b.ctypes.data_as(dp), c.ctypes.data_as(dp), out.ctypes.data_as(dp)),
          None, "pp_project_box3d_host")
    return out


def pil_nearest_index(n_in, n_out):
    out = np.empty(n_out, np.int32)
    check(lib.pp_pil_nearest_index_host(n_in, n_out, out.ctypes.data_as(C.POINTER(C.c_int))), None, "pp_pil_nearest_index_host")
    return out


def anchor_targets(ctx, anchors, gt_offset, gt_boxes, gt_labels, gt_box3d, gt_mask_ids, id_masks, mask_hw, image_hw,
                   num_classes, out_mh, out_mw, neg=0.4, pos=0.5):
    """anchors: cuda f64 [N,4]; gt_* cuda tensors packed over the batch; id_masks cuda uint8 [B,H,W] or None."""
    B = len(gt_offset) - 1
    N = anchors.shape[0]
    reg = torch.empty((B, N, 17), dtype=torch.float32, device="cuda")
    lab = torch.empty((B, N, num_classes + 1), dtype=torch.float32, device="cuda")
    msk = torch.empty((B, out_mh * out_mw, num_classes + 1), dtype=torch.float32, device="cuda")
    mh, mw = (id_masks.shape[1], id_masks.shape[2]) if id_masks is not None else (0, 0)
    check(lib.pp_anchor_targets(ctx.handle, N, _ptr(anchors), B, _iarr(gt_offset), _ptr(gt_boxes), _ptr(gt_labels),
                                _ptr(gt_box3d), _ptr(gt_mask_ids), _ptr(id_masks), mh, mw,
                                _iarr(np.asarray(mask_hw).reshape(-1)) if mask_hw is not None else None,
                                _iarr(np.asarray(image_hw).reshape(-1)), num_classes, neg, pos, out_mh, out_mw,
                                _ptr(reg), _ptr(lab), _ptr(msk)), ctx.handle, "pp_anchor_targets")
    return reg, lab, msk


def box3d_decode(ctx, anchors_f32, regression):
    B, N = regression.shape[0], regression.shape[1]
    out = torch.empty_like(regression)
    check(lib.pp_box3d_decode(ctx.handle, B, N, _ptr(anchors_f32), _ptr(regression), _ptr(out)), ctx.handle, "pp_box3d_decode")
    return out


def score_threshold_compact(ctx, scores, thr=0.5, cap=None):
    B, N, Cc = scores.shape
    cap = int(cap or N)
    idx = torch.empty((B, Cc, cap), dtype=torch.int32, device="cuda")
    cnt = torch.empty((B, Cc), dtype=torch.int32, device="cuda")
    check(lib.pp_score_threshold_compact(ctx.handle, B, N, Cc, _ptr(scores), thr, cap, _ptr(idx), _ptr(cnt)), ctx.handle,
          "pp_score_threshold_compact")
    return idx, cnt


def filter_detections_batch(ctx, boxes, boxes3d, scores, score_thr=0.05, iou_thr=0.5, max_det=300):
    """boxes [B,N,4], boxes3d [B,N,16], scores [B,N,C] -> ([B,max_det,4], [B,max_det,16], [B,max_det], [B,max_det] int32)."""
    Bn, N, Cc = scores.shape
    ws = torch.empty((Bn * lib.pp_filter_workspace_bytes(N, Cc, max_det),), dtype=torch.uint8, device="cuda")
    ob = torch.empty((Bn, max_det, 4), dtype=torch.float32, device="cuda")
    o3 = torch.empty((Bn, max_det, 16), dtype=torch.float32, device="cuda")
    osc = torch.empty((Bn, max_det), dtype=torch.float32, device="cuda")
    ol = torch.empty((Bn, max_det), dtype=torch.int32, device="cuda")
    check(lib.pp_filter_detections_batch(ctx.handle, Bn, N, Cc, _ptr(boxes), _ptr(boxes3d), _ptr(scores), score_thr, iou_thr, max_det,
                                         _ptr(ws), _ptr(ob), _ptr(o3), _ptr(osc), _ptr(ol)), ctx.handle, "pp_filter_detections_batch")
    return ob, o3, osc, ol


def filter_detections(ctx, boxes, boxes3d, scores, score_thr=0.05, iou_thr=0.5, max_det=300):
    N, Cc = scores.shape
    ws = torch.empty((lib.pp_filter_workspace_bytes(N, Cc, max_det),), dtype=torch.uint8, device="cuda")
    ob = torch.empty((max_det, 4), dtype=torch.float32, device="cuda")
    o3 = torch.empty((max_det, 16), dtype=torch.float32, device="cuda")
    osc = torch.empty((max_det,), dtype=torch.float32, device="cuda")
    ol = torch.empty((max_det,), dtype=torch.int32, device="cuda")
    check(lib.pp_filter_detections(ctx.handle, N, Cc, _ptr(boxes), _ptr(boxes3d), _ptr(scores), score_thr, iou_thr, max_det,
                                   _ptr(ws), _ptr(ob), _ptr(o3), _ptr(osc), _ptr(ol)), ctx.handle, "pp_filter_detections")
    return ob, o3, osc, ol


def pose_errors(ctx, pts, R_est, t_est, R_gt, t_gt, symmetric=False):
    """ADD (symmetric=False) or ADD-S / ADI (True) of n poses against one model: cuda float64 tensors
    pts [n_pts,3], R_* [n,3,3], t_* [n,3] -> float64 [n]."""
    n, n_pts = R_est.shape[0], pts.shape[0]
    args = [t.contiguous() for t in (pts, R_est, t_est, R_gt, t_gt)]
    for t in args:
        assert t.dtype == torch.float64 and t.is_cuda
    ws = torch.empty((max(1, lib.pp_pose_error_workspace_bytes(n, n_pts)),), dtype=torch.uint8, device="cuda")
    out = torch.empty((n,), dtype=torch.float64, device="cuda")
    fn = lib.pp_pose_adi_f64 if symmetric else lib.pp_pose_add_f64
    check(fn(ctx.handle, n, n_pts, *[_ptr(t) for t in args], _ptr(ws), _ptr(out)), ctx.handle, "pp_pose_adi_f64" if symmetric else "pp_pose_add_f64")
    return out


def pnp_ransac(ctx, offsets, obj, img, K4, iterations=300, reproj_error=5.0, seed=0, points_per_vote=8):
    """Batched RANSAC-PnP (pp_pnp_ransac_f64): cuda tensors offsets int32 [P+1], obj float64 [N,3], img float64 [N,2],
    K4 float64 [P,4] -> (R [P,3,3], t [P,3], n_inliers int32 [P], inlier mask uint8 [N], ok int32 [P])."""
    P = int(offsets.numel()) - 1
    N = int(obj.shape[0])
    assert offsets.dtype == torch.int32 and obj.dtype == torch.float64 and img.dtype == torch.float64 and K4.dtype == torch.float64
    assert obj.shape == (N, 3) and img.shape == (N, 2) and K4.shape == (P, 4)
    obj, img, K4, offsets = obj.contiguous(), img.contiguous(), K4.contiguous(), offsets.contiguous()
    dev = obj.device
    ws = torch.empty((max(1, lib.pp_pnp_ransac_workspace_bytes(P, int(iterations))),), dtype=torch.uint8, device=dev)
    R = torch.empty((P, 3, 3), dtype=torch.float64, device=dev)
    t = torch.empty((P, 3), dtype=torch.float64, device=dev)
    n_in = torch.zeros((P,), dtype=torch.int32, device=dev)
    mask = torch.zeros((max(N, 1),), dtype=torch.uint8, device=dev)
    ok = torch.zeros((P,), dtype=torch.int32, device=dev)
    check(lib.pp_pnp_ransac_f64(ctx.handle, P, _ptr(offsets), N, _ptr(obj), _ptr(img), _ptr(K4), int(iterations), float(reproj_error),
                                int(seed) & 0xFFFFFFFFFFFFFFFF, int(points_per_vote), _ptr(ws), _ptr(R), _ptr(t), _ptr(n_in), _ptr(mask),
                                _ptr(ok)), ctx.handle, "pp_pnp_ransac_f64")
    return R, t, n_in, mask[:N], ok
