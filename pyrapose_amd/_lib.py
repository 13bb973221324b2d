"""ctypes binding of libpyrapose_hip.so (the C ABI declared in include/pyrapose_hip.h).

There is NO fallback: if the shared library is missing or fails to load, importing this module
raises ImportError, and every op raises on a non-zero status.  Negative statuses (argument / shape
errors caught on the host) become ValueError -- mirroring the ValueError the reference's Cython
boundary raises on a wrong dtype/ndim (utils/compute_overlap.pyx:13-16) -- positive ones
(hipError_t) become RuntimeError.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PP_LIB (development only): another build of the library beside the default one, for same-box A/B runs of two builds through
# the same Python (tools/ab_lib.sh); entry points such a build lacks are left unbound and their features switch off
LIB_PATH = os.path.join(_HERE, os.environ.get("PP_LIB") or "libpyrapose_hip.so")

PP_MAX_SEG = 5


class RowSpace(C.Structure):
    _fields_ = [("n_img", C.c_int), ("n_seg", C.c_int), ("h", C.c_int * PP_MAX_SEG), ("w", C.c_int * PP_MAX_SEG)]

    @classmethod
    def make(cls, n_img, shapes):
        rs = cls()
        rs.n_img = int(n_img)
        rs.n_seg = len(shapes)
        for i, (h, w) in enumerate(shapes):
            rs.h[i] = int(h)
            rs.w[i] = int(w)
        return rs

    def rows(self):
        return sum(self.n_img * self.h[i] * self.w[i] for i in range(self.n_seg))

    def shapes(self):
        return [(self.h[i], self.w[i]) for i in range(self.n_seg)]


class ConvDesc(C.Structure):
    _fields_ = [("in_", RowSpace), ("out", RowSpace), ("cin", C.c_int), ("cout", C.c_int), ("kh", C.c_int),
                ("kw", C.c_int), ("stride", C.c_int), ("pad_t", C.c_int), ("pad_l", C.c_int), ("ld_x", C.c_int),
                ("ld_y", C.c_int), ("ld_w", C.c_int)]


class ParamDesc(C.Structure):
    _fields_ = [("offset", C.c_longlong), ("count", C.c_longlong), ("ld", C.c_int), ("trainable", C.c_int),
                ("scale_off", C.c_longlong), ("l2", C.c_float)]


class SplitJob(C.Structure):
    _fields_ = [("w", C.c_void_p), ("fwd_hi", C.c_void_p), ("fwd_lo", C.c_void_p), ("dg_hi", C.c_void_p), ("dg_lo", C.c_void_p),
                ("taps", C.c_int), ("cin", C.c_int), ("cout", C.c_int), ("ld_w", C.c_int), ("tile_begin", C.c_int), ("reserved", C.c_int)]


class TView(C.Structure):
    """pp_tview: a tensor as float32 or as bf16 (hi, lo) planes of the same [rows][ld] geometry"""
    _fields_ = [("f32", C.c_void_p), ("hi", C.c_void_p), ("lo", C.c_void_p)]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "pyrapose_amd: %s not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C pyrapose_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    try:
        return C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise ImportError("pyrapose_amd: cannot load %s: %s" % (LIB_PATH, e))


lib = _load()

_p = C.c_void_p
_i = C.c_int
_f = C.c_float
_d = C.c_double
_ll = C.c_longlong
_sz = C.c_size_t

_SIGS = {
    "pp_ctx_create": (_i, [C.POINTER(_p), _i, _p]),
    "pp_ctx_destroy": (None, [_p]),
    "pp_ctx_set_stream": (_i, [_p, _p]),
    "pp_last_error_string": (C.c_char_p, [_p]),
    "pp_version": (C.c_char_p, []),
    "pp_device_info": (_i, [_p, C.POINTER(_i), C.c_char_p, _i]),
    "pp_conv2d_nhwc_fwd": (_i, [_p, C.POINTER(ConvDesc), _p, _p, _p, _p, _i, _i, _p]),
    "pp_conv2d_nhwc_bwd_data": (_i, [_p, C.POINTER(ConvDesc), _p, _p, _p, _i, _p, _i, _p]),
    "pp_conv2d_nhwc_bwd_weight": (_i, [_p, C.POINTER(ConvDesc), _p, _p, _p, _p]),
    "pp_conv_split_weights_bf16x3": (_i, [_p, C.POINTER(ConvDesc), _p, _p, _p, _p, _p]),
    "pp_split_planes_bf16x3": (_i, [_p, _sz, _p, _p, _p]),
    "pp_ctx_set_planes_format": (_i, [_p, _i]),
    "pp_convert_planes": (_i, [_p, _sz, _p, _p, _i, _p, _p, _i, _p, _i, _p]),
    "pp_split_planes_scaled_bf16x3": (_i, [_p, _sz, _p, _p, _p, _p]),
    "pp_grad_scale_from_counts": (_i, [_p, _p, _i, _p]),
    "pp_grad_scale_from_counts_adj": (_i, [_p, _p, _i, _p, _i]),
    "pp_ctx_set_grad_scale": (_i, [_p, _p]),
    "pp_planes_stats": (_i, [_p, _p, _p, _ll, _i, _i, _p, _p]),
    "pp_comm_available": (_i, []),
    "pp_comm_unique_id": (_i, [_p, _p]),
    "pp_comm_init": (_i, [_p, _i, _i, _p, C.POINTER(_p)]),
    "pp_comm_destroy": (_i, [_p]),
    "pp_allreduce_bucket": (_i, [_p, _p, _p, C.c_size_t]),
    "pp_allreduce_counts": (_i, [_p, _p, _p, _i]),
    "pp_ctx_set_workspace": (_i, [_p, _p, C.c_size_t]),
    "pp_ctx_set_split_capture": (_i, [_p, _p, _p]),
    "pp_row_block_list": (_i, [_p, _p, _i, _i, _i, _p, _p]),
    "pp_ctx_set_row_block_skip": (_i, [_p, _p, _p]),
    "pp_ctx_set_row_block_out": (_i, [_p, _p, _p]),
    "pp_ctx_set_row_block_lazy": (_i, [_p, _i, _i]),
    "pp_positive_row_blocks": (_i, [_p, _p, _i, _i, _p, _p]),
    "pp_row_block_dilate": (_i, [_p, _p, _p, _p]),
    "pp_row_block_list_planes": (_i, [_p, _p, _p, _i, _i, _i, _p, _p]),
    "pp_row_block_list_planes_within": (_i, [_p, _p, _p, _i, _i, _i, _p, _p, _p]),
    "pp_ctx_set_epilogue_planes": (_i, [_p, _p, _p, _p]),
    "pp_add_n_v": (_i, [_p, _sz, C.POINTER(TView), C.POINTER(TView), C.POINTER(TView), C.POINTER(TView)]),
    "pp_relu_fwd_v": (_i, [_p, _sz, C.POINTER(TView), C.POINTER(TView)]),
    "pp_upsample_nearest_add_fwd_v": (_i, [_p, _i, _i, _i, _i, _i, _i, C.POINTER(TView), C.POINTER(TView), C.POINTER(TView)]),
    "pp_upsample_nearest_add_bwd_v": (_i, [_p, _i, _i, _i, _i, _i, _i, C.POINTER(TView), C.POINTER(TView), C.POINTER(TView)]),
    "pp_merge_planes_bf16x3": (_i, [_p, _sz, _p, _p, _p]),
    "pp_warp_affine_u8": (_i, [_p, _i, _i, _i, _i, C.POINTER(_d), _i, _i, _i, _p, _p]),
    "pp_resize_scale": (_i, [_i, _i, _i, _i, C.POINTER(_d)]),
    "pp_resize_linear_u8": (_i, [_p, _i, _i, _i, _i, _d, _i, _i, _p, _p]),
    "pp_conv_split_weights_bf16x3_batch": (_i, [_p, _i, _p, _i]),
    "pp_conv2d_nhwc_fwd_bf16x3": (_i, [_p, C.POINTER(ConvDesc), _p, _p, _p, _p, _p, _p, _p, _i, _i, _p, _p, _p]),
    "pp_conv2d_nhwc_bwd_data_bf16x3": (_i, [_p, C.POINTER(ConvDesc), _p, _p, _p, _p, _p, _p, _i, _p, _i, _p, _p, _p]),
    "pp_conv2d_nhwc_bwd_weight_bf16x3": (_i, [_p, C.POINTER(ConvDesc), _p, _p, _p, _p, _p, _p, _p, _p]),
    "pp_maxpool3x3s2_fwd": (_i, [_p, _i, _i, _i, _i, _p, _i, _i, _p]),
    "pp_upsample_nearest_add_fwd": (_i, [_p, _i, _i, _i, _i, _i, _i, _p, _p, _p]),
    "pp_upsample_nearest_add_bwd": (_i, [_p, _i, _i, _i, _i, _i, _i, _p, _p, _p]),
    "pp_add_n": (_i, [_p, _sz, _p, _p, _p, _p]),
    "pp_relu_fwd": (_i, [_p, C.c_size_t, _p, _p]),
    "pp_preprocess_caffe_u8": (_i, [_p, _i, _i, _i, C.POINTER(C.c_int), _p, _p]),
    "pp_preprocess_caffe_u8_padded": (_i, [_p, _i, _i, _i, _i, _i, _i, C.POINTER(C.c_int), _p, _p]),
    "pp_pack_rgb_to_4_padded": (_i, [_p, _i, _i, _i, _i, _i, _i, _p, _p]),
    "pp_stem7x7s2_fwd_bf16x3": (_i, [_p, _i, _i, _i, _i, _i, _p, _p, _p, _i, _p, _i, _p, _i]),
    "pp_pack_rgb_to_4": (_i, [_p, _sz, _p, _p]),
    "pp_export_head": (_i, [_p, C.POINTER(RowSpace), _i, _i, _p, _i, _i, _p]),
    "pp_count_positives": (_i, [_p, _sz, _p, _sz, _i, _p, _sz, _i, _p, _p]),
    "pp_sigmoid_focal_fwd_bwd": (_i, [_p, C.POINTER(RowSpace), _i, _i, _p, _i, _p, _f, _f, _p, _f, _p, _p]),
    "pp_orth_smoothl1_fwd_bwd": (_i, [_p, C.POINTER(RowSpace), _i, _p, _i, _p, _f, _f, _p, _f, _p, _p]),
    "pp_optimizer_create": (_i, [_p, C.POINTER(_p), C.POINTER(ParamDesc), _i, _ll]),
    "pp_optimizer_destroy": (None, [_p]),
    "pp_grad_global_norm": (_i, [_p, _p, _p, _p, _p, _p, _p]),
    "pp_adam_step_clipnorm": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _f, _f, _f, _f, _f, _ll]),
    "pp_generate_base_anchors_host": (_i, [_i, C.POINTER(_f), _i, C.POINTER(_f), _i, C.POINTER(_d)]),
    "pp_anchors_shift_f64": (_i, [_p, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), _i, C.POINTER(_d), _p]),
    "pp_anchors_shift_f32": (_i, [_p, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), _i, C.POINTER(_d), _p]),
    "pp_compute_overlap_f64": (_i, [_p, _i, _p, _i, _p, _p]),
    "pp_compute_gt_annotations": (_i, [_p, _i, _p, _i, _p, _d, _d, _p, _p]),
    "pp_anchor_targets": (_i, [_p, _i, _p, _i, C.POINTER(_i), _p, _p, _p, _p, _p, _i, _i, C.POINTER(_i), C.POINTER(_i), _i, _d, _d,
                               _i, _i, _p, _p, _p]),
    "pp_project_box3d_host": (_i, [C.POINTER(_d), C.POINTER(_d), C.POINTER(_d), C.POINTER(_d)]),
    "pp_pil_nearest_index_host": (_i, [_i, _i, C.POINTER(_i)]),
    "pp_box3d_decode": (_i, [_p, _i, _i, _p, _p, _p]),
    "pp_score_threshold_compact": (_i, [_p, _i, _i, _i, _p, _f, _i, _p, _p]),
    "pp_filter_workspace_bytes": (_sz, [_i, _i, _i]),
    "pp_pose_error_workspace_bytes": (_sz, [_i, _i]),
    "pp_filter_detections": (_i, [_p, _i, _i, _p, _p, _p, _f, _f, _i, _p, _p, _p, _p, _p]),
    "pp_pose_add_f64": (_i, [_p, _i, _i, _p, _p, _p, _p, _p, _p, _p]),
    "pp_pose_adi_f64": (_i, [_p, _i, _i, _p, _p, _p, _p, _p, _p, _p]),
    "pp_filter_detections_batch": (_i, [_p, _i, _i, _i, _p, _p, _p, _f, _f, _i, _p, _p, _p, _p, _p]),
    "pp_pnp_ransac_workspace_bytes": (_sz, [_i, _i]),
    "pp_pnp_ransac_f64": (_i, [_p, _i, _p, _i, _p, _p, _p, _i, _d, C.c_ulonglong, _i, _p, _p, _p, _p, _p, _p]),
}

EXPORTS = sorted(_SIGS)

MISSING = []
for _name, (_res, _args) in _SIGS.items():
    try:
        _fn = getattr(lib, _name)  # AttributeError here = the .so does not export what the header declares
    except AttributeError:
        if not os.environ.get("PP_LIB"):
            raise
        MISSING.append(_name)
        continue
    _fn.restype = _res
    _fn.argtypes = _args


def check(rc, ctx=None, what=""):
    if rc == 0:
        return
    msg = ""
    if ctx is not None:
        s = lib.pp_last_error_string(ctx)
        msg = s.decode("utf-8", "replace") if s else ""
    if rc < 0:
        raise ValueError("%s failed (%d): %s" % (what or "pyrapose_hip", rc, msg))
    raise RuntimeError("%s failed (hipError %d): %s" % (what or "pyrapose_hip", rc, msg))
