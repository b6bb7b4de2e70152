"""ctypes binding of libssdk.so (C-ABI: include/ssdk.h) -- the only door to the HIP kernels.

There is NO fallback: if the shared library is missing or a call fails, an exception is raised.
Tensors are passed as raw ``data_ptr()`` values plus the caller's current HIP stream, so every call is
asynchronous and hipGraph-capturable.  This module is the `ssds._C` the reference names but never
ships (reference ssds/modeling/layers/box.py:3-4, 419-421, 483-485).
"""
import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "csrc", "libssdk.so")

MAX_LEVELS = 8
MAX_ANCHORS = 16
MAX_TOPN = 1024
MAX_NDET = 1024
MAX_NMS_N = 8192
MAX_GT = 256

F32, BF16, F16 = 0, 1, 2
ACT = {"none": 0, "relu": 1, "relu6": 2, "silu": 3, "sigmoid": 4}
_DTYPES = {torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16}
U8 = 3  # ssdk_preprocess source only


class Level(ctypes.Structure):
    _fields_ = [
        ("cls", ctypes.c_void_p),
        ("box", ctypes.c_void_p),
        ("A", ctypes.c_int32),
        ("C", ctypes.c_int32),
        ("H", ctypes.c_int32),
        ("W", ctypes.c_int32),
        ("stride", ctypes.c_int32),
        ("anchors", ctypes.c_float * (MAX_ANCHORS * 4)),
    ]


class ConvDesc(ctypes.Structure):
    _fields_ = [
        ("x", ctypes.c_void_p), ("w", ctypes.c_void_p), ("scale", ctypes.c_void_p), ("bias", ctypes.c_void_p),
        ("residual", ctypes.c_void_p), ("y", ctypes.c_void_p), ("y2", ctypes.c_void_p),
        ("N", ctypes.c_int32), ("Cin", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32),
        ("Cout", ctypes.c_int32), ("k", ctypes.c_int32), ("stride", ctypes.c_int32), ("groups", ctypes.c_int32),
        ("act", ctypes.c_int32), ("act2", ctypes.c_int32), ("split", ctypes.c_int32),
        ("dtype", ctypes.c_int32), ("in_layout", ctypes.c_int32), ("out_layout", ctypes.c_int32),
        ("res_mode", ctypes.c_int32), ("w_frag", ctypes.c_void_p),
    ]


class MbConvDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in (
        "x", "y", "w_expand", "scale_expand", "bias_expand", "w_dw", "bias_dw", "w_project",
        "scale_project", "bias_project")] + [(n, ctypes.c_int32) for n in (
            "N", "H", "W", "Cin", "Chid", "Cout", "stride", "residual", "dtype", "stem", "variant",
            "image_nw")] + [("w_image", ctypes.c_void_p), ("w_image_bytes", ctypes.c_size_t)]


class FuseDesc(ctypes.Structure):
    _fields_ = [("a", ctypes.c_void_p), ("b", ctypes.c_void_p), ("c", ctypes.c_void_p), ("y", ctypes.c_void_p),
                ("w0", ctypes.c_float), ("w1", ctypes.c_float), ("w2", ctypes.c_float),
                ("mode_b", ctypes.c_int32), ("mode_c", ctypes.c_int32),
                ("N", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32), ("C", ctypes.c_int32),
                ("hb", ctypes.c_int32), ("wb", ctypes.c_int32), ("hc", ctypes.c_int32), ("wc", ctypes.c_int32),
                ("dtype", ctypes.c_int32)]


class StemDesc(ctypes.Structure):
    _fields_ = [("x", ctypes.c_void_p), ("w", ctypes.c_void_p), ("scale", ctypes.c_void_p), ("bias", ctypes.c_void_p),
                ("y", ctypes.c_void_p)] + [(n, ctypes.c_int32) for n in (
                    "N", "H", "W", "Cin", "Cout", "act", "dtype", "in_layout")]


class PoolDesc(ctypes.Structure):
    _fields_ = [("x", ctypes.c_void_p), ("y", ctypes.c_void_p)] + [(n, ctypes.c_int32) for n in (
        "N", "H", "W", "C", "dtype", "pad")]


class XpairDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ("x", "y", "w1", "scale1", "bias1", "w2", "scale2", "bias2")] + [
        (n, ctypes.c_int32) for n in ("N", "H", "W", "Cin", "Cmid", "Cout", "act1", "act2", "dtype", "pad")] + [
        ("w1_frag", ctypes.c_void_p), ("w2_frag", ctypes.c_void_p)]


class Op(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int32), ("lane", ctypes.c_int32), ("conv", ConvDesc), ("mb", MbConvDesc),
                ("fuse", FuseDesc), ("stem", StemDesc), ("pool", PoolDesc), ("xpair", XpairDesc)]


OP_CONV, OP_MBCONV, OP_FUSE, OP_STEM7, OP_POOL, OP_XPAIR = 0, 1, 2, 3, 4, 5
FUSE_SAME, FUSE_UP2, FUSE_POOL2 = 0, 1, 2
NCHW, NHWC = 0, 1


ABI_VERSION = 244  # include/ssdk.h SSDK_VERSION this module's ctypes mirrors and prototypes are written for


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libssdk.so not found at {} -- build it first: `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C ssds.pytorch_amd/csrc` (there is no CPU fallback)".format(LIB_PATH)
        )
    lib = ctypes.CDLL(LIB_PATH)
    c = ctypes
    vp, i32, f32, sz = c.c_void_p, c.c_int, c.c_float, c.c_size_t
    lib.ssdk_version.restype = i32
    # version FIRST: a library older than ABI 230 has no ssdk_struct_size, and a bare AttributeError from
    # the symbol lookup below would hide what is wrong
    try:
        have = int(lib.ssdk_version())
    except AttributeError:
        have = 0
    if have < ABI_VERSION:
        raise ImportError("libssdk.so at {} is ABI {} but ssds/_native.py is written for ABI {}: rebuild it "
                          "(`make -C ssds.pytorch_amd/csrc`)".format(LIB_PATH, have, ABI_VERSION))
    lib.ssdk_struct_size.argtypes = [i32]
    lib.ssdk_struct_size.restype = sz
    # the ctypes mirrors below must have the layout the library was BUILT with: a shorter struct would be read past its end
    for which, cls in enumerate((Level, ConvDesc, MbConvDesc, FuseDesc, StemDesc, PoolDesc, XpairDesc, Op)):
        want = lib.ssdk_struct_size(which)
        if want != ctypes.sizeof(cls):
            raise ImportError("libssdk.so at {} was built against another include/ssdk.h: sizeof({}) is {} there, {} in ssds/_native.py"
                              .format(LIB_PATH, cls.__name__, want, ctypes.sizeof(cls)))
    lib.ssdk_last_error.restype = c.c_char_p
    lib.ssdk_abi_check.argtypes = [i32, sz]
    lib.ssdk_abi_check.restype = i32
    if lib.ssdk_abi_check(ABI_VERSION, ctypes.sizeof(Op)) != 0:
        raise ImportError("libssdk.so at {}: {}".format(LIB_PATH, lib.ssdk_last_error().decode()))
    lib.ssdk_last_kernel.restype = c.c_char_p
    lib.ssdk_mbk_image_bytes.argtypes = [i32] * 6 + [c.POINTER(i32)]
    lib.ssdk_mbk_image_bytes.restype = sz
    lib.ssdk_fuse.argtypes = [c.POINTER(FuseDesc), vp]
    lib.ssdk_fuse.restype = i32
    lib.ssdk_pw_prepare.argtypes = [vp, vp, vp, i32, i32, i32, vp]
    lib.ssdk_pw_forward.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]
    lib.ssdk_pw_stats_workspace_bytes.argtypes = [i32] * 4
    lib.ssdk_pw_stats_workspace_bytes.restype = sz
    lib.ssdk_pw_forward_stats.argtypes = [vp, vp, vp, vp, vp, vp, sz, i32, i32, i32, i32, i32, vp]
    lib.ssdk_pw_forward_stats.restype = i32
    lib.ssdk_bn_act_train_fwd_sums.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, i32, i32, i32, f32, f32, i32, i32, vp]
    lib.ssdk_bn_act_train_fwd_sums.restype = i32
    lib.ssdk_pw_wgrad_workspace_bytes.argtypes = [i32] * 4
    lib.ssdk_pw_wgrad_workspace_bytes.restype = sz
    lib.ssdk_pw_wgrad.argtypes = [vp, vp, vp, vp, sz, i32, i32, i32, i32, i32, vp]
    lib.ssdk_sgd_step.argtypes = [i32, vp, vp, vp, vp, vp, f32, f32, f32, i32, vp, vp]
    lib.ssdk_sgd_step.restype = i32
    lib.ssdk_im2col3x3.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, vp]
    lib.ssdk_col2im3x3.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, vp]
    lib.ssdk_im2col3x3_folded.argtypes = lib.ssdk_col2im3x3_folded.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, vp]
    lib.ssdk_im2col3x3_folded.restype = lib.ssdk_col2im3x3_folded.restype = i32
    lib.ssdk_stem3x3s2_wgrad_workspace_bytes.argtypes = [i32, i32]
    lib.ssdk_stem3x3s2_wgrad_workspace_bytes.restype = sz
    lib.ssdk_stem3x3s2_fwd.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]
    lib.ssdk_stem3x3s2_wgrad.argtypes = [vp, vp, vp, vp, sz, i32, i32, i32, i32, i32, i32, vp]
    lib.ssdk_stem3x3s2_fwd.restype = lib.ssdk_stem3x3s2_wgrad.restype = i32
    lib.ssdk_pack_conv3x3.argtypes = [vp, vp, i32, vp, vp, i32, i32, vp, vp, vp, i32, vp]
    lib.ssdk_pack_conv3x3.restype = i32
    lib.ssdk_pack_conv3x3_dgrad.argtypes = [vp, i32, vp, i32, i32, i32, vp, vp, i32, vp]
    lib.ssdk_pack_conv3x3_dgrad.restype = i32
    lib.ssdk_concat_nchw_to_nhwc.argtypes = [vp, i32, vp, i32, vp, i32, i32, i32, i32, vp]
    lib.ssdk_concat_nchw_to_nhwc.restype = i32
    for _n in ("ssdk_pw_prepare", "ssdk_pw_forward", "ssdk_pw_wgrad", "ssdk_im2col3x3", "ssdk_col2im3x3"):
        getattr(lib, _n).restype = i32
    lib.ssdk_dwconv_fwd.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]
    lib.ssdk_dwconv_fwd_stats_workspace_bytes.argtypes = [i32] * 6
    lib.ssdk_dwconv_fwd_stats_workspace_bytes.restype = sz
    lib.ssdk_dwconv_fwd_stats.argtypes = [vp, vp, vp, vp, vp, sz, i32, i32, i32, i32, i32, i32, vp]
    lib.ssdk_dwconv_fwd_stats.restype = i32
    lib.ssdk_dwconv_affine_supported.argtypes = [i32] * 6
    lib.ssdk_dwconv_affine_supported.restype = i32
    lib.ssdk_dwconv_fwd_affine.argtypes = [vp, vp, i32, vp, vp, vp, vp, sz, i32, i32, i32, i32, i32, i32, vp]
    lib.ssdk_dwconv_fwd_affine.restype = i32
    lib.ssdk_dwconv_bwd_weight_affine.argtypes = [vp, vp, i32, vp, vp, vp, sz, i32, i32, i32, i32, i32, i32, vp]
    lib.ssdk_dwconv_bwd_weight_affine.restype = i32
    lib.ssdk_bn_act_train_stats.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, i32, i32, i32, f32, f32, i32, vp]
    lib.ssdk_bn_act_train_stats.restype = i32
    lib.ssdk_dwconv_bwd_data.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]
    lib.ssdk_dwconv_bwd_weight_workspace_bytes.argtypes = [i32] * 5
    lib.ssdk_dwconv_bwd_weight_workspace_bytes.restype = sz
    lib.ssdk_dwconv_bwd_weight.argtypes = [vp, vp, vp, vp, sz, i32, i32, i32, i32, i32, i32, vp]
    lib.ssdk_dwconv_plan.argtypes = [i32] * 7 + [ctypes.POINTER(i32)]
    for _n in ("ssdk_dwconv_fwd", "ssdk_dwconv_bwd_data", "ssdk_dwconv_bwd_weight", "ssdk_dwconv_plan"):
        getattr(lib, _n).restype = i32
    lib.ssdk_bn_workspace_bytes.argtypes = [i32, i32]
    lib.ssdk_bn_workspace_bytes.restype = sz
    lib.ssdk_bn_train_fwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, i32, i32, i32, f32, f32, i32, vp]
    lib.ssdk_bn_train_bwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, i32, i32, i32, i32, vp]
    lib.ssdk_bn_act_train_fwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, i32, i32, i32, f32, f32, i32, i32, vp]
    lib.ssdk_bn_act_train_bwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, i32, i32, i32, i32, i32, vp]
    lib.ssdk_bn_act_train_fwd.restype = i32
    lib.ssdk_bn_act_train_bwd.restype = i32
    lib.ssdk_bn_train_fwd.restype = i32
    lib.ssdk_bn_train_bwd.restype = i32
    lib.ssdk_preprocess.argtypes = [vp, i32, i32, i32, i32, i32, i32, c.POINTER(f32), c.POINTER(f32), vp, i32, vp]
    lib.ssdk_preprocess.restype = i32
    lib.ssdk_conv_stem7.argtypes = [c.POINTER(StemDesc), vp]
    lib.ssdk_conv_stem7.restype = i32
    lib.ssdk_maxpool3x3s2.argtypes = [c.POINTER(PoolDesc), vp]
    lib.ssdk_maxpool3x3s2.restype = i32
    lib.ssdk_set_op_profiling.argtypes = [i32]
    lib.ssdk_get_op_timings.argtypes = [c.POINTER(f32), c.POINTER(c.c_char_p), i32]
    lib.ssdk_device_info.argtypes = [c.POINTER(i32), c.POINTER(i32), c.POINTER(sz), c.c_char_p, i32]
    lib.ssdk_generate_anchors.argtypes = [i32, c.POINTER(f32), i32, c.POINTER(f32), i32, c.POINTER(f32)]
    lib.ssdk_decode_workspace_bytes.restype = sz
    lib.ssdk_decode_workspace_bytes.argtypes = [c.POINTER(Level), i32, i32, i32, i32]
    lib.ssdk_decode.argtypes = [c.POINTER(Level), i32, i32, f32, i32, i32, vp, vp, vp, vp, sz, vp]
    lib.ssdk_nms_workspace_bytes.restype = sz
    lib.ssdk_nms_workspace_bytes.argtypes = [i32, i32, i32]
    lib.ssdk_nms.argtypes = [vp, vp, vp, i32, i32, f32, i32, i32, vp, vp, vp, vp, sz, vp]
    lib.ssdk_decode_nms_workspace_bytes.restype = sz
    lib.ssdk_decode_nms_workspace_bytes.argtypes = [c.POINTER(Level), i32, i32, i32, i32, i32]
    lib.ssdk_decode_nms.argtypes = [c.POINTER(Level), i32, i32, i32, f32, i32, i32, f32, i32, i32,
                                    vp, vp, vp, vp, vp, vp, vp, sz, vp]
    lib.ssdk_match_targets.argtypes = [vp, i32, i32, c.POINTER(f32), i32, i32, i32, i32, i32, f32, f32,
                                       f32, vp, vp, vp, vp]
    lib.ssdk_match_targets_by_scale.argtypes = [vp, i32, i32, c.POINTER(f32), i32, i32, i32, i32, i32, f32,
                                                f32, i32, vp, vp, vp, vp]
    lib.ssdk_match_loss_workspace_bytes.restype = sz
    lib.ssdk_match_loss_workspace_bytes.argtypes = [i32] * 4
    lib.ssdk_match_loss.restype = i32
    lib.ssdk_match_loss.argtypes = [vp, i32, i32, c.POINTER(f32), i32, i32, i32, i32, i32, i32, f32, f32, f32,
                                    vp, vp, i32, f32, f32, f32, i32, vp, vp, vp, vp, sz, vp]
    lib.ssdk_match_multibox_loss_workspace_bytes.restype = sz
    lib.ssdk_match_multibox_loss_workspace_bytes.argtypes = [i32] * 4
    lib.ssdk_match_multibox_loss.restype = i32
    lib.ssdk_match_multibox_loss.argtypes = [vp, i32, i32, c.POINTER(f32), i32, i32, i32, i32, i32, i32, f32, f32, f32,
                                             vp, vp, i32, f32, f32, i32, vp, vp, vp, vp, sz, vp]
    lib.ssdk_debug_lds_probe.restype = i32
    lib.ssdk_debug_lds_probe.argtypes = [vp, vp]
    lib.ssdk_set_decode_tail_stream.restype = i32
    lib.ssdk_set_decode_tail_stream.argtypes = [vp]
    lib.ssdk_ctx_create.restype = vp
    lib.ssdk_ctx_create.argtypes = []
    lib.ssdk_ctx_destroy.restype = None
    lib.ssdk_ctx_destroy.argtypes = [vp]
    for _n, _a in (("ssdk_ctx_set_tail_stream", [vp, vp]), ("ssdk_ctx_set_side_lane", [vp, i32]),
                   ("ssdk_ctx_set_profiling", [vp, i32]), ("ssdk_ctx_get_timings", [vp, i32, c.POINTER(f32), i32]),
                   ("ssdk_ctx_set_op_profiling", [vp, i32]),
                   ("ssdk_ctx_get_op_timings", [vp, c.POINTER(f32), c.POINTER(c.c_char_p), i32]),
                   ("ssdk_ctx_get_tail_stamps", [vp, c.POINTER(c.c_ulonglong), i32]),
                   ("ssdk_run_ops_ctx", [vp, c.POINTER(Op), i32, vp, sz, vp]),
                   ("ssdk_decode_nms_ctx", [vp, c.POINTER(Level), i32, i32, i32, f32, i32, i32, f32, i32, i32,
                                            vp, vp, vp, vp, vp, vp, vp, sz, vp])):
        getattr(lib, _n).restype = i32
        getattr(lib, _n).argtypes = _a
    lib.ssdk_map_match.restype = i32
    lib.ssdk_map_match.argtypes = [vp, vp, vp, i32, i32, vp, i32, i32, f32, f32, vp, vp, vp, vp]
    lib.ssdk_map_average_precision.restype = i32
    lib.ssdk_map_average_precision.argtypes = [vp, vp, vp, i32, vp, vp]
    lib.ssdk_weight_frag_bytes.restype = sz
    lib.ssdk_weight_frag_bytes.argtypes = [i32, i32]
    lib.ssdk_conv_workspace_bytes.restype = sz
    lib.ssdk_conv_workspace_bytes.argtypes = [i32] * 8
    lib.ssdk_conv_bn_act.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32,
                                     vp, vp, sz, vp]
    lib.ssdk_conv.argtypes = [c.POINTER(ConvDesc), vp, sz, vp]
    lib.ssdk_conv.restype = i32
    lib.ssdk_conv_sequence.argtypes = [c.POINTER(ConvDesc), i32, vp, sz, vp]
    lib.ssdk_conv_sequence.restype = i32
    lib.ssdk_mbconv.argtypes = [c.POINTER(MbConvDesc), vp]
    lib.ssdk_mbconv.restype = i32
    lib.ssdk_xpair.argtypes = [c.POINTER(XpairDesc), vp]
    lib.ssdk_xpair.restype = i32
    lib.ssdk_run_ops.argtypes = [c.POINTER(Op), i32, vp, sz, vp]
    lib.ssdk_run_ops.restype = i32
    lib.ssdk_set_profiling.argtypes = [i32]
    lib.ssdk_get_timings.argtypes = [i32, c.POINTER(f32), i32]
    for name in ("ssdk_set_profiling", "ssdk_get_timings", "ssdk_device_info", "ssdk_generate_anchors", "ssdk_decode", "ssdk_nms",
                 "ssdk_decode_nms", "ssdk_match_targets", "ssdk_match_targets_by_scale", "ssdk_conv_bn_act"):
        getattr(lib, name).restype = i32
    return lib


lib = _load()
EXPORTS = ("ssdk_version", "ssdk_struct_size", "ssdk_abi_check", "ssdk_last_error", "ssdk_last_kernel", "ssdk_set_op_profiling", "ssdk_get_op_timings", "ssdk_device_info", "ssdk_generate_anchors",
           "ssdk_decode_workspace_bytes", "ssdk_decode", "ssdk_nms_workspace_bytes", "ssdk_nms",
           "ssdk_decode_nms_workspace_bytes", "ssdk_decode_nms", "ssdk_match_targets",
           "ssdk_match_targets_by_scale", "ssdk_match_loss_workspace_bytes", "ssdk_match_loss",
           "ssdk_match_multibox_loss_workspace_bytes", "ssdk_match_multibox_loss",
           "ssdk_map_match", "ssdk_map_average_precision", "ssdk_set_decode_tail_stream", "ssdk_debug_lds_probe",
           "ssdk_ctx_create", "ssdk_ctx_destroy", "ssdk_ctx_set_tail_stream", "ssdk_ctx_set_side_lane", "ssdk_ctx_set_profiling",
           "ssdk_ctx_get_timings", "ssdk_ctx_set_op_profiling", "ssdk_ctx_get_op_timings", "ssdk_ctx_get_tail_stamps",
           "ssdk_run_ops_ctx", "ssdk_decode_nms_ctx",
           "ssdk_weight_frag_bytes", "ssdk_conv_workspace_bytes", "ssdk_conv", "ssdk_conv_sequence", "ssdk_mbconv", "ssdk_mbk_image_bytes", "ssdk_xpair", "ssdk_fuse", "ssdk_preprocess", "ssdk_dwconv_fwd_stats_workspace_bytes", "ssdk_dwconv_fwd_stats", "ssdk_dwconv_affine_supported", "ssdk_dwconv_fwd_affine", "ssdk_dwconv_bwd_weight_affine", "ssdk_bn_act_train_stats", "ssdk_pw_prepare", "ssdk_pw_forward", "ssdk_pw_stats_workspace_bytes", "ssdk_pw_forward_stats", "ssdk_bn_act_train_fwd_sums", "ssdk_pw_wgrad_workspace_bytes", "ssdk_pw_wgrad", "ssdk_im2col3x3", "ssdk_col2im3x3", "ssdk_im2col3x3_folded", "ssdk_col2im3x3_folded", "ssdk_stem3x3s2_wgrad_workspace_bytes", "ssdk_stem3x3s2_fwd", "ssdk_stem3x3s2_wgrad", "ssdk_pack_conv3x3", "ssdk_pack_conv3x3_dgrad", "ssdk_concat_nchw_to_nhwc", "ssdk_sgd_step", "ssdk_dwconv_fwd", "ssdk_dwconv_bwd_data",
           "ssdk_dwconv_bwd_weight_workspace_bytes", "ssdk_dwconv_bwd_weight", "ssdk_dwconv_plan", "ssdk_bn_workspace_bytes",
           "ssdk_bn_train_fwd", "ssdk_bn_train_bwd", "ssdk_bn_act_train_fwd", "ssdk_bn_act_train_bwd", "ssdk_conv_stem7", "ssdk_maxpool3x3s2", "ssdk_run_ops", "ssdk_conv_bn_act", "ssdk_set_profiling", "ssdk_get_timings")


class Context(object):
    """A caller-owned ``ssdk_ctx`` (include/ssdk.h "Contexts"): the HIP objects behind the multi-launch entry points
    (side stream + fork/join events of the plan executor, tail-stream fork events and the profiling rings of the
    decode stage).  Bound to the device that is current when it is created; used by one host thread at a time --
    every ``Decoder`` and every ``ConvPlan`` owns one, so two threads / two devices share no state."""

    def __init__(self, device):
        self.device = torch.device(device)
        with torch.cuda.device(self.device):
            self.ptr = lib.ssdk_ctx_create()
        if not self.ptr:
            raise SsdkError("ssdk_ctx_create failed: " + lib.ssdk_last_error().decode())
        self.ptr = ctypes.c_void_p(self.ptr)

    def __del__(self):
        try:
            if getattr(self, "ptr", None):
                lib.ssdk_ctx_destroy(self.ptr)
                self.ptr = None
        except Exception:  # interpreter shutdown
            pass

    def _call(self, fn, *a):
        with torch.cuda.device(self.device):
            return fn(self.ptr, *a)

    def set_tail_stream(self, stream):
        check(self._call(lib.ssdk_ctx_set_tail_stream, ctypes.c_void_p(stream.cuda_stream) if stream is not None else None),
              "ctx_set_tail_stream")

    def set_side_lane(self, on):
        """True / False: the caller's explicit choice (a plan with side-stream chains respects it); None: back to the default
        (SSDK_SIDE_STREAM, and a plan that recorded side-stream chains turns the lane on again by itself)."""
        self.side_user = None if on is None else bool(on)
        self.side_auto = False
        check(self._call(lib.ssdk_ctx_set_side_lane, -1 if on is None else int(bool(on))), "ctx_set_side_lane")

    def auto_side_lane(self):
        """A plan whose small pyramid levels were recorded for the side stream (fused_conv.ConvPlan.launch): on, unless the
        caller chose (set_side_lane True / False) or the environment says SSDK_SIDE_STREAM=0."""
        if getattr(self, "side_user", None) is not None or getattr(self, "side_auto", False):
            return
        if os.environ.get("SSDK_SIDE_STREAM", "") == "0":
            return
        check(self._call(lib.ssdk_ctx_set_side_lane, 1), "ctx_set_side_lane")
        self.side_auto = True

    def set_profiling(self, on):
        """True / 1: hipEvents around every launch of the decode stage; 2: ONE interval around the whole stage (no event
        between its launches: an event costs ~4.6 us of GPU time on this stack and flushes caches between the kernels it
        separates), read back as timings_ms()[0]; False: off."""
        check(self._call(lib.ssdk_ctx_set_profiling, 2 if on == 2 else (1 if on else 0)), "ctx_set_profiling")

    def timings_ms(self, back=0):
        """(scan_kernel, tail_kernel | level_kernel, nms_kernel | 0) ms of the profiled decode_nms call ``back`` calls ago."""
        ms = (ctypes.c_float * 3)()
        check(self._call(lib.ssdk_ctx_get_timings, int(back), ms, 3), "ctx_get_timings")
        return float(ms[0]), float(ms[1]), float(ms[2])

    def set_op_profiling(self, on):
        check(self._call(lib.ssdk_ctx_set_op_profiling, 1 if on else 0), "ctx_set_op_profiling")

    def op_timings(self):
        ms = (ctypes.c_float * 128)()
        names = (ctypes.c_char_p * 128)()
        n = self._call(lib.ssdk_ctx_get_op_timings, ms, names, 128)
        if n < 0:
            raise SsdkError(lib.ssdk_last_error().decode())
        return [(names[i].decode() if names[i] else "", float(ms[i])) for i in range(n)]

    def tail_stamps(self, n=48):
        """Debug stamps (SSDK_TAIL_STAMPS=1): [0, 24) tail_kernel, [24, 48) scan kernel phases of workgroup 0 (shader clock);
        n = 48 + 2 * W also returns wall-clock (100 MHz) start / end pairs of the first W <= 4096 scan workgroups."""
        out = (ctypes.c_ulonglong * n)()
        check(self._call(lib.ssdk_ctx_get_tail_stamps, out, n), "ctx_get_tail_stamps")
        return [int(out[i]) for i in range(n)]


def op_timings():
    """[(kernel name, ms)] of the most recent ssdk_run_ops call made while op profiling was on (stream synchronised);
    the calling thread's default context (plans own their context: ``ConvPlan.ctx.op_timings()``)."""
    ms = (ctypes.c_float * 128)()
    names = (ctypes.c_char_p * 128)()
    n = lib.ssdk_get_op_timings(ms, names, 128)
    if n < 0:
        raise SsdkError(lib.ssdk_last_error().decode())
    return [(names[i].decode() if names[i] else "", float(ms[i])) for i in range(n)]


def last_kernel():
    """Name of the kernel this thread launched last (which variant a layer was dispatched to)."""
    return lib.ssdk_last_kernel().decode()


class SsdkError(RuntimeError):
    pass


def check(rc, what):
    if rc != 0:
        raise SsdkError("{} failed ({}): {}".format(what, rc, lib.ssdk_last_error().decode()))


def dtype_code(t):
    try:
        return _DTYPES[t.dtype]
    except KeyError:
        raise TypeError("unsupported dtype {} (float32, bfloat16, float16)".format(t.dtype))


def require_device(t, what):
    if not t.is_cuda:
        raise SsdkError(
            "{}: tensor is on '{}' -- the MI355X path has no CPU fallback; move it to a HIP device".format(
                what, t.device))
    return t


def stream_ptr(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


_ws_lock = threading.Lock()
_ws = {}


def workspace(device, nbytes):
    """Grow-only per-(device, stream) scratch buffer.  Kernels are stream-ordered, so consecutive
    calls on the same stream may share it."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    nbytes = max(int(nbytes), 256)
    with _ws_lock:
        buf = _ws.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(nbytes + nbytes // 4, dtype=torch.uint8, device=device)
            _ws[key] = buf
    return buf


def make_level(cls, box, stride, anchors):
    """anchors: CPU float tensor / ndarray [A,4].  cls [B, A*C, H, W], box [B, A*4, H, W]."""
    import numpy as np

    anc = np.ascontiguousarray(
        anchors.detach().cpu().numpy() if torch.is_tensor(anchors) else anchors, dtype=np.float32)
    A = int(anc.shape[0])
    if A > MAX_ANCHORS:
        raise SsdkError("at most {} anchors per location (got {})".format(MAX_ANCHORS, A))
    if cls.shape[1] % A or box.shape[1] != 4 * A:
        raise SsdkError("head channels {} / {} do not match {} anchors".format(cls.shape[1], box.shape[1], A))
    lv = Level()
    lv.cls = cls.data_ptr()
    lv.box = box.data_ptr()
    lv.A = A
    lv.C = int(cls.shape[1] // A)
    lv.H, lv.W = int(cls.shape[-2]), int(cls.shape[-1])
    lv.stride = int(stride)
    flat = anc.reshape(-1)
    for i in range(flat.shape[0]):
        lv.anchors[i] = float(flat[i])
    return lv


def set_profiling(on):
    check(lib.ssdk_set_profiling(1 if on else 0), "set_profiling")


def timings_ms(back=0):
    """(scan_kernel, tail_kernel | level_kernel, nms_kernel | 0) milliseconds of the profiled decode_nms call `back`
    calls before the most recent one (the calling thread's default context; a ``Decoder`` owns its own)."""
    ms = (ctypes.c_float * 3)()
    check(lib.ssdk_get_timings(int(back), ms, 3), "get_timings")
    return float(ms[0]), float(ms[1]), float(ms[2])


def device_info():
    cu, khz, mem = ctypes.c_int(), ctypes.c_int(), ctypes.c_size_t()
    arch = ctypes.create_string_buffer(64)
    check(lib.ssdk_device_info(ctypes.byref(cu), ctypes.byref(khz), ctypes.byref(mem), arch, 64), "device_info")
    return {"cu_count": cu.value, "clock_khz": khz.value, "hbm_bytes": mem.value, "arch": arch.value.decode()}
