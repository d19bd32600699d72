"""ctypes binding of libuemda_hip.so (the C ABI in include/uemda_hip.h).

The product path has NO fallback: if the shared library is missing or a kernel launch fails the
caller gets an exception.  Nothing here imports `oracle/`.
"""
import ctypes
import os
import subprocess
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int64, c_uint64, c_void_p

import torch  # noqa: F401  -- MUST precede dlopen: torch brings its own libamdhip64.so.7; loading ours first
#                              would put a second HIP runtime in the process that cannot see torch's device state

_HERE = os.path.dirname(os.path.abspath(__file__))
# UEM_LIB_PATH: another build of the same library (A/B runs of scripts/ against an older build on one device)
LIB_PATH = os.environ.get("UEM_LIB_PATH") or os.path.join(_HERE, "libuemda_hip.so")
_lib = None


class PrepJob(Structure):
    """mirror of `uem_prep_job`"""
    _fields_ = [("src", c_void_p), ("dst", c_void_p), ("kind", c_int), ("cout", c_int), ("cin", c_int), ("taps", c_int)]


class ConvShape(Structure):
    """mirror of `uem_conv_shape`"""
    _fields_ = [(n, c_int) for n in ("N", "H", "W", "Cin", "Ho", "Wo", "Cout", "KH", "KW", "stride", "pad",
                                     "dil", "x_ld", "y_ld")]


P = c_void_p          # device pointers travel as integers
F = c_float
I = c_int
L = c_int64

# name -> argtypes (return type is always int unless listed in _RESTYPE)
SIGNATURES = {
    "uem_version": [],
    "uem_last_error": [],
    "uem_conv2d_fwd": [P, P, P, P, P, P, POINTER(ConvShape), I, P],
    "uem_conv2d_fwd_stats": [P, P, P, P, P, POINTER(ConvShape), I, P, P],
    "uem_conv2d_dgrad_bnbwd": [P, P, P, POINTER(ConvShape), P, P, P, c_int, P],
    "uem_conv2d_dgrad_tail": [P, P, P, POINTER(ConvShape), P, P, P, P, P, P, c_int, P],
    "uem_conv2d_stem_fwd": [P, P, P, I, I, I, P],
    "uem_conv2d_wgrad": [P, P, P, P, P, POINTER(ConvShape), I, P],
    "uem_conv2d_stem_wgrad": [P, P, P, I, I, I, P],
    "uem_stem_conv_fwd": [P, P, P, I, I, I, P, P],
    "uem_stem_conv_wgrad_workspace_floats": [],
    "uem_stem_conv_wgrad": [P, P, P, P, I, I, I, P],
    "uem_stem_conv_wgrad_bf16": [P, P, P, P, I, I, I, P],
    "uem_weight_transpose": [P, P, I, I, I, I, P],
    "uem_stem_pack_weight": [P, P, P],
    "uem_stem_unpack_grad": [P, P, P],
    "uem_nchw3_to_nhwc4": [P, P, I, I, I, P],
    "uem_bias_grad": [P, P, I, I, I, P],
    "uem_aspp_gather_fwd": [P, P, P, P, I, I, I, I, I, I, POINTER(c_int), P],
    "uem_aspp_gather_bwd": [P, P, P, I, I, I, I, I, I, POINTER(c_int), P],
    "uem_aspp_pack": [POINTER(c_void_p), POINTER(c_void_p), P, P, I, I, I, I, I, P],
    "uem_aspp_unpack_grad": [P, P, POINTER(c_void_p), POINTER(c_void_p), I, I, I, I, P],
    "uem_bn_stats": [P, I, I, I, P, P, F, F, P, P, P, P, P, P, P, P],
    "uem_bn_workspace_floats": [I, I],
    "uem_bn_stats_from_tiles": [P, I, I, I, P, P, F, F, P, P, P, P, P, P, P],
    "uem_bn_eval_affine": [P, P, P, P, F, P, P, P, P, I, P],
    "uem_affine_act": [P, P, P, P, P, P, P, L, I, I, P, P],
    "uem_bn_bwd_reduce": [P, P, P, P, P, P, P, I, I, I, P, P, P, P, P, P],
    "uem_bn_bwd_from_tiles": [P, I, I, P, P, P, P, P],
    "uem_bn_bwd_apply": [P, P, P, P, P, P, P, P, P, I, I, I, P, P, P],
    "uem_bn_bwd_apply_pair": [P] * 14 + [I, I, P, P, P],
    "uem_bn_bwd_apply_pair_bf16": [P] * 14 + [I, I, P, P, P],
    "uem_affine_act_bwd": [P, P, P, P, P, L, I, I, P, P, P],
    "uem_maxpool3x3s2_fwd": [P, P, P, I, I, I, I, P],
    "uem_maxpool3x3s2_bwd": [P, P, P, I, I, I, I, P],
    "uem_maxpool3x3s2_affine_fwd": [P, P, P, P, P, I, I, I, I, P],
    "uem_bn_bwd_reduce_pool": [P, P, P, P, P, P, P, I, I, I, I, I, P, P, P, P, P, P],
    "uem_bn_bwd_apply_pool": [P, P, P, P, P, P, P, P, P, I, I, I, I, I, P, P],
    "uem_conv2d_stem_fwd_stats": [P, P, P, I, I, I, P, I, P],
    "uem_conv2d_stem_wgrad_prec": [P, P, P, I, I, I, I, P],
    "uem_instnorm_fwd": [P, P, P, P, I, I, I, F, P],
    "uem_instnorm_bwd": [P, P, P, P, I, I, I, P],
    "uem_adaptive_avgpool_fwd": [P, P, I, I, I, I, I, P],
    "uem_adaptive_avgpool_bwd": [P, P, I, I, I, I, I, P],
    "uem_ppm_feat_grad": [P, I, P, P, I, P, I, I, I, I, P],
    "uem_bilinear_up_fwd": [P, P, I, I, I, I, I, I, I, I, P, P, I, P],
    "uem_bilinear_up_bwd": [P, P, I, I, I, I, I, I, I, I, P],
    "uem_dropout2d": [P, P, P, I, I, I, F, c_uint64, P],
    "uem_add_inplace": [P, P, L, P],
    "uem_add_clear": [P, P, L, P],
    "uem_nhwc_to_nchw": [P, P, I, I, I, P],
    "uem_nchw_to_nhwc": [P, P, I, I, I, P],
    "uem_pearson_sim": [P, P, P, P, I, I, I, P],
    "uem_pearson_dist": [P, P, P, P, I, I, I, P],
    "uem_index_max": [P, L, P, P],
    "uem_scatter": [P, P, P, P, I, I, I, I, I, P],
    "uem_segment_max_planar": [P, P, P, I, I, I, I, I, P, P],
    "uem_label_refine": [P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, F, I, P],
    "uem_label_refine_select_workspace_bytes": [I, I, I],
    "uem_label_refine_select": [P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, F, I, F, F, L, P],
    "uem_label_refine_workspace_floats": [I, I, I, I, I],
    "uem_plane_max": [P, P, I, I, L, P],
    "uem_pseudo_select": [P, P, P, P, I, I, L, F, F, L, P],
    "uem_downscale_label": [P, P, I, I, I, I, I, L, F, P],
    "uem_superpixel_shrink": [P, P, I, I, I, I, I, P],
    "uem_proto_sums": [P, P, P, P, P, I, I, I, L, P],
    "uem_proto_ema": [P, P, P, I, I, F, P],
    "uem_ce_upsampled": [P, P, P, P, P, P, P, P, I, I, I, I, I, I, L, F, P],
    "uem_uvem_upsampled": [P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, F, F, F, L, F, P],
    "uem_loss_workspace_floats": [I, I, I, I],
    "uem_scale_by_scalar": [P, P, L, P, P],
    "uem_upsample_softmax_avg": [P, P, P, I, I, I, I, I, I, P],
    "uem_uvem_weight": [P, P, L, F, F, F, P],
    "uem_class_count": [P, L, I, L, P, P],
    "uem_class_weight_gather": [P, P, P, L, I, L, P],
    "uem_window_accumulate": [P, P, P, I, I, I, I, I, I, I, I, I, I, P],
    "uem_window_normalize": [P, P, I, I, I, I, P],
    "uem_scale": [P, L, F, P],
    "uem_argmax_confusion": [P, P, P, P, I, I, L, P],
    "uem_proto_mean": [P, P, P, I, I, P],
    "uem_pcl_loss": [P, P, P, P, P, P, I, I, I, F, L, P],
    "uem_pcl_workspace_floats": [I, I],
    "uem_coral_finish": [P, P, P, P, I, I, I, P, P, P, P, P],
    "uem_negate": [P, P, I, P],
    "uem_conv2d_stem_fwd_stats_bf16": [P, P, P, I, I, I, P, P],
    "uem_conv2d_stem_wgrad_bf16": [P, P, P, I, I, I, P],
    "uem_bn_bwd_reduce_pool_bf16": [P, P, P, P, P, P, P, I, I, I, I, I, P, P, P, P, P, P],
    "uem_bn_bwd_apply_pool_bf16": [P, P, P, P, P, P, P, P, P, I, I, I, I, I, P, P],
    "uem_maxpool3x3s2_affine_fwd_bf16": [P, P, P, P, P, I, I, I, I, P],
    "uem_instnorm_fwd_bf16": [P, P, P, P, I, I, I, F, P],
    "uem_instnorm_bwd_bf16": [P, P, P, P, I, I, I, P],
    "uem_conv2d_bf16": [P, P, P, POINTER(ConvShape), I, P, P],
    "uem_conv2d_dgrad_tail_bf16": [P, P, P, POINTER(ConvShape), P, P, P, P, P, P, I, P],
    "uem_conv2d_wgrad_bf16": [P, P, P, POINTER(ConvShape), P],
    "uem_affine_act_bf16": [P, P, P, P, P, P, P, L, I, I, P, P],
    "uem_bn_bwd_reduce_bf16": [P, P, P, P, P, P, P, I, I, I, P, P, P, P, P, P],
    "uem_bn_bwd_apply_bf16": [P, P, P, P, P, P, P, P, P, I, I, I, P, P, P],
    "uem_weight_transpose_bf16": [P, P, I, I, I, I, P],
    "uem_cast_f32_bf16": [P, P, L, P],
    "uem_cast_bf16_f32": [P, P, L, P],
    "uem_wino_filter": [P, P, I, I, I, I, P],
    "uem_wino_input": [P, P, P, I, P, I, I, I, I, I, I, I, P],
    "uem_wino_gemm": [P, P, P, I, I, I, I, I, P],
    "uem_wino_output": [P, P, I, I, I, I, I, I, P, P, P, P, P],
    "uem_wino_dy": [P, P, I, I, I, I, I, I, P, L, P],
    "uem_wino_wgrad_gemm": [P, P, P, I, I, I, I, P],
    "uem_wino_filter_grad": [P, P, I, I, I, P],
    "uem_weight_prep_blocks": [I, I, I, I],
    "uem_weight_prep": [P, P, I, I, P],
    "uem_comm_unique_id": [P],
    "uem_comm_init": [POINTER(c_void_p), P, I, I],
    "uem_allreduce_flat": [P, P, L, P],
    "uem_comm_destroy": [P],
    "uem_grad_sqnorm": [P, L, P, P, P],
    "uem_sgd_clip_step": [P, P, P, L, P, F, F, F, F, I, F, P, P],
}
_RESTYPE = {"uem_last_error": c_char_p, "uem_bn_workspace_floats": c_int64,
            "uem_label_refine_workspace_floats": c_int64, "uem_label_refine_select_workspace_bytes": c_int64, "uem_stem_conv_wgrad_workspace_floats": c_int64, "uem_pcl_workspace_floats": c_int64,
            "uem_loss_workspace_floats": c_int64}

# compile-time constants mirrored from the header
UEM_MAX_CLASSES = 16
UEM_PROTO_SPLIT = 256
UEM_NORM_BLOCKS = 1024
CONV_IN_AFFINE, CONV_IN_RELU, CONV_ACCUMULATE, CONV_TRANSPOSED = 1, 2, 4, 8
CONV_PREC_BF16 = 32
PREP_TRANSPOSE, PREP_WINO2, PREP_WINO2_T, PREP_WINO4, PREP_WINO4_T, PREP_STEM_PACK, PREP_TRANSPOSE_BF16 = range(7)


class UemError(RuntimeError):
    pass


def build(verbose=False):
    """Compile every HIP source for gfx950 into uemda_amd/libuemda_hip.so (hipcc cross-compiles)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j", str(min(8, os.cpu_count() or 1))]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout[-4000:])
        print(res.stderr[-4000:])
    if res.returncode != 0:
        raise UemError("hipcc build of libuemda_hip.so failed")
    return LIB_PATH


def load():
    """dlopen the library and type every entry point.  Raises (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise UemError(f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(there is no CPU / PyTorch fallback for the HIP path)")
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)               # AttributeError if the symbol is not exported
        fn.argtypes = argtypes
        fn.restype = _RESTYPE.get(name, c_int)
    _lib = lib
    return lib


ERR_UNSUPPORTED = -2          # UEM_ERR_UNSUPPORTED: the entry point does not take this configuration and launched nothing


def try_call(name, *args):
    """Like call(), for entry points that may decline a configuration: False on UEM_ERR_UNSUPPORTED (nothing was launched; the caller
    takes its other path), True on success, UemError on anything else."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc == ERR_UNSUPPORTED:
        return False
    if rc != 0:
        msg = lib.uem_last_error()
        raise UemError(f"{name} failed (code {rc}): {msg.decode() if msg else '?'}")
    return True


def call(name, *args):
    """Invoke an entry point; raise UemError with the library's message on a non-zero code."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.uem_last_error()
        raise UemError(f"{name} failed (code {rc}): {msg.decode() if msg else '?'}")
    return rc
