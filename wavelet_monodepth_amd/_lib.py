"""ctypes binding of libwmd_hip.so (C ABI: include/wmd.h).

There is deliberately no fallback: if the library is missing or a call fails, this raises.
PyTorch is used only as the owner of device memory and of the current HIP stream.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("WMD_LIB_PATH") or os.path.join(_HERE, "libwmd_hip.so")   # WMD_LIB_PATH: A/B builds (development)

PAD = {"zero": 0, "constant": 0, "reflect": 1, "reflection": 1, "replicate": 2}
ACT = {"none": 0, None: 0, "elu": 1, "leaky": 2, "sigmoid": 3}


class WmdError(RuntimeError):
    pass


class ConvArgs(C.Structure):
    _fields_ = [("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("C1", C.c_int), ("up1", C.c_int), ("C2", C.c_int),
                ("Cout", C.c_int), ("ksize", C.c_int), ("pad_mode", C.c_int), ("act", C.c_int), ("slope", C.c_float),
                ("x1", C.c_void_p), ("x2", C.c_void_p), ("wp", C.c_void_p), ("bias", C.c_void_p), ("y", C.c_void_p),
                ("workspace", C.c_void_p), ("workspace_floats", C.c_size_t), ("tune_cfg", C.c_int), ("tune_ksplit", C.c_int),
                ("wp_wino", C.c_void_p), ("gate", C.c_void_p), ("gate_act", C.c_int), ("gate_slope", C.c_float),
                ("in_mask", C.c_void_p), ("out_mask", C.c_void_p), ("in_mask_2x2", C.c_int),
                ("out_tiles", C.c_void_p), ("out_tile_count", C.c_void_p), ("out_tile_h", C.c_int), ("out_tile_w", C.c_int)]


class ConvDgradArgs(C.Structure):
    _fields_ = [("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("C1", C.c_int), ("up1", C.c_int), ("C2", C.c_int),
                ("Cout", C.c_int), ("ksize", C.c_int), ("pad_mode", C.c_int),
                ("dz", C.c_void_p), ("wp_dgrad", C.c_void_p), ("dx1", C.c_void_p), ("dx2", C.c_void_p),
                ("workspace", C.c_void_p), ("workspace_floats", C.c_size_t), ("tune_cfg", C.c_int), ("tune_ksplit", C.c_int),
                ("wp_dgrad_wino", C.c_void_p), ("x1_fwd", C.c_void_p), ("x1_act", C.c_int), ("x1_slope", C.c_float)]


class ConvWgradArgs(C.Structure):
    _fields_ = [("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("C1", C.c_int), ("up1", C.c_int), ("C2", C.c_int),
                ("Cout", C.c_int), ("ksize", C.c_int), ("pad_mode", C.c_int),
                ("x1", C.c_void_p), ("x2", C.c_void_p), ("dz", C.c_void_p), ("dw", C.c_void_p), ("dbias", C.c_void_p),
                ("workspace", C.c_void_p), ("workspace_floats", C.c_size_t), ("tune_cfg", C.c_int), ("tune_nsplit", C.c_int)]


class HeadArgs(C.Structure):
    _fields_ = [("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("C", C.c_int), ("Cout", C.c_int),
                ("pad_mode", C.c_int), ("mode", C.c_int), ("scale", C.c_float),
                ("xp", C.c_void_p), ("wgt_p", C.c_void_p), ("bias_p", C.c_void_p),
                ("xn", C.c_void_p), ("wgt_n", C.c_void_p), ("bias_n", C.c_void_p),
                ("y", C.c_void_p), ("sig_p", C.c_void_p), ("sig_n", C.c_void_p),
                ("xp_bstride", C.c_size_t), ("xn_bstride", C.c_size_t),
                ("workspace", C.c_void_p), ("workspace_floats", C.c_size_t)]


class HeadFusedArgs(C.Structure):
    _fields_ = [("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("C", C.c_int), ("slope", C.c_float),
                ("x", C.c_void_p), ("wp1", C.c_void_p), ("bias1", C.c_void_p), ("wp2", C.c_void_p), ("t", C.c_void_p),
                ("chain", C.c_int), ("t_planes", C.c_int), ("run_mask", C.c_void_p), ("ll_wp1", C.c_void_p),
                ("ll_bias1", C.c_void_p), ("ll_wp2", C.c_void_p), ("mid_out", C.c_void_p), ("mid_ct", C.c_int),
                ("mid_off_p", C.c_int), ("mid_off_n", C.c_int), ("mid_off_ll", C.c_int)]


class HeadLevelArgs(C.Structure):
    _fields_ = [("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("C", C.c_int), ("pad_mode", C.c_int),
                ("slope", C.c_float), ("scale", C.c_float),
                ("x", C.c_void_p), ("wp1", C.c_void_p), ("bias1", C.c_void_p), ("wp2", C.c_void_p),
                ("bias_p", C.c_void_p), ("bias_n", C.c_void_p), ("yh", C.c_void_p),
                ("yl", C.c_void_p), ("out", C.c_void_p), ("disp", C.c_void_p), ("disp_scale", C.c_float),
                ("clamp01", C.c_int), ("yh_mask", C.c_void_p), ("mid_out", C.c_void_p), ("mid_ct", C.c_int),
                ("mid_off_p", C.c_int), ("mid_off_n", C.c_int), ("sig_p", C.c_void_p), ("sig_n", C.c_void_p)]


class HeadBwdHead(C.Structure):
    _fields_ = [("row0", C.c_int), ("nrows", C.c_int), ("ch0", C.c_int), ("nch", C.c_int), ("w3", C.c_void_p),
                ("dw3", C.c_void_p), ("db3", C.c_void_p)]


class Head3x3BwdArgs(C.Structure):
    _fields_ = [("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("Ct", C.c_int), ("n_out", C.c_int), ("pad_mode", C.c_int),
                ("act", C.c_int), ("slope", C.c_float), ("dy3", C.c_void_p), ("mid", C.c_void_p), ("dzmid", C.c_void_p),
                ("n_heads", C.c_int), ("head", HeadBwdHead * 3), ("workspace", C.c_void_p), ("workspace_floats", C.c_size_t)]


class Head1x1BwdArgs(C.Structure):
    _fields_ = [("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("C", C.c_int), ("Ct", C.c_int), ("x_act", C.c_int),
                ("x_slope", C.c_float), ("dz", C.c_void_p), ("x", C.c_void_p), ("w1", C.c_void_p), ("dx", C.c_void_p),
                ("dw1", C.c_void_p), ("db1", C.c_void_p), ("workspace", C.c_void_p), ("workspace_floats", C.c_size_t)]


class EvalKittiArgs(C.Structure):
    _fields_ = [("B", C.c_int), ("h", C.c_int), ("w", C.c_int), ("H", C.c_int), ("W", C.c_int),
                ("min_depth", C.c_float), ("max_depth", C.c_float), ("mask_mode", C.c_int), ("pred_scale", C.c_float),
                ("median_scaling", C.c_int), ("pred_disp", C.c_void_p), ("gt_depth", C.c_void_p), ("out", C.c_void_p),
                ("workspace", C.c_void_p), ("workspace_floats", C.c_size_t)]


class WarpArgs(C.Structure):
    _fields_ = [("B", C.c_int), ("C", C.c_int), ("H", C.c_int), ("W", C.c_int), ("Hs", C.c_int), ("Ws", C.c_int),
                ("eps", C.c_float), ("src", C.c_void_p), ("depth", C.c_void_p), ("K", C.c_void_p), ("inv_K", C.c_void_p),
                ("T", C.c_void_p)]


class DwConvArgs(C.Structure):
    _fields_ = [("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("C1", C.c_int), ("up1", C.c_int), ("C2", C.c_int),
                ("pad_mode", C.c_int), ("x1", C.c_void_p), ("x2", C.c_void_p), ("w", C.c_void_p)]


class HeadShiftsumArgs(C.Structure):
    _fields_ = [("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("pad_mode", C.c_int), ("scale", C.c_float),
                ("t", C.c_void_p), ("bias_p", C.c_void_p), ("bias_n", C.c_void_p), ("yh", C.c_void_p),
                ("yl", C.c_void_p), ("out", C.c_void_p), ("disp", C.c_void_p), ("disp_scale", C.c_float),
                ("clamp01", C.c_int), ("bias_ll", C.c_void_p), ("scale_ll", C.c_float), ("yl_out", C.c_void_p),
                ("yh_mask", C.c_void_p), ("range_keys", C.c_void_p), ("sig_p", C.c_void_p), ("sig_n", C.c_void_p),
                ("sig_ll", C.c_void_p)]


class LevelSpec(C.Structure):
    _fields_ = [("up", C.c_int), ("radius", C.c_int), ("out", C.c_void_p), ("count", C.c_int), ("tile_h", C.c_int),
                ("tile_w", C.c_int), ("tile_list", C.c_void_p), ("tile_count", C.c_void_p), ("and_mask", C.c_void_p)]


class MaskLevelArgs(C.Structure):
    _fields_ = [("B", C.c_int), ("h", C.c_int), ("w", C.c_int), ("yl", C.c_void_p), ("n_yl", C.c_size_t), ("yh", C.c_void_p),
                ("thresh_ratio", C.c_float), ("mask0", C.c_void_p), ("minmax", C.c_void_p), ("range_keys", C.c_void_p),
                ("specs", C.POINTER(LevelSpec)), ("n", C.c_int), ("scratch", C.c_void_p), ("counts", C.c_void_p),
                ("ring_slots", C.c_int), ("slot_ints", C.c_int), ("counts_off", C.c_int), ("ncounts", C.c_int),
                ("advance", C.c_int)]


class DilateSpec(C.Structure):
    _fields_ = [("up", C.c_int), ("radius", C.c_int), ("out", C.c_void_p), ("nnz", C.c_void_p), ("nnz_stride", C.c_int)]


class PackItem(C.Structure):
    _fields_ = [("w", C.c_void_p), ("Cout", C.c_int), ("Cin", C.c_int), ("ksize", C.c_int), ("fwd", C.c_void_p),
                ("dgrad", C.c_void_p), ("wino_fwd", C.c_void_p), ("wino_dgrad", C.c_void_p)]


class CompactSpec(C.Structure):
    _fields_ = [("mask", C.c_void_p), ("npix", C.c_int), ("coords", C.c_void_p), ("nnz", C.c_void_p)]


class SparseConvArgs(C.Structure):
    _fields_ = [("H", C.c_int), ("W", C.c_int), ("C1", C.c_int), ("up1", C.c_int), ("C1tot", C.c_int), ("c1_off", C.c_int),
                ("C2", C.c_int), ("Cout", C.c_int), ("ksize", C.c_int), ("pad_mode", C.c_int), ("act", C.c_int),
                ("slope", C.c_float), ("x1", C.c_void_p), ("x2", C.c_void_p), ("in_mask", C.c_void_p),
                ("out_coords", C.c_void_p), ("out_nnz", C.c_void_p), ("max_out", C.c_int),
                ("wp", C.c_void_p), ("bias", C.c_void_p), ("wp2", C.c_void_p), ("bias2", C.c_void_p),
                ("c1_off2", C.c_int), ("out_scale", C.c_float), ("y", C.c_void_p), ("split_waves", C.c_int),
                ("B", C.c_int), ("nnz_stride", C.c_int)]


_lib = None

# name -> (restype, argtypes); the single source for the "library exports what wmd.h declares" test
SIGNATURES = {
    "wmd_version": (C.c_int, []),
    "wmd_last_error": (C.c_char_p, []),
    "wmd_status_string": (C.c_char_p, [C.c_int]),
    "wmd_profile_begin": (C.c_int, []),
    "wmd_profile_end": (C.c_long, [C.c_char_p, C.c_size_t]),
    "wmd_idwt_haar_fwd": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 3 + [C.c_float, C.c_int, C.c_void_p]),
    "wmd_idwt_haar_bwd": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 3 + [C.c_float, C.c_int, C.c_void_p]),
    "wmd_dwt_haar_fwd": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 3 + [C.c_void_p]),
    "wmd_dwt_haar_reflect_fwd": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 3 + [C.c_void_p]),
    "wmd_conv_packed_weight_floats": (C.c_size_t, [C.c_int] * 3),
    "wmd_conv_pack_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "wmd_conv_pack_weights_dgrad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "wmd_conv_packed_weight_floats_wino": (C.c_size_t, [C.c_int, C.c_int]),
    "wmd_conv_pack_weights_wino": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "wmd_conv_pack_many": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "wmd_conv_fwd": (C.c_int, [C.POINTER(ConvArgs), C.c_void_p]),
    "wmd_conv_fwd_workspace_floats": (C.c_size_t, [C.POINTER(ConvArgs)]),
    "wmd_conv_num_configs": (C.c_int, []),
    "wmd_conv_config_name": (C.c_char_p, [C.c_int]),
    "wmd_act_bwd": (C.c_int, [C.c_void_p] * 3 + [C.c_size_t, C.c_int, C.c_float, C.c_void_p]),
    "wmd_conv_dgrad_workspace_floats": (C.c_size_t, [C.POINTER(ConvDgradArgs)]),
    "wmd_conv_dgrad": (C.c_int, [C.POINTER(ConvDgradArgs), C.c_void_p]),
    "wmd_conv_wgrad_workspace_floats": (C.c_size_t, [C.POINTER(ConvWgradArgs)]),
    "wmd_conv_wgrad": (C.c_int, [C.POINTER(ConvWgradArgs), C.c_void_p]),
    "wmd_conv_wgrad_num_configs": (C.c_int, []),
    "wmd_conv_wgrad_config_name": (C.c_char_p, [C.c_int]),
    "wmd_head3x3_fwd": (C.c_int, [C.POINTER(HeadArgs), C.c_void_p]),
    "wmd_head3x3_workspace_floats": (C.c_size_t, [C.POINTER(HeadArgs)]),
    "wmd_head_fused_fwd": (C.c_int, [C.POINTER(HeadFusedArgs), C.c_void_p]),
    "wmd_head_fused_multi_fwd": (C.c_int, [C.POINTER(HeadFusedArgs), C.c_int, C.c_void_p]),
    "wmd_head_shiftsum_fwd": (C.c_int, [C.POINTER(HeadShiftsumArgs), C.c_void_p]),
    "wmd_head_shiftsum_chain_fwd": (C.c_int, [C.POINTER(HeadShiftsumArgs), C.c_int, C.c_void_p]),
    "wmd_head_level_supported": (C.c_int, [C.c_int]),
    "wmd_head_level_fwd": (C.c_int, [C.POINTER(HeadLevelArgs), C.c_void_p]),
    "wmd_head_level_pyramid_supported": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "wmd_head_level_pyramid_fwd": (C.c_int, [C.POINTER(HeadLevelArgs), C.POINTER(HeadShiftsumArgs), C.c_int, C.c_void_p]),
    "wmd_head3x3_bwd_workspace_floats": (C.c_size_t, [C.POINTER(Head3x3BwdArgs)]),
    "wmd_head3x3_bwd": (C.c_int, [C.POINTER(Head3x3BwdArgs), C.c_void_p]),
    "wmd_head1x1_bwd_workspace_floats": (C.c_size_t, [C.POINTER(Head1x1BwdArgs)]),
    "wmd_head1x1_bwd": (C.c_int, [C.POINTER(Head1x1BwdArgs), C.c_void_p]),
    "wmd_head_bwd": (C.c_int, [C.POINTER(Head3x3BwdArgs), C.POINTER(Head1x1BwdArgs), C.c_void_p]),
    "wmd_minmax": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "wmd_mask_threshold": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "wmd_mask_dilate_multi": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(DilateSpec), C.c_int, C.c_void_p]),
    "wmd_mask_level": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_float, C.c_int, C.c_int, C.POINTER(DilateSpec), C.c_int,
                       C.c_void_p]),
    "wmd_mask_compact_multi": (C.c_int, [C.POINTER(CompactSpec), C.c_int, C.c_void_p]),
    "wmd_mask_dilate_multi_b": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(DilateSpec), C.c_int, C.c_void_p]),
    "wmd_mask_level_b": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_int, C.POINTER(DilateSpec), C.c_int,
                         C.c_void_p, C.c_void_p]),
    "wmd_mask_compact_multi_b": (C.c_int, [C.POINTER(CompactSpec), C.c_int, C.c_int, C.c_void_p]),
    "wmd_mask_level_scratch_ints": (C.c_size_t, [C.c_int]),
    "wmd_mask_level_lists": (C.c_int, [C.POINTER(MaskLevelArgs), C.c_void_p]),
    "wmd_conv_list_tile_supported": (C.c_int, [C.c_int, C.c_int]),
    "wmd_sparse_conv": (C.c_int, [C.POINTER(SparseConvArgs), C.c_void_p]),
    "wmd_upsample_bilinear_fwd": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 6 + [C.c_float, C.c_float, C.c_void_p]),
    "wmd_upsample_bilinear_bwd": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 6 + [C.c_float, C.c_float, C.c_void_p]),
    "wmd_eval_workspace_floats": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "wmd_eval_kitti": (C.c_int, [C.POINTER(EvalKittiArgs), C.c_void_p]),
    "wmd_eval_errors": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p]),
    "wmd_flip_postprocess": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "wmd_ssim_fwd": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 5 + [C.c_float, C.c_float, C.c_void_p]),
    "wmd_ssim_bwd_workspace_floats": (C.c_size_t, [C.c_int] * 4),
    "wmd_ssim_bwd": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 5 + [C.c_float, C.c_float, C.c_void_p, C.c_size_t, C.c_void_p]),
    "wmd_warp_fwd": (C.c_int, [C.POINTER(WarpArgs), C.c_void_p, C.c_void_p]),
    "wmd_warp_bwd_workspace_floats": (C.c_size_t, [C.POINTER(WarpArgs)]),
    "wmd_warp_bwd": (C.c_int, [C.POINTER(WarpArgs), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "wmd_smooth_workspace_floats": (C.c_size_t, [C.c_int] * 3),
    "wmd_smooth_fwd": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 4 + [C.c_float, C.c_void_p, C.c_size_t, C.c_void_p]),
    "wmd_smooth_bwd": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_float, C.c_void_p]),
    "wmd_dwconv3x3_fwd": (C.c_int, [C.POINTER(DwConvArgs), C.c_void_p, C.c_void_p]),
    "wmd_dwconv3x3_bwd_workspace_floats": (C.c_size_t, [C.POINTER(DwConvArgs)]),
    "wmd_dwconv3x3_bwd": (C.c_int, [C.POINTER(DwConvArgs)] + [C.c_void_p] * 6 + [C.c_size_t, C.c_void_p]),
    "wmd_comm_unique_id": (C.c_int, [C.c_void_p]),
    "wmd_comm_init": (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_int]),
    "wmd_comm_allreduce": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_float, C.c_void_p]),
    "wmd_comm_broadcast": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "wmd_comm_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "wmd_comm_destroy": (C.c_int, [C.c_void_p]),
}


def lib():
    """Load libwmd_hip.so once. Raises WmdError when it has not been built (no silent fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise WmdError("%s not found: build it with `python -m wavelet_monodepth_amd.build` "
                           "(or __graft_entry__.build()); there is no CPU fallback" % LIB_PATH)
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(status, what=""):
    if status != 0:
        l = lib()
        raise WmdError("%s failed: %s (%s)" % (what or "libwmd_hip call", l.wmd_status_string(status).decode(),
                                               l.wmd_last_error().decode()))


def ptr(t):
    """Device pointer of a tensor (None -> NULL). The tensor must be fp32/int and contiguous."""
    if t is None:
        return None
    if not t.is_cuda:   # last line of defence: a host pointer must never reach a kernel (there is no CPU path)
        raise WmdError("libwmd_hip expects device tensors, got a %s tensor" % t.device)
    assert t.is_contiguous(), "libwmd_hip expects contiguous tensors"
    return t.data_ptr()


def current_stream():
    import torch

    return torch.cuda.current_stream().cuda_stream


def profile_begin():
    check(lib().wmd_profile_begin(), "wmd_profile_begin")


def profile_end():
    """-> list of {"kernel", "calls", "ms", "flops", "bytes"} aggregated per kernel name."""
    import json

    buf = C.create_string_buffer(1 << 16)
    n = lib().wmd_profile_end(buf, len(buf))
    if n < 0:
        check(int(n), "wmd_profile_end")
    return json.loads(buf.value.decode())
