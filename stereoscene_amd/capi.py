"""ctypes binding of libssbev_hip.so (include/ssbev.h).  PyTorch only supplies device memory and
streams; every argument that crosses this boundary is a raw pointer, an int or a POD struct.

There is NO fallback: if the library is missing or a tensor is not on the GPU the call raises.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_lib", "libssbev_hip.so")

OK, EINVAL, EWORKSPACE, ELAUNCH = 0, -1, -2, -3
_ERR = {EINVAL: "SSBEV_EINVAL (bad dims / null pointer / unsupported configuration)",
        EWORKSPACE: "SSBEV_EWORKSPACE (workspace too small)",
        ELAUNCH: "SSBEV_ELAUNCH (HIP launch failure)"}


class SsbevError(RuntimeError):
    pass


class PoolDims(C.Structure):
    _fields_ = [("B", C.c_int), ("P", C.c_int), ("C", C.c_int), ("nx", C.c_int), ("ny", C.c_int), ("nz", C.c_int),
                ("origin", C.c_float * 3), ("dx", C.c_float * 3)]


class LiftDims(C.Structure):
    _fields_ = [("N", C.c_int), ("D", C.c_int), ("HW", C.c_int)]


class GwcDims(C.Structure):
    _fields_ = [("B", C.c_int), ("C", C.c_int), ("G", C.c_int), ("D", C.c_int), ("H", C.c_int), ("W", C.c_int),
                ("down", C.c_float), ("align_corners", C.c_int)]


class ConvDims(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "B", "Cin", "Cout", "Di", "Hi", "Wi", "Do", "Ho", "Wo", "kd", "kh", "kw", "sd", "sh", "sw",
        "pd", "ph", "pw", "dd", "dh", "dw", "transposed", "relu", "accumulate", "tile_hint", "precision")]


class GeomDims(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("B", "N", "D", "H", "W")]


class NormDims(C.Structure):
    _fields_ = [("B", C.c_int), ("C", C.c_int), ("G", C.c_int), ("S", C.c_int64), ("eps", C.c_float),
                ("relu", C.c_int), ("stats_given", C.c_int), ("pre_act", C.c_int), ("ld_y", C.c_int64), ("ld_gy", C.c_int64),
                ("io_dtype", C.c_int)]


class Norm2Dims(C.Structure):
    _fields_ = [("B", C.c_int), ("C", C.c_int), ("Ga", C.c_int), ("Gb", C.c_int), ("S", C.c_int64), ("eps_a", C.c_float),
                ("eps_b", C.c_float), ("relu", C.c_int), ("a_batch", C.c_int), ("b_batch", C.c_int), ("io_dtype", C.c_int)]


class Gemm16Dims(C.Structure):
    _fields_ = [("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("batch", C.c_int), ("out_fp32", C.c_int)]


class NormExt(C.Structure):
    _fields_ = [("running_mean", C.c_void_p), ("running_var", C.c_void_p), ("momentum", C.c_float),
                ("n", C.c_int64)]


class Norm2Ext(C.Structure):
    _fields_ = [("running_mean_a", C.c_void_p), ("running_var_a", C.c_void_p), ("momentum_a", C.c_float),
                ("running_mean_b", C.c_void_p), ("running_var_b", C.c_void_p), ("momentum_b", C.c_float)]


class DcnDims(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("B", "C", "H", "W", "G", "k", "pad", "dil")]


class DwDims(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("B", "C", "Hi", "Wi", "Ho", "Wo", "k", "stride", "pad_t", "pad_l")]


class WinoDims(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("B", "D", "H", "W", "C")]


class UpsampleDims(C.Structure):
    _fields_ = [("B", C.c_int), ("D", C.c_int), ("H", C.c_int), ("W", C.c_int), ("C", C.c_int)]


class GemmDims(C.Structure):
    _fields_ = [("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("batch", C.c_int),
                ("lda", C.c_int64), ("ldb", C.c_int64), ("ldc", C.c_int64), ("sa", C.c_int64), ("sb", C.c_int64), ("sc", C.c_int64),
                ("relu", C.c_int),
                ("d2s_D", C.c_int), ("d2s_H", C.c_int), ("d2s_W", C.c_int), ("d2s_kd", C.c_int), ("d2s_kh", C.c_int),
                ("d2s_kw", C.c_int), ("d2s_Co", C.c_int), ("d2s_rowoff", C.c_void_p),
                ("ep_mul", C.c_void_p), ("ep_rowsub", C.c_void_p)]


class AdamWCfg(C.Structure):
    _fields_ = [("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("weight_decay", C.c_float), ("max_grad_norm", C.c_float), ("step", C.c_int)]


_P = C.c_void_p
# name -> (restype, argtypes); this table is checked against include/ssbev.h by tests/test_capi_symbols.py
SIGNATURES = {
    "ssbev_version": (C.c_int, []),
    "ssbev_build_arch": (C.c_char_p, []),
    "ssbev_env_refresh": (None, []),
    "ssbev_voxel_index": (C.c_int, [_P, _P, _P, C.POINTER(PoolDims), _P]),
    "ssbev_coords_to_vox": (C.c_int, [_P, C.c_int, _P, C.POINTER(PoolDims), _P]),
    "ssbev_pool_prepare_workspace": (C.c_size_t, [C.c_int, C.POINTER(PoolDims)]),
    "ssbev_pool_prepare": (C.c_int, [_P, C.c_int, _P, _P, C.POINTER(PoolDims), _P, C.c_size_t, _P]),
    "ssbev_pool_long_list_elems": (C.c_size_t, [C.c_int]),
    "ssbev_pool_prepare2": (C.c_int, [_P, C.c_int, _P, _P, _P, C.POINTER(PoolDims), _P, C.c_size_t, _P]),
    "ssbev_bev_pool_fwd": (C.c_int, [_P, _P, _P, _P, C.POINTER(PoolDims), _P]),
    "ssbev_bev_pool_bwd": (C.c_int, [_P, _P, C.c_int, _P, C.POINTER(PoolDims), _P]),
    "ssbev_lift_splat_fwd": (C.c_int, [_P, _P, _P, _P, _P, C.POINTER(PoolDims), C.POINTER(LiftDims), _P]),
    "ssbev_lift_splat_fwd2": (C.c_int, [_P, _P, _P, _P, _P, _P, C.POINTER(PoolDims), C.POINTER(LiftDims), _P]),
    "ssbev_lift_splat_bwd": (C.c_int, [_P, _P, _P, _P, _P, _P, C.POINTER(PoolDims), C.POINTER(LiftDims), _P]),
    "ssbev_gwc_warp_fwd": (C.c_int, [_P, _P, _P, _P, C.POINTER(GwcDims), _P]),
    "ssbev_gwc_warp_bwd": (C.c_int, [_P, _P, _P, _P, _P, _P, C.POINTER(GwcDims), _P]),
    "ssbev_gwc_warp_bwd_workspace": (C.c_size_t, [C.POINTER(GwcDims)]),
    "ssbev_gwc_warp_bwd_fused": (C.c_int, [_P, _P, _P, _P, _P, _P, C.POINTER(GwcDims), _P, C.c_size_t, _P]),
    "ssbev_conv_packed_weight_elems": (C.c_size_t, [C.POINTER(ConvDims)]),
    "ssbev_conv_kernel_class": (C.c_int, [C.POINTER(ConvDims), C.c_int]),
    "ssbev_frustum_geometry": (C.c_int, [_P] * 9 + [C.POINTER(GeomDims), _P]),
    "ssbev_conv_pack_weight": (C.c_int, [_P, _P, C.POINTER(ConvDims), C.c_int, _P]),
    "ssbev_conv_thin_workspace": (C.c_size_t, [C.POINTER(ConvDims), C.c_int]),
    "ssbev_conv_thin_packed_elems": (C.c_size_t, [C.POINTER(ConvDims), C.c_int]),
    "ssbev_conv_thin_pack": (C.c_int, [_P, _P, C.POINTER(ConvDims), C.c_int, _P]),
    "ssbev_conv_thin_run": (C.c_int, [_P, _P, _P, _P, C.POINTER(ConvDims), C.c_int, _P, C.c_size_t, _P]),
    "ssbev_conv_fwd": (C.c_int, [_P, _P, _P, _P, C.POINTER(ConvDims), _P]),
    "ssbev_conv_bwd_data": (C.c_int, [_P, _P, _P, C.POINTER(ConvDims), _P]),
    "ssbev_conv_fwd_bf16": (C.c_int, [_P, _P, _P, _P, C.POINTER(ConvDims), _P]),
    "ssbev_conv_bwd_data_bf16": (C.c_int, [_P, _P, _P, C.POINTER(ConvDims), _P]),
    "ssbev_conv_bwd_weight_bf16": (C.c_int, [_P, _P, _P, C.POINTER(ConvDims), _P, C.c_size_t, _P]),
    "ssbev_conv_bwd_weight_workspace": (C.c_size_t, [C.POINTER(ConvDims)]),
    "ssbev_conv_bwd_weight": (C.c_int, [_P, _P, _P, C.POINTER(ConvDims), _P, C.c_size_t, _P]),
    "ssbev_gemm16_packed_elems": (C.c_size_t, [C.POINTER(Gemm16Dims)]),
    "ssbev_gemm16_pack": (C.c_int, [_P, _P, C.POINTER(Gemm16Dims), _P]),
    "ssbev_gemm16_nn": (C.c_int, [_P, _P, _P, C.POINTER(Gemm16Dims), _P]),
    "ssbev_gemm16_tn_workspace": (C.c_size_t, [C.POINTER(Gemm16Dims)]),
    "ssbev_gemm16_tn": (C.c_int, [_P, _P, _P, C.POINTER(Gemm16Dims), _P, C.c_size_t, _P]),
    "ssbev_groupnorm_workspace": (C.c_size_t, [C.POINTER(NormDims)]),
    "ssbev_groupnorm_fwd": (C.c_int, [_P] * 7 + [C.POINTER(NormDims), _P, C.c_size_t, _P]),
    "ssbev_groupnorm_bwd": (C.c_int, [_P] * 10 + [C.POINTER(NormDims), _P, C.c_size_t, _P]),
    "ssbev_groupnorm_mask_words": (C.c_size_t, [C.POINTER(NormDims)]),
    "ssbev_groupnorm_fwd_mask": (C.c_int, [_P] * 8 + [C.POINTER(NormDims), _P, C.c_size_t, _P]),
    "ssbev_groupnorm_bwd_mask": (C.c_int, [_P] * 10 + [C.POINTER(NormDims), _P, C.c_size_t, _P]),
    "ssbev_groupnorm_fwd_ext": (C.c_int, [_P] * 8 + [C.POINTER(NormDims), C.POINTER(NormExt), _P, C.c_size_t, _P]),
    "ssbev_groupnorm_bwd_ext": (C.c_int, [_P] * 11 + [C.POINTER(NormDims), C.POINTER(NormExt), _P, C.c_size_t, _P]),
    "ssbev_groupnorm2_fwd_ext": (C.c_int, [_P] * 12 + [C.POINTER(Norm2Dims), C.POINTER(Norm2Ext), _P, C.c_size_t, _P]),
    "ssbev_groupnorm2_bwd_ext": (C.c_int, [_P] * 16 + [C.POINTER(Norm2Dims), C.POINTER(Norm2Ext), _P, C.c_size_t, _P]),
    "ssbev_groupnorm2_workspace": (C.c_size_t, [C.POINTER(Norm2Dims)]),
    "ssbev_groupnorm2_fwd": (C.c_int, [_P] * 12 + [C.POINTER(Norm2Dims), _P, C.c_size_t, _P]),
    "ssbev_groupnorm2_bwd": (C.c_int, [_P] * 16 + [C.POINTER(Norm2Dims), _P, C.c_size_t, _P]),
    "ssbev_trilinear2x_fwd": (C.c_int, [_P, _P, C.POINTER(UpsampleDims), _P]),
    "ssbev_trilinear2x_bwd": (C.c_int, [_P, _P, C.POINTER(UpsampleDims), _P]),
    "ssbev_dcn_im2col": (C.c_int, [_P, _P, _P, C.POINTER(DcnDims), _P]),
    "ssbev_dcn_col2im_workspace": (C.c_size_t, [C.POINTER(DcnDims)]),
    "ssbev_dcn_col2im": (C.c_int, [_P, _P, _P, _P, _P, C.POINTER(DcnDims), _P, C.c_size_t, _P]),
    "ssbev_bn_update_running": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_float, C.c_float, C.c_int64, _P]),
    "ssbev_resize_pil_u8": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, _P, _P, C.c_int, _P, _P, C.c_int, C.c_int, _P]),
    "ssbev_crop_normalize_u8": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, _P]),
    "ssbev_depth_bce_workspace": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "ssbev_depth_bce_fwd": (C.c_int, [_P, _P, _P] + [C.c_int] * 5 + [C.c_float] * 3 + [_P, C.c_size_t, _P]),
    "ssbev_depth_bce_bwd": (C.c_int, [_P, _P, _P, _P] + [C.c_int] * 5 + [C.c_float] * 3 + [_P, _P]),
    "ssbev_lidar_depth_workspace": (C.c_size_t, [C.c_int, C.c_int]),
    "ssbev_lidar_depth_map": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int, _P, C.c_size_t, _P]),
    "ssbev_dwconv2d_fwd": (C.c_int, [_P, _P, _P, C.POINTER(DwDims), _P]),
    "ssbev_dwconv2d_bwd_data": (C.c_int, [_P, _P, _P, C.POINTER(DwDims), _P]),
    "ssbev_dwconv2d_bwd_weight_workspace": (C.c_size_t, [C.POINTER(DwDims)]),
    "ssbev_dwconv2d_bwd_weight": (C.c_int, [_P, _P, _P, C.POINTER(DwDims), _P, C.c_size_t, _P]),
    "ssbev_swish_fwd": (C.c_int, [_P, _P, C.c_int64, _P]),
    "ssbev_swish_bwd": (C.c_int, [_P, _P, _P, C.c_int64, _P]),
    "ssbev_chan_sum_workspace": (C.c_size_t, [C.c_int, C.c_int64, C.c_int]),
    "ssbev_chan_sum": (C.c_int, [_P, _P, _P, C.c_int, C.c_int64, C.c_int, C.c_float, _P, C.c_size_t, _P]),
    "ssbev_chan_scale": (C.c_int, [_P, _P, _P, C.c_int, C.c_int64, C.c_int, C.c_float, _P]),
    "ssbev_wino_input_transform": (C.c_int, [_P, _P, C.POINTER(WinoDims), _P]),
    "ssbev_wino_output_transform": (C.c_int, [_P, _P, C.POINTER(WinoDims), _P]),
    "ssbev_wino_output_adjoint": (C.c_int, [_P, _P, C.POINTER(WinoDims), _P]),
    "ssbev_wino_weight_transform": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "ssbev_wino_weight_grad": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P]),
    "ssbev_wino_dgemm_packed_elems": (C.c_size_t, [C.c_int, C.c_int]),
    "ssbev_wino_dgemm_pack": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P]),
    "ssbev_wino_dgemm": (C.c_int, [_P, _P, _P, C.POINTER(WinoDims), C.c_int, _P]),
    "ssbev_wino_bgemm": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int, C.c_int, _P]),
    "ssbev_gemm_d2s_rowoff": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "ssbev_gemm_nn_workspace": (C.c_size_t, [C.POINTER(GemmDims)]),
    "ssbev_gemm_nt_workspace": (C.c_size_t, [C.POINTER(GemmDims)]),
    "ssbev_gemm_nn": (C.c_int, [_P, _P, _P, _P, C.POINTER(GemmDims), _P, C.c_size_t, _P]),
    "ssbev_gemm_nt": (C.c_int, [_P, _P, _P, _P, C.POINTER(GemmDims), _P, C.c_size_t, _P]),
    "ssbev_gemm_tn_workspace": (C.c_size_t, [C.POINTER(GemmDims)]),
    "ssbev_gemm_tn": (C.c_int, [_P, _P, _P, C.POINTER(GemmDims), _P, C.c_size_t, _P]),
    "ssbev_wino43_df_supported": (C.c_int, [C.POINTER(WinoDims), C.c_int]),
    "ssbev_wino43_df_instance": (C.c_int, [C.POINTER(WinoDims), C.c_int]),
    "ssbev_wino43_df_packed_elems": (C.c_size_t, [C.c_int, C.c_int]),
    "ssbev_wino43_df_pack": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P]),
    "ssbev_wino43_df_gemm": (C.c_int, [_P, _P, _P, C.POINTER(WinoDims), C.c_int, _P]),
    "ssbev_wino43_df_wgrad_workspace": (C.c_size_t, [C.POINTER(WinoDims), C.c_int]),
    "ssbev_wino43_df_wgrad": (C.c_int, [_P, _P, _P, C.POINTER(WinoDims), C.c_int, _P, C.c_size_t, _P]),
    "ssbev_wino2d_input_transform": (C.c_int, [_P, _P, C.POINTER(WinoDims), _P]),
    "ssbev_wino2d_output_transform": (C.c_int, [_P, _P, C.POINTER(WinoDims), _P]),
    "ssbev_wino2d_output_adjoint": (C.c_int, [_P, _P, C.POINTER(WinoDims), _P]),
    "ssbev_wino43_input_transform": (C.c_int, [_P, _P, C.POINTER(WinoDims), _P]),
    "ssbev_wino43_input_transform_bf16": (C.c_int, [_P, _P, C.POINTER(WinoDims), _P]),
    "ssbev_wino43_output_transform": (C.c_int, [_P, _P, C.POINTER(WinoDims), _P]),
    "ssbev_wino43_output_transform_bf16": (C.c_int, [_P, _P, C.POINTER(WinoDims), _P]),
    "ssbev_wino43_output_adjoint": (C.c_int, [_P, _P, C.POINTER(WinoDims), _P]),
    "ssbev_wino43_output_adjoint_bf16": (C.c_int, [_P, _P, C.POINTER(WinoDims), _P]),
    "ssbev_wino43_2d_input_transform": (C.c_int, [_P, _P, C.POINTER(WinoDims), _P]),
    "ssbev_wino43_2d_input_transform_bf16": (C.c_int, [_P, _P, C.POINTER(WinoDims), _P]),
    "ssbev_wino43_2d_output_transform": (C.c_int, [_P, _P, C.POINTER(WinoDims), _P]),
    "ssbev_wino43_2d_output_transform_acc": (C.c_int, [_P, _P, C.POINTER(WinoDims), _P]),
    "ssbev_wino2d_output_transform_acc": (C.c_int, [_P, _P, C.POINTER(WinoDims), _P]),
    "ssbev_wino43_2d_output_transform_bf16": (C.c_int, [_P, _P, C.POINTER(WinoDims), _P]),
    "ssbev_wino43_2d_output_adjoint": (C.c_int, [_P, _P, C.POINTER(WinoDims), _P]),
    "ssbev_wino43_2d_output_adjoint_bf16": (C.c_int, [_P, _P, C.POINTER(WinoDims), _P]),
    "ssbev_wino444_input_transform": (C.c_int, [_P, _P, C.POINTER(WinoDims), _P]),
    "ssbev_wino444_output_transform": (C.c_int, [_P, _P, C.POINTER(WinoDims), _P]),
    "ssbev_wino444_output_adjoint": (C.c_int, [_P, _P, C.POINTER(WinoDims), _P]),
    "ssbev_wino43_weight_transform": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "ssbev_wino43_weight_grad": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P]),
    "ssbev_wino_input_transform_bf16": (C.c_int, [_P, _P, C.POINTER(WinoDims), _P]),
    "ssbev_wino_output_transform_bf16": (C.c_int, [_P, _P, C.POINTER(WinoDims), _P]),
    "ssbev_wino_output_adjoint_bf16": (C.c_int, [_P, _P, C.POINTER(WinoDims), _P]),
    "ssbev_wino2d_input_transform_bf16": (C.c_int, [_P, _P, C.POINTER(WinoDims), _P]),
    "ssbev_wino2d_output_transform_bf16": (C.c_int, [_P, _P, C.POINTER(WinoDims), _P]),
    "ssbev_wino2d_output_adjoint_bf16": (C.c_int, [_P, _P, C.POINTER(WinoDims), _P]),
    "ssbev_wino_input_transform_bf16a": (C.c_int, [_P, _P, C.POINTER(WinoDims), _P]),
    "ssbev_wino_output_transform_bf16a": (C.c_int, [_P, _P, C.POINTER(WinoDims), _P]),
    "ssbev_wino_output_adjoint_bf16a": (C.c_int, [_P, _P, C.POINTER(WinoDims), _P]),
    "ssbev_wino2d_input_transform_bf16a": (C.c_int, [_P, _P, C.POINTER(WinoDims), _P]),
    "ssbev_wino2d_output_transform_bf16a": (C.c_int, [_P, _P, C.POINTER(WinoDims), _P]),
    "ssbev_wino2d_output_adjoint_bf16a": (C.c_int, [_P, _P, C.POINTER(WinoDims), _P]),
    "ssbev_softmax_axis_fwd": (C.c_int, [_P, _P, C.c_int64, C.c_int, C.c_int64, _P]),
    "ssbev_softmax_axis_bwd": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int, C.c_int64, _P]),
    "ssbev_softmax_rows_fwd": (C.c_int, [_P, _P, C.c_int64, C.c_int, _P]),
    "ssbev_softmax_rows_bwd": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int, _P]),
    "ssbev_occ_loss_num_sums": (C.c_int, []),
    "ssbev_occ_loss_workspace": (C.c_size_t, [C.POINTER(UpsampleDims)]),
    "ssbev_occ_loss_fwd": (C.c_int, [_P, _P, _P, _P, C.POINTER(UpsampleDims), _P, C.c_size_t, _P]),
    "ssbev_occ_loss_tail": (C.c_int, [_P, C.c_float, C.c_float, C.c_float, _P, _P, _P]),
    "ssbev_bri_shell_chunks": (C.c_int, []),
    "ssbev_bri_shell_pre_fwd": (C.c_int, [_P] * 12 + [C.c_int] * 3 + [_P]),
    "ssbev_bri_shell_post_fwd": (C.c_int, [_P] * 4 + [C.c_int] * 3 + [_P]),
    "ssbev_bri_shell_post_bwd": (C.c_int, [_P] * 6 + [C.c_int] * 3 + [_P]),
    "ssbev_bri_shell_pre_bwd": (C.c_int, [_P] * 15 + [C.c_int] * 3 + [_P]),
    "ssbev_occ_loss_bwd_workspace": (C.c_size_t, [C.POINTER(UpsampleDims)]),
    "ssbev_occ_loss_bwd": (C.c_int, [_P, _P, _P, _P, _P, C.POINTER(UpsampleDims), _P, C.c_size_t, _P]),
    "ssbev_grad_norm_workspace": (C.c_size_t, []),
    "ssbev_grad_norm": (C.c_int, [_P, C.c_int64, _P, _P, C.c_size_t, _P]),
    "ssbev_adamw_step": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.POINTER(AdamWCfg), _P, _P]),
}

_lib = None


def load():
    """dlopen the in-tree library (built by ``stereoscene_amd.build`` / ``__graft_entry__.build``)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SsbevError(f"{LIB_PATH} is missing: run `python -m stereoscene_amd.build` "
                             "(there is no CPU / PyTorch fallback for the HIP operators)")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)   # AttributeError if the .so does not export what the header declares
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def check(code, what):
    if code != OK:
        raise SsbevError(f"{what}: {_ERR.get(code, code)}")


def ptr(t):
    """Device pointer of a contiguous GPU tensor (or NULL for None)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise SsbevError("ssbev operators run on the MI355X only: got a CPU tensor (no CPU fallback exists)")
    if not t.is_contiguous():
        raise SsbevError("ssbev operators need contiguous buffers")
    return C.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream():
    """hipStream_t of torch's CURRENT stream on the current device.  ~880 calls per step: the raw accessor (one C call) instead of
    building a torch.cuda.Stream object each time (device-index resolution + is_available + an environment lookup: 8 us per call
    under the profiler, 3-4 ms of host time per step; tools/host_profile.py)."""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
