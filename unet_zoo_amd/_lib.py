"""ctypes binding of libunetzoo_hip.so (C ABI declared in include/unetzoo_hip.h).

The product path has no CPU or PyTorch fallback: if the shared library is missing, or an op is
handed a non-CUDA tensor, it raises.  Build with ``python -c "import __graft_entry__ as g; g.build()"``
(or ``make -C unet_zoo_amd/csrc``).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_void_p

import torch

UZ_F32, UZ_BF16 = 0, 1
TAPS_CONV, TAPS_GATHER2X2, TAPS_CONV_UP2, TAPS_CONV_S2 = 0, 1, 2, 3
STORE_PLAIN, STORE_SHUFFLE2X2 = 0, 1
PACK_CONV_FWD, PACK_CONV_DGRAD, PACK_CONVT_FWD, PACK_CONVT_DGRAD, PACK_IM2COL, PACK_VEC_REPEAT = range(6)

LIB_NAME = "libunetzoo_hip.so"
# UNET_ZOO_AMD_LIB: another build of the same ABI (tools/kbench.py points it at the ablation build); the product
# never sets it, and bench.py refuses a library whose uz_build_ablate() is 1
LIB_PATH = os.environ.get("UNET_ZOO_AMD_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), LIB_NAME)

# every symbol include/unetzoo_hip.h declares (tests check the .so exports all of them)
EXPORTS = (
    "uz_abi_version", "uz_last_error_string", "uz_build_ablate", "uz_source_hash", "uz_set_cu_reserve", "uz_get_cu_reserve", "uz_clock_probe", "uz_conv_igemm_grid_m", "uz_conv_igemm",
    "uz_wgrad_split", "uz_wgrad_workspace_bytes", "uz_wgrad", "uz_wgrad_kernel_name", "uz_wgrad_phase", "uz_wgrad_xf_supported", "uz_wgrad_xf", "uz_conv3x3_first_supported", "uz_conv3x3_first_rows", "uz_conv3x3_first_fwd",
    "uz_conv3x3_first_wgrad_workspace_bytes", "uz_conv3x3_first_wgrad", "uz_pack_weights", "uz_pack_weights_batched", "uz_pack_conv3x3_batched", "uz_im2col3x3_nchw", "uz_bn_finalize",
    "uz_bn_eval_scale", "uz_bn_relu_apply", "uz_bn_relu_bwd_workspace_bytes", "uz_bn_relu_bwd_reduce", "uz_bn_relu_bwd_apply",
    "uz_outconv_fwd", "uz_outconv_fwd_xf", "uz_outconv_bwd_workspace_bytes", "uz_outconv_bwd", "uz_outconv_bwd_rows", "uz_outconv_bwd_bnred",
    "uz_colsum",
    "uz_attn_grid", "uz_attn_psi_fwd", "uz_attn_gate_fwd", "uz_attn_bwd_psi", "uz_attn_bwd_reduce",
    "uz_attn_bwd_apply", "uz_sum_rows", "uz_sum_rows_f32", "uz_sum2x2",
    "uz_bn_relu_add_apply", "uz_bn_relu_add_apply_fin", "uz_bn_relu_bwd_reduce_rows", "uz_bn_relu_bwd_apply_fin", "uz_bilinear_fwd", "uz_bilinear_bwd", "uz_resize_bilinear_fwd", "uz_resize_bilinear_bwd", "uz_resample2",
    "uz_pool_grad_combine",
    "uz_sideconv3x3_fwd", "uz_sideconv3x3_bwd_workspace_bytes", "uz_sideconv3x3_bwd",
    "uz_fuse1x1_fwd", "uz_fuse1x1_bwd_workspace_bytes", "uz_fuse1x1_bwd",
    "uz_conv_igemm_workspace_bytes", "uz_conv_igemm_ws_grid_m", "uz_conv_igemm_ws",
    "uz_conv_igemm_bnred_supported", "uz_conv_igemm_bnred", "uz_conv_igemm_xf_supported", "uz_conv_igemm_xf", "uz_conv_igemm_kernel_name", "uz_bn_bwd_finalize", "uz_profile_arm", "uz_profile_disarm",
    "uz_patchify", "uz_layernorm_fwd", "uz_layernorm_bwd_rows", "uz_layernorm_bwd", "uz_layernorm_act_bwd",
    "uz_ln_head_fwd", "uz_ln_head_bwd_workspace_bytes", "uz_ln_head_bwd", "uz_sum_rows_f32_ld",
    "uz_winattn_fwd", "uz_winattn_bwd_rows", "uz_winattn_bwd",
    "uz_colsum_workspace_bytes", "uz_colsum_ws", "uz_colstats_rows", "uz_colstats", "uz_cpb_fwd", "uz_cpb_bwd",
    "uz_cpb_fwd_batched", "uz_cpb_bwd_batched_workspace_bytes", "uz_cpb_bwd_batched",
    "uz_clip_adamw_workspace_bytes", "uz_clip_adamw",
    "uz_gelu_fwd", "uz_gelu_bwd", "uz_dwconv3x3", "uz_dwconv3x3_wgrad_rows", "uz_dwconv3x3_wgrad",
    "uz_space_to_depth", "uz_im2col_nchw", "uz_sra_fwd", "uz_sra_bwd_workspace_bytes", "uz_sra_bwd",
    "uz_bce_dice_workspace_bytes", "uz_bce_dice", "uz_colsum_batched_workspace_bytes", "uz_colsum_batched", "uz_sum_rows_f32_batched", "uz_conv_igemm_res", "uz_wgrad_multi_workspace_bytes", "uz_wgrad_multi", "uz_conv_igemm_res_ws",
    "uz_add_relu", "uz_relu_bwd", "uz_pil_resample_h_u8", "uz_pil_resample_v_f32",
    "uz_gemm_nt", "uz_softmax_fwd", "uz_softmax_bwd", "uz_adaptive_avgpool_fwd", "uz_adaptive_avgpool_bwd",
    "uz_rowdot_f32", "uz_cast_rows", "uz_wgrad_batched_workspace_bytes", "uz_wgrad_batched", "uz_wgrad_batched2",
    "uz_softmax_workspace_bytes", "uz_add_map", "uz_dropout", "uz_chanscale_relu", "uz_chanattn_probs_fwd", "uz_chanattn_probs_bwd",
)


class ConvDesc(Structure):
    _fields_ = [(n, c_int) for n in (
        "dtype", "N", "H", "W", "Hin", "Win", "Cin", "ldx", "Nout", "ldy", "ntaps", "taps_mode",
        "dil", "store_mode", "Co", "Hout", "Wout")]


class WgradDesc(Structure):
    _fields_ = [(n, c_int) for n in (
        "dtype", "N", "H", "W", "Hr", "Wr", "Ci", "ldl", "Cj", "ldr", "ntaps", "taps_mode", "dil")]


class BnBwdDesc(Structure):
    _fields_ = [(n, c_int) for n in (
        "dtype", "N", "H", "W", "C", "ldy", "ldg0", "ldg1", "ldgp", "lddy", "pool_ceil")]


class LnDesc(Structure):
    _fields_ = [(n, c_int) for n in ("dtype", "N", "Ho", "Wo", "C", "ldx", "ldy", "ldr", "ldg", "lddx", "mode", "r")] \
        + [("eps", c_float), ("act", c_int)]


class WinAttnDesc(Structure):
    _fields_ = [(n, c_int) for n in ("dtype", "B", "H", "W", "C", "heads", "ws", "shift", "Nt", "ldq", "ldo")] \
        + [("scale", c_float)]


class SraDesc(Structure):
    _fields_ = [(n, c_int) for n in ("dtype", "B", "N", "NK", "heads", "head_dim", "kps", "ldq", "ldk", "ldv", "ldo")] \
        + [("scale", c_float)]


class GemmDesc(Structure):
    _fields_ = [(n, c_int) for n in ("dtype", "batch", "M", "N", "K", "ldx", "ldw", "ldy", "ldres")] \
        + [(n, ctypes.c_longlong) for n in ("xb", "wb", "yb", "resb")] + [("batch2", c_int)] \
        + [(n, ctypes.c_longlong) for n in ("xb2", "wb2", "yb2", "resb2")]


class WgradItem(Structure):
    _fields_ = [("desc", WgradDesc), ("L", c_void_p), ("R", c_void_p), ("out", c_void_p)]


class SumRowsItem(Structure):
    _fields_ = [("partial", c_void_p), ("out0", c_void_p), ("out1", c_void_p), ("rows", c_int), ("n", c_int), ("n0", c_int),
                ("reserved", c_int)]


class ColsumItem(Structure):
    _fields_ = [("x", c_void_p), ("out", c_void_p), ("P", c_int), ("C", c_int), ("ld", c_int), ("reserved", c_int)]


LN_PLAIN, LN_MERGE, LN_EXPAND = 0, 1, 2


class CpbItem(Structure):
    """uz_cpb_item: one continuous-position-bias MLP of a batched launch"""
    _fields_ = [(n, c_void_p) for n in ("idx", "w1", "b1", "w2", "b2")] \
        + [(n, c_int) for n in ("R", "hidden", "heads", "reserved")] \
        + [(n, c_void_p) for n in ("bias", "G", "dw1", "db1", "dw2", "db2")]


class PackItem(Structure):
    _fields_ = [("src", c_void_p), ("dst", c_void_p), ("begin", ctypes.c_longlong), ("mode", c_int),
                ("Co", c_int), ("Ci", c_int), ("T", c_int), ("Kpad", c_int), ("pad_", c_int)]


class Pack3x3Item(Structure):
    _fields_ = [("src", c_void_p), ("dst_fwd", c_void_p), ("dst_dgrad", c_void_p), ("Co", c_int), ("Ci", c_int),
                ("tile_begin", c_int), ("pad_", c_int)]


class HipLibraryError(RuntimeError):
    pass


_lib = None


def load():
    """Load the kernel library once; raise loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            f"{LIB_PATH} not found: the HIP kernel library is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` from the repo root. "
            "unet_zoo_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    lib.uz_abi_version.restype = c_int
    lib.uz_last_error_string.restype = c_char_p
    lib.uz_source_hash.restype = c_char_p
    lib.uz_source_hash.argtypes = []
    vp, ip, fp = c_void_p, c_int, c_float
    lib.uz_set_cu_reserve.argtypes = [c_int]
    lib.uz_clock_probe.argtypes = [c_int, vp, c_int, vp]
    lib.uz_get_cu_reserve.argtypes = []
    lib.uz_conv_igemm_grid_m.argtypes = [POINTER(ConvDesc)]
    lib.uz_conv_igemm.argtypes = [POINTER(ConvDesc), vp, vp, vp, vp, vp, vp]
    lib.uz_conv_igemm_res.argtypes = [POINTER(ConvDesc), vp, vp, vp, vp, ip, vp, vp]
    lib.uz_conv_igemm_res_ws.argtypes = [POINTER(ConvDesc), vp, vp, vp, vp, ip, vp, vp, vp]
    lib.uz_conv_igemm_workspace_bytes.argtypes = [POINTER(ConvDesc)]
    lib.uz_conv_igemm_ws_grid_m.argtypes = [POINTER(ConvDesc)]
    lib.uz_conv_igemm_ws.argtypes = [POINTER(ConvDesc), vp, vp, vp, vp, vp, vp, vp]
    lib.uz_conv_igemm_bnred_supported.argtypes = [POINTER(ConvDesc)]
    lib.uz_conv_igemm_xf_supported.argtypes = [POINTER(ConvDesc)]
    lib.uz_conv_igemm_xf.argtypes = [POINTER(ConvDesc), vp, vp, vp, vp, vp, vp, vp, vp]
    lib.uz_conv_igemm_kernel_name.argtypes = [POINTER(ConvDesc), c_int, c_char_p, c_int]
    lib.uz_profile_arm.argtypes = [vp, vp]
    lib.uz_profile_disarm.argtypes = []
    lib.uz_conv_igemm_bnred.argtypes = [POINTER(ConvDesc), vp, vp, vp, vp, c_int, vp, vp, vp, vp, vp, vp]
    lib.uz_bn_bwd_finalize.argtypes = [vp, c_int, c_int, vp, vp, vp, vp]
    lib.uz_patchify.argtypes = [ip, vp, ip, ip, ip, ip, ip, ip, vp, vp]
    lib.uz_layernorm_fwd.argtypes = [POINTER(LnDesc), vp, vp, vp, vp, vp, vp, vp, vp]
    lib.uz_layernorm_act_bwd.argtypes = [POINTER(LnDesc), vp, vp, vp, vp, vp, vp, vp, vp]
    lib.uz_layernorm_bwd_rows.argtypes = [POINTER(LnDesc)]
    lib.uz_layernorm_bwd.argtypes = [POINTER(LnDesc), vp, vp, vp, vp, vp, vp, vp, vp]
    lib.uz_ln_head_fwd.argtypes = [POINTER(LnDesc), vp, vp, vp, vp, vp, ip, vp, vp, vp]
    lib.uz_ln_head_bwd_workspace_bytes.argtypes = [POINTER(LnDesc), ip]
    lib.uz_ln_head_bwd.argtypes = [POINTER(LnDesc), vp, vp, vp, vp, ip, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.uz_sum_rows_f32_ld.argtypes = [vp, ip, ip, ip, vp, ip, vp, vp]
    lib.uz_winattn_fwd.argtypes = [POINTER(WinAttnDesc), vp, vp, vp, vp, vp, vp]
    lib.uz_winattn_bwd_rows.argtypes = [POINTER(WinAttnDesc)]
    lib.uz_winattn_bwd.argtypes = [POINTER(WinAttnDesc), vp, vp, vp, vp, vp, vp, ip, vp, ip, vp, vp]
    lib.uz_colstats_rows.argtypes = [ip, ip, ip]
    lib.uz_colstats.argtypes = [ip, vp, ip, ip, ip, vp, vp]
    lib.uz_cpb_fwd.argtypes = [vp, vp, vp, vp, vp, ip, ip, ip, vp, vp]
    lib.uz_cpb_bwd.argtypes = [vp, vp, vp, vp, vp, ip, ip, ip, vp, vp, vp, vp, vp]
    lib.uz_cpb_fwd_batched.argtypes = [vp, ip, vp]
    lib.uz_cpb_bwd_batched_workspace_bytes.argtypes = [vp, ip]
    lib.uz_cpb_bwd_batched.argtypes = [vp, ip, vp, vp]
    lib.uz_clip_adamw_workspace_bytes.argtypes = []
    lib.uz_clip_adamw.argtypes = [vp, vp, vp, vp, ctypes.c_longlong, fp, fp, fp, fp, fp, fp, vp, vp, vp]
    lib.uz_wgrad_split.argtypes = [POINTER(WgradDesc)]
    lib.uz_wgrad_workspace_bytes.argtypes = [POINTER(WgradDesc)]
    lib.uz_wgrad.argtypes = [POINTER(WgradDesc), vp, vp, vp, vp, vp]
    lib.uz_wgrad_kernel_name.argtypes = [POINTER(WgradDesc), c_char_p, c_int]
    lib.uz_wgrad_phase.argtypes = [POINTER(WgradDesc), vp, vp, vp, vp, vp, c_int]
    lib.uz_wgrad_xf_supported.argtypes = [POINTER(WgradDesc)]
    lib.uz_wgrad_xf.argtypes = [POINTER(WgradDesc), vp, vp, vp, vp, vp, vp, vp, c_int]
    lib.uz_conv3x3_first_supported.argtypes = [ip, ip, ip]
    lib.uz_conv3x3_first_rows.argtypes = [ip, ip, ip]
    lib.uz_conv3x3_first_fwd.argtypes = [ip, vp, ip, ip, ip, ip, vp, vp, ip, vp, ip, vp, vp]
    lib.uz_conv3x3_first_wgrad_workspace_bytes.argtypes = [ip, ip, ip, ip]
    lib.uz_conv3x3_first_wgrad_workspace_bytes.restype = ctypes.c_longlong
    lib.uz_conv3x3_first_wgrad.argtypes = [ip, vp, ip, ip, ip, ip, vp, ip, ip, vp, vp, vp]
    lib.uz_pack_weights.argtypes = [ip, ip, vp, ip, ip, ip, ip, vp, vp]
    lib.uz_pack_weights_batched.argtypes = [ip, vp, ip, ctypes.c_longlong, vp]
    lib.uz_pack_conv3x3_batched.argtypes = [ip, vp, ip, ip, vp]
    lib.uz_im2col3x3_nchw.argtypes = [ip, vp, ip, ip, ip, ip, ip, vp, vp]
    lib.uz_bn_finalize.argtypes = [vp, ip, ip, c_double, vp, vp, fp, fp, vp, vp, vp, vp, vp, vp, vp]
    lib.uz_bn_eval_scale.argtypes = [ip, vp, vp, vp, vp, fp, vp, vp, vp]
    lib.uz_bn_relu_apply.argtypes = [ip, vp, ip, vp, vp, ip, ip, ip, ip, vp, ip, vp, ip, vp]
    lib.uz_bn_relu_bwd_workspace_bytes.argtypes = [POINTER(BnBwdDesc), ip]
    lib.uz_bn_relu_bwd_reduce.argtypes = [POINTER(BnBwdDesc), vp, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                          vp, vp, vp]
    lib.uz_bn_relu_bwd_apply.argtypes = [POINTER(BnBwdDesc), vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                         c_double, vp, vp]
    lib.uz_outconv_fwd.argtypes = [ip, vp, ip, ip, ip, ip, vp, vp, ip, vp, vp]
    lib.uz_outconv_fwd_xf.argtypes = [ip, vp, ip, ip, ip, ip, vp, vp, vp, vp, ip, vp, vp]
    lib.uz_outconv_bwd_workspace_bytes.argtypes = [ip, ip, ip, ip, ip]
    lib.uz_outconv_bwd.argtypes = [ip, vp, ip, ip, ip, ip, vp, ip, vp, vp, ip, vp, vp, vp, vp]
    lib.uz_outconv_bwd_rows.argtypes = [ip, ip, ip, ip]
    lib.uz_outconv_bwd_bnred.argtypes = [ip, vp, ip, ip, ip, ip, vp, ip, vp, vp, ip, vp, vp, vp, vp, ip, vp, vp, vp, vp, vp, vp]
    lib.uz_colsum.argtypes = [ip, vp, ip, ip, ip, vp, vp]
    lib.uz_colsum_workspace_bytes.argtypes = [ip, ip, ip]
    lib.uz_colsum_ws.argtypes = [ip, vp, ip, ip, ip, vp, vp, vp]
    lib.uz_attn_grid.argtypes = [ip, ip, ip]
    lib.uz_attn_psi_fwd.argtypes = [ip, vp, ip, vp, ip, vp, vp, vp, vp, ip, ip, vp, vp, vp]
    lib.uz_attn_gate_fwd.argtypes = [ip, vp, ip, vp, vp, ip, ip, vp, ip, vp]
    lib.uz_attn_bwd_psi.argtypes = [ip, vp, ip, vp, ip, vp, vp, ip, ip, vp, ip, vp, vp, vp]
    lib.uz_attn_bwd_reduce.argtypes = [ip, vp, ip, vp, ip, vp, vp, vp, vp, vp, vp, vp, ip, ip, vp, vp]
    lib.uz_attn_bwd_apply.argtypes = [ip, vp, ip, vp, ip, vp, vp, vp, vp, vp, vp, vp, vp, ip, ip, vp, ip,
                                      vp, ip, vp]
    lib.uz_sum_rows.argtypes = [vp, ip, ip, vp, vp]
    lib.uz_sum_rows_f32.argtypes = [vp, ip, ip, vp, ip, vp, vp]
    lib.uz_sum2x2.argtypes = [ip, vp, ip, ip, ip, ip, ip, vp, ip, vp]
    ll = ctypes.c_longlong
    lib.uz_bn_relu_add_apply.argtypes = [ip, vp, ip, vp, vp, ip, ip, ip, ip, vp, ip, vp, ip, vp, ip, ip, vp]
    lib.uz_bn_relu_add_apply_fin.argtypes = [ip, vp, ip, vp, ip, ctypes.c_double, vp, vp, c_float, c_float, vp, vp, vp, vp,
                                             ip, ip, ip, ip, vp, ip, vp, ip, vp, ip, ip, vp]
    lib.uz_bn_relu_bwd_reduce_rows.argtypes = [POINTER(BnBwdDesc), vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.uz_bn_relu_bwd_apply_fin.argtypes = [POINTER(BnBwdDesc), vp, vp, vp, vp, vp, vp, vp, vp, vp, ip, vp, vp, vp, vp,
                                             ctypes.c_double, vp, vp]
    lib.uz_bilinear_fwd.argtypes = [ip, vp, ip, ll, ip, ip, ip, ip, vp, ip, ll, ip, ip, vp]
    lib.uz_bilinear_bwd.argtypes = [ip, vp, ip, ll, ip, ip, ip, ip, vp, ip, ll, ip, ip, vp]
    lib.uz_resize_bilinear_fwd.argtypes = [ip, vp, ip, ll, ip, ip, ip, ip, vp, ip, ll, ip, ip, ip, vp]
    lib.uz_resize_bilinear_bwd.argtypes = [ip, vp, ip, ll, ip, ip, ip, ip, vp, ip, ll, ip, ip, ip, vp]
    lib.uz_resample2.argtypes = [ip, vp, ip, ip, ip, ip, ip, vp, ip, ip, ip, ip, vp]
    lib.uz_pool_grad_combine.argtypes = [ip, ip, ip, ip, ip, vp, ip, vp, ip, vp, ip, vp, ip, vp, ip, ip, vp]
    lib.uz_sideconv3x3_fwd.argtypes = [ip, vp, ip, ip, ip, ip, ip, vp, vp, vp, vp, ll, vp]
    lib.uz_sideconv3x3_bwd_workspace_bytes.argtypes = [ip, ip, ip, ip, ip]
    lib.uz_sideconv3x3_bwd.argtypes = [ip, vp, ip, ip, ip, ip, ip, vp, vp, ll, vp, ip, vp, vp, vp, vp]
    lib.uz_fuse1x1_fwd.argtypes = [vp, ip, ip, ip, ip, vp, vp, vp, vp]
    lib.uz_fuse1x1_bwd_workspace_bytes.argtypes = [ip, ip, ip, ip]
    lib.uz_fuse1x1_bwd.argtypes = [vp, ip, ip, ip, ip, vp, vp, POINTER(c_void_p), ip, vp, vp, vp, vp, vp]
    lib.uz_sum_rows_f32_batched.argtypes = [POINTER(SumRowsItem), ip, vp]
    lib.uz_wgrad_multi_workspace_bytes.argtypes = [POINTER(WgradItem), ip]
    lib.uz_wgrad_multi_workspace_bytes.restype = ctypes.c_longlong
    lib.uz_wgrad_multi.argtypes = [POINTER(WgradItem), ip, vp, vp]
    lib.uz_colsum_batched_workspace_bytes.argtypes = [ip, POINTER(ColsumItem), ip]
    lib.uz_colsum_batched.argtypes = [ip, POINTER(ColsumItem), ip, vp, vp]
    lib.uz_bce_dice_workspace_bytes.argtypes = [ll]
    lib.uz_bce_dice.argtypes = [vp, vp, ll, vp, vp, vp, vp]
    lib.uz_gelu_fwd.argtypes = [ip, vp, ip, vp, ip, ll, ip, vp]
    lib.uz_gelu_bwd.argtypes = [ip, vp, ip, vp, ip, vp, ip, ll, ip, vp]
    lib.uz_dwconv3x3.argtypes = [ip, vp, ip, vp, vp, vp, ip, ip, ip, ip, ip, ip, vp]
    lib.uz_dwconv3x3_wgrad_rows.argtypes = [ip, ip, ip, ip, ip]
    lib.uz_dwconv3x3_wgrad.argtypes = [ip, vp, ip, vp, ip, vp, ip, ip, ip, ip, vp]
    lib.uz_space_to_depth.argtypes = [ip, vp, ip, vp, ip, ip, ip, ip, ip, ip, ip, vp]
    lib.uz_im2col_nchw.argtypes = [ip, vp, ip, ip, ip, ip, ip, ip, ip, ip, vp, vp]
    lib.uz_sra_fwd.argtypes = [POINTER(SraDesc), vp, vp, vp, vp, vp, vp]
    lib.uz_sra_bwd_workspace_bytes.argtypes = [POINTER(SraDesc)]
    lib.uz_sra_bwd.argtypes = [POINTER(SraDesc), vp, vp, vp, vp, vp, vp, ip, vp, ip, vp, ip, vp, vp]
    lib.uz_add_relu.argtypes = [ip, vp, ip, vp, ip, vp, ip, ll, ip, vp]
    lib.uz_relu_bwd.argtypes = [ip, vp, ip, vp, ip, vp, ip, ll, ip, vp]
    lib.uz_pil_resample_h_u8.argtypes = [vp, ip, ip, ip, vp, vp, ip, ip, vp, vp]
    lib.uz_pil_resample_v_f32.argtypes = [vp, ip, ip, ip, vp, vp, ip, ip, POINTER(c_float), POINTER(c_float), ip, vp, vp]
    lib.uz_wgrad_batched_workspace_bytes.argtypes = [POINTER(WgradDesc), ip]
    lib.uz_wgrad_batched.argtypes = [POINTER(WgradDesc), ip, vp, ll, vp, ll, vp, ll, vp, vp]
    lib.uz_wgrad_batched2.argtypes = [POINTER(WgradDesc), ip, ip, vp, ll, ll, vp, ll, ll, vp, ll, vp, vp]
    lib.uz_gemm_nt.argtypes = [POINTER(GemmDesc), vp, vp, vp, vp, vp, vp]
    lib.uz_softmax_workspace_bytes.argtypes = [ip, ip, ip, ip]
    lib.uz_softmax_fwd.argtypes = [ip, vp, ip, ll, ip, ip, ip, ip, c_float, vp, vp]
    lib.uz_chanattn_probs_fwd.argtypes = [ip, vp, ip, ip, ip, ip, c_float, c_float, vp, vp, vp]
    lib.uz_chanattn_probs_bwd.argtypes = [ip, vp, vp, ip, ip, ip, ip, c_float, c_float, vp, vp, vp]
    lib.uz_dropout.argtypes = [ip, vp, ip, vp, c_float, vp, ip, ll, ip, vp]
    lib.uz_chanscale_relu.argtypes = [ip, ip, vp, ip, vp, ip, vp, vp, ip, ip, ip, vp, ip, vp]
    lib.uz_add_map.argtypes = [ip, vp, ip, vp, vp, ip, ll, ip, ip, vp]
    lib.uz_softmax_bwd.argtypes = [ip, vp, vp, ip, ll, ip, ip, ip, ip, c_float, vp, ip, vp]
    lib.uz_adaptive_avgpool_fwd.argtypes = [ip, vp, ip, ip, ip, ip, ip, vp, ip, ip, ip, vp]
    lib.uz_adaptive_avgpool_bwd.argtypes = [ip, vp, ip, ip, ip, ip, ip, vp, ip, ip, ip, ip, vp]
    lib.uz_rowdot_f32.argtypes = [ip, vp, ip, vp, ip, ll, ip, vp, vp]
    lib.uz_cast_rows.argtypes = [ip, vp, ip, vp, ip, ll, ip, ip, vp]
    for name in EXPORTS:
        fn = getattr(lib, name)
        if name not in ("uz_last_error_string", "uz_source_hash"):
            fn.restype = ctypes.c_longlong if name.endswith("_workspace_bytes") else c_int
    if lib.uz_abi_version() != 1:
        raise HipLibraryError("libunetzoo_hip.so ABI version mismatch; rebuild it")
    _lib = lib
    return lib


def _fail(rc: int, what: str):
    msg = load().uz_last_error_string().decode("utf-8", "replace")
    raise HipLibraryError(f"{what} failed (code {rc}): {msg}")


def check(rc: int, what: str) -> None:
    """Status-returning entry points: anything but 0 is an error."""
    if rc != 0:
        _fail(rc, what)


def check_count(rc: int, what: str) -> int:
    """Count-returning entry points (grid_m, split): negative is an error."""
    if rc < 0:
        _fail(rc, what)
    return rc


def set_cu_reserve(n: int) -> None:
    """uz_set_cu_reserve(): size every persistent grid of the library for 256 - n CUs (process-wide; call before any
    plan is queried / any graph is captured)"""
    check(load().uz_set_cu_reserve(int(n)), "uz_set_cu_reserve")


def get_cu_reserve() -> int:
    return int(load().uz_get_cu_reserve())


def mfma_clock_ghz(settle_s: float = 0.6, iters: int = 40000) -> dict:
    """uz_clock_probe(): shader clock held under a dense bf16 MFMA stream on every CU, after `settle_s` seconds of that
    same load (the DVFS controller needs time); median / min / max over the 256 workgroups of the last launch, and the
    bf16 TFLOP/s the probe itself sustained.  Runs ~1 s; bench.py calls it OUTSIDE the timed region."""
    import time
    lib = load()
    dev = torch.device("cuda", torch.cuda.current_device())
    out = torch.zeros(256, 2, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    t_end = time.perf_counter() + settle_s
    while time.perf_counter() < t_end:
        for _ in range(4):
            check(lib.uz_clock_probe(iters, out.data_ptr(), 256, stream_ptr()), "uz_clock_probe")
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(lib.uz_clock_probe(iters, out.data_ptr(), 256, stream_ptr()), "uz_clock_probe")
    e1.record()
    torch.cuda.synchronize()
    o = out.double().cpu()
    ghz = (o[:, 0] / o[:, 1] / 10.0)
    flops = 256.0 * 8 * iters * 8 * 2.0 * 16 * 16 * 32
    return {"median_ghz": round(float(ghz.median()), 4), "min_ghz": round(float(ghz.min()), 4),
            "max_ghz": round(float(ghz.max()), 4),
            "probe_tflops": round(flops / (e0.elapsed_time(e1) * 1e-3) / 1e12, 1)}


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return UZ_F32
    if dt == torch.bfloat16:
        return UZ_BF16
    raise ValueError(f"unsupported run dtype {dt}; use torch.float32 or torch.bfloat16")


def require_cuda(*tensors: torch.Tensor) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise HipLibraryError(
                "unet_zoo_amd kernels run on an MI355X only: got a CPU tensor. Move the model and "
                "inputs to 'cuda' (there is no CPU fallback in the product path).")
