"""ctypes binding of libpixelhip.so (the C-ABI declared in include/pixelhip.h).

The product path has NO fallback: if the HIP library is missing this module raises at import of
the first symbol, and every op raises `PixelHipError` on a non-zero return code.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# (PXL_LIB_PATH: a variant build of the same library for A/B runs -- tools/build_alt.sh; the default is the in-tree build)
LIB_PATH = os.environ.get("PXL_LIB_PATH") or os.path.join(_HERE, "libpixelhip.so")

PXL_F32, PXL_BF16 = 0, 1
OP_INPUT, OP_CONV, OP_MAXPOOL, OP_RESIDUAL, OP_HEAD, OP_ACT, OP_IBN = 0, 1, 2, 3, 4, 5, 6
OP_AVGPOOL, OP_CONCAT, OP_UPCAT, OP_PIXSHUF = 7, 8, 9, 10


class PixelHipError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("B", C.c_int32), ("Hi", C.c_int32), ("Wi", C.c_int32),
                ("Cin", C.c_int32), ("Ho", C.c_int32), ("Wo", C.c_int32), ("Cout", C.c_int32),
                ("Kreal", C.c_int32), ("ntaps", C.c_int32), ("out_stride", C.c_int32),
                ("div", C.c_int32), ("relu_in", C.c_int32), ("tile_cfg", C.c_int32),
                ("stats_rep", C.c_int32), ("split_k", C.c_int32),
                ("dy", C.c_int16 * 64), ("dx", C.c_int16 * 64)]


class Op(C.Structure):
    _fields_ = [("kind", C.c_int32), ("in0", C.c_int32), ("in1", C.c_int32), ("out", C.c_int32),
                ("bn_in0", C.c_int32), ("bn_in1", C.c_int32), ("bn_out", C.c_int32),
                ("ngroups", C.c_int32), ("w_off", C.c_int32 * 4), ("b_off", C.c_int32 * 4),
                ("dil", C.c_int32 * 4), ("pads", C.c_int32 * 4), ("cin", C.c_int32),
                ("cout", C.c_int32), ("kh", C.c_int32), ("kw", C.c_int32), ("stride", C.c_int32),
                ("need_dgrad", C.c_int32), ("slope", C.c_float), ("c_off", C.c_int32)]


class PackItem(C.Structure):
    _fields_ = [("src_off", C.c_int64), ("wf_off", C.c_int64), ("wt_off", C.c_int64), ("K", C.c_int32),
                ("T", C.c_int32), ("C", C.c_int32), ("Cp", C.c_int32), ("T_total", C.c_int32),
                ("t_off", C.c_int32), ("Kp", C.c_int32)]


class UpdSeg(C.Structure):
    _fields_ = [("off", C.c_int64), ("n", C.c_int64), ("s_pk", C.c_int64), ("t_pk", C.c_int64)]


class BnDesc(C.Structure):
    _fields_ = [("C", C.c_int32), ("gamma_off", C.c_int32), ("beta_off", C.c_int32),
                ("rmean_off", C.c_int32), ("rvar_off", C.c_int32), ("eps", C.c_float),
                ("momentum", C.c_float)]


class BnFin(C.Structure):
    _fields_ = [("stats", C.c_void_p), ("nrep", C.c_int32), ("count", C.c_float), ("gamma", C.c_void_p),
                ("beta", C.c_void_p), ("running_mean", C.c_void_p), ("running_var", C.c_void_p),
                ("momentum", C.c_float), ("eps", C.c_float), ("training", C.c_int32), ("clamp_var", C.c_int32),
                ("coef", C.c_void_p)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p)
UPDATE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_long, C.c_long, C.c_void_p)      # pxl_update_fn(user, lo, hi, stream)

_P = C.c_void_p
_I = C.c_int
_L = C.c_long
_F = C.c_float
_Z = C.c_size_t

# name -> (restype, argtypes); every symbol include/pixelhip.h declares
SIGNATURES = {
    "pxl_last_error": (C.c_char_p, []),
    "pxl_version": (_I, []),
    "pxl_conv_igemm": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "pxl_conv_dma_bnin": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, C.POINTER(BnFin), _I, _P, _P]),
    "pxl_conv_dma_trace": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P]),
    "pxl_conv_dgrad_bnreduce": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _I, _P, _P]),
    "pxl_conv_dgrad_joinreduce": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "pxl_conv_wgrad": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _I, _I, _P]),
    "pxl_pack_weights": (_I, [_I, _P, _I, _I, _I, _P, _I, _I, _I, _P, _I, _P]),
    "pxl_pack_weights_batched": (_I, [_I, _P, _P, C.POINTER(PackItem), _I, _P]),
    "pxl_nchw_to_nhwc": (_I, [_I, _P, _P, _I, _I, _I, _I, _I, _P]),
    "pxl_nhwc_to_nchw": (_I, [_I, _P, _P, _I, _I, _I, _I, _I, _P]),
    "pxl_normalize_u8": (_I, [_I, _I, _L, _P, _P, _P, _P, _P]),
    "pxl_u8_to_f32": (_I, [_L, _P, _P, _I, _F, _P]),
    "pxl_bn_finalize": (_I, [_I, _P, _I, _F, _P, _P, _P, _P, _F, _F, _I, _I, _P, _P]),
    "pxl_bn_apply_fwd": (_I, [_I, _L, _I, _P, _P, _I, _P, _P]),
    "pxl_bn_fold_replicas": (_I, [_I, _I, _P, _P]),
    "pxl_bn_bwd_reduce": (_I, [_I, _I, _I, _P, _P, _P, _I, _P, _I, _P]),
    "pxl_conv_dma_finalize": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, C.POINTER(BnFin), _P, _P]),
    "pxl_bn_finalize_apply_fwd": (_I, [_I, _L, _I, _P, C.POINTER(BnFin), _I, _P, _P]),
    "pxl_residual_finalize_fwd": (_I, [_I, _L, _I, _P, C.POINTER(BnFin), _P, C.POINTER(BnFin), _P, _P]),
    "pxl_residual_bwd_reduce": (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "pxl_bn_bwd_finalize": (_I, [_I, _P, _I, _F, _P, _P, _P, _I, _P]),
    "pxl_bn_bwd_apply": (_I, [_I, _I, _I, _P, _P, _P, _P, _I, _P, _P]),
    "pxl_bn_param_grad": (_I, [_I, _P, _P, _P, _P]),
    "pxl_bn_bwd_apply_fused": (_I, [_I, _I, _I, _P, _P, _P, _P, _F, _I, _I, _P, _P, _P, _P]),
    "pxl_ibn_stats": (_I, [_I, _I, _I, _I, _P, _P, _P]),
    "pxl_ibn_stats_acc": (_I, [_I, _I, _I, _I, _P, _P, _P]),
    "pxl_ibn_fold": (_I, [_I, _I, _I, _P, _P, _P, _P, _P]),
    "pxl_ibn_coef": (_I, [_I, _I, _I, _I, _F, _P, _P, _P, _P, _P, _P, _F, _F, _I, _I, _P, _P]),
    "pxl_ibn_apply_fwd": (_I, [_I, _I, _I, _I, _P, _P, _F, _P, _P]),
    "pxl_ibn_bwd_reduce": (_I, [_I, _I, _I, _I, _P, _P, _P, _F, _P, _P]),
    "pxl_ibn_bwd_reduce_acc": (_I, [_I, _I, _I, _I, _P, _P, _P, _F, _P, _P]),
    "pxl_ibn_bwd_apply": (_I, [_I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _F, _I, _F, _P, _P]),
    "pxl_residual_fwd": (_I, [_I, _L, _I, _P, _P, _P, _P, _P, _P]),
    "pxl_residual_fwd_bits": (_I, [_I, _L, _I, _P, _P, _P, _P, _P, _P, _P]),
    "pxl_residual_finalize_fwd_bits": (_I, [_I, _L, _I, _P, C.POINTER(BnFin), _P, C.POINTER(BnFin), _P, _P, _P]),
    "pxl_conv_dgrad_joinreduce_bits": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "pxl_leaky_fwd": (_I, [_I, _L, _P, _F, _P, _P]),
    "pxl_leaky_bwd": (_I, [_I, _L, _P, _P, _F, _P, _P]),
    "pxl_relu_mask": (_I, [_I, _L, _P, _P, _P, _P, _P]),
    "pxl_colsum": (_I, [_I, _I, _I, _I, _P, _P, _P]),
    "pxl_vec_sum4": (_I, [_I, _P, _P, _P, _P, _P, _P]),
    "pxl_add_inplace": (_I, [_I, _L, _P, _P, _P]),
    "pxl_maxpool3x3s2_fwd": (_I, [_I, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "pxl_maxpool3x3s2_bwd": (_I, [_I, _I, _I, _I, _I, _P, _P, _P, _P]),
    "pxl_adaptive_avgpool_fwd": (_I, [_I, _I, _I, _I, _I, _I, _P, _P, _P]),
    "pxl_adaptive_avgpool_bwd": (_I, [_I, _I, _I, _I, _I, _I, _P, _P, _I, _P]),
    "pxl_upsample_slice_fwd": (_I, [_I, _I, _I, _I, _I, _I, _P, _P, _I, _I, _I, _P, _I, _I, _P]),
    "pxl_upsample_slice_bwd": (_I, [_I, _I, _I, _I, _I, _I, _P, _I, _I, _I, _I, _P, _P]),
    "pxl_slice_copy": (_I, [_I, _L, _I, _P, _I, _I, _P, _I, _I, _I, _P]),
    "pxl_pixshuf_relu_fwd": (_I, [_I, _I, _I, _I, _I, _I, _P, _P, _I, _P]),
    "pxl_pixshuf_relu_bwd": (_I, [_I, _I, _I, _I, _I, _I, _P, _I, _P, _P, _P]),
    "pxl_latent_perturb": (_I, [_I, _I, _L, _P, _P, _P, _P, _P, _F, _P, _P]),
    "pxl_fg_mask_nearest": (_I, [_I, _I, _I, _I, _P, _I, _I, _I, _P, _P]),
    "pxl_chan_mean": (_I, [_I, _I, _L, _P, _P, _P]),
    "pxl_fdrop_mask": (_I, [_I, _L, _P, _F, _P, _P]),
    "pxl_l2_normalize_persample": (_I, [_I, _L, _P, _F, _P, _P, _P]),
    "pxl_sub_scale": (_I, [_L, _P, _P, _F, _P, _P]),
    "pxl_external_contour_boxes_host": (_I, [_P, _I, _I, _I, _P, _I, C.POINTER(_I)]),
    "pxl_upsample_softmax_fwd": (_I, [_I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    "pxl_upsample_bwd_workspace": (_Z, [_I, _I, _I, _I]),
    "pxl_upsample_softmax_bwd": (_I, [_I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _Z, _P]),
    "pxl_head_loss_lds_bytes": (_Z, [_I, _I, _I]),
    "pxl_head_loss": (_I, [_I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _I, _I, _I, _I, _F, _F, _P, _P, _Z, _P, _P]),
    "pxl_ce_fwd": (_I, [_I, _I, _I, _P, _P, _I, _P, _P]),
    "pxl_ce_bwd": (_I, [_I, _I, _I, _P, _P, _I, _P, _P, _P]),
    "pxl_ce_mse_bwd": (_I, [_I, _I, _I, _P, _P, _I, _I, _P, _P, _I, _I, _P, _P, _P]),
    "pxl_bce_logits_masked_fwd": (_I, [_I, _L, _P, _P, _I, _F, _P, _P]),
    "pxl_bce_logits_masked_bwd": (_I, [_I, _L, _P, _P, _I, _F, _P, _P, _P]),
    "pxl_cutmix_mix": (_I, [_I, _I, _L, _P, _P, _P, _P, _F, _P, _P]),
    "pxl_fcd_prepare": (_I, [_L, _P, _P, _I, _F, _P, _P, _P]),
    "pxl_bce_logits_fwd": (_I, [_I, _L, _P, _P, _P, _P]),
    "pxl_bce_logits_bwd": (_I, [_I, _L, _P, _P, _P, _P, _P]),
    "pxl_softmax_nchw_fwd": (_I, [_I, _I, _L, _P, _P, _P]),
    "pxl_softmax_nchw_bwd": (_I, [_I, _I, _L, _P, _P, _P, _P]),
    "pxl_confusion_matrix": (_I, [_I, _I, _L, _P, _P, _P, _P]),
    "pxl_argmax_u8": (_I, [_I, _I, _L, _P, _P, _P]),
    "pxl_mse_fwd": (_I, [_L, _P, _P, _P, _P]),
    "pxl_mse_bwd": (_I, [_L, _P, _P, _P, _P, _P]),
    "pxl_absdiff_chansum": (_I, [_I, _I, _L, _P, _P, _I, _F, _P, _P]),
    "pxl_absdiff_chansum_dense": (_I, [_I, _I, _L, _P, _P, _F, _P, _P]),
    "pxl_onehot_ignore": (_I, [_I, _I, _L, _P, _I, _P, _P]),
    "pxl_gauss_sep_reflect": (_I, [_I, _I, _I, _P, _P, _I, _P, _P, _P]),
    "pxl_clamp_min0_inplace": (_I, [_L, _P, _P]),
    "pxl_dilate3_reflect": (_I, [_I, _I, _I, _P, _P, _P]),
    "pxl_minmax_norm_persample": (_I, [_I, _L, _P, _F, _P, _P, _P]),
    "pxl_dcgt": (_I, [_I, _I, _L, _P, _P, _P, _P, _F, _P, _P, _P, _P]),
    "pxl_masked_sq_mean_fwd": (_I, [_L, _P, _P, _P, _P]),
    "pxl_masked_sq_mean_bwd": (_I, [_L, _P, _P, _P, _P, _P]),
    "pxl_mse_persample_fwd": (_I, [_I, _L, _P, _P, _P, _P]),
    "pxl_mse_persample_bwd": (_I, [_I, _L, _P, _P, _P, _P, _P]),
    "pxl_sgd_step": (_I, [_L, _P, _P, _P, _F, _F, _F, _I, _P]),
    "pxl_sgd_step_general": (_I, [_L, _P, _P, _P, _F, _F, _F, _F, _I, _I, _P]),
    "pxl_adam_step_wd": (_I, [_L, _P, _P, _P, _P, _F, _F, _F, _F, _F, _I, _P]),
    "pxl_gaussian_noise_apply": (_I, [_I, _L, _P, _P, _P, _P]),
    "pxl_rotate_append": (_I, [_I, _I, _I, _I, _P, _P, _P, _P]),
    "pxl_adam_step": (_I, [_L, _P, _P, _P, _P, _F, _F, _F, _F, _I, _P]),
    "pxl_ema_update": (_I, [_L, _P, _P, _F, _P]),
    "pxl_scale_inplace": (_I, [_L, _P, _F, _P]),
    "pxl_tune_set": (_I, [_I, _I]),
    "pxl_comm_available": (_I, []),
    "pxl_comm_unique_id": (_I, [_P]),
    "pxl_comm_init": (_I, [_P, _I, _I, C.POINTER(_P)]),
    "pxl_comm_destroy": (None, [_P]),
    "pxl_comm_allreduce_sum": (_I, [_P, _P, _L, _P]),
    "pxl_comm_allreduce_hook": (_I, [_P, _P, _I, _P]),
    "pxl_stream_pool_init": (_I, [_P, C.POINTER(_I)]),
    "pxl_stream_role": (_P, [_I]),
    "pxl_stream_pool_probes": (_I, []),
    "pxl_peer_create": (_I, [_I, _I, _I, _I, C.POINTER(_P)]),
    "pxl_peer_handle": (_I, [_P, _P]),
    "pxl_peer_open": (_I, [_P, _P]),
    "pxl_peer_destroy": (None, [_P]),
    "pxl_peer_allreduce_sum": (_I, [_P, _P, _L, _P]),
    "pxl_peer_allreduce_fold": (_I, [_P, _P, _P, _L, _I, _P]),
    "pxl_peer_allreduce_bnbwd": (_I, [_P, _P, _I, _P, _P, _P]),
    "pxl_peer_allreduce_hook": (_I, [_P, _P, _I, _P]),
    "pxl_peer_status": (_I, [_P, C.POINTER(_I)]),
    "pxl_peer_status_nosync": (_I, [_P]),
    "pxl_peer_exchanges": (_L, [_P]),
    # seams between the library's translation units (include/pixelhip.h, last section)
    "pxl_conv_dma_eligible": (_I, [C.POINTER(ConvDesc), _P, _P]),
    "pxl_conv_dma": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "pxl_splitk_finish": (_I, [_I, _L, _I, _I, _P, _P, _P, _P]),
    "pxl_net_make_patches": (_I, [_P, _P, _P, _Z, C.POINTER(_P), C.POINTER(_Z), _P]),
    "pxl_net_borrow_patches": (_I, [_P, _P, _Z]),
    "pxl_conv_dma_slabs": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _Z, _I, _P]),
    "pxl_aspp_col2im": (_I, [_I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _I, _Z, _P, _P, _I, _P]),
    "pxl_aspp_dp_gather": (_I, [_I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _I, _P, _P]),
    "pxl_aspp_dw_scatter": (_I, [_P, _I, _I, _I, _I, _I, _I, _P, _P, _P]),
    "pxl_aspp_pack": (_I, [_I, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P]),
    "pxl_splitk_finish_slabs": (_I, [_I, _L, _I, _I, _I, _P, _P, _P, _P]),
    "pxl_conv_wgrad_dma_eligible": (_I, [C.POINTER(ConvDesc), _P]),
    "pxl_conv_wgrad_dma": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _I, _I, _P]),
    "pxl_dma_capture_begin": (None, [C.POINTER(_P)]),
    "pxl_dma_capture_end": (None, []),
    "pxl_dma_launch_captured": (_I, [_P, _P]),
    "pxl_elt_pair_begin": (None, []),
    "pxl_elt_pair_end": (_I, []),
    "pxl_net_create": (_I, [_I, _I, C.POINTER(Op), _I, C.POINTER(BnDesc), _I, _I, C.POINTER(_P)]),
    "pxl_net_destroy": (None, [_P]),
    "pxl_net_plan": (_I, [_P, _I, _I, _I]),
    "pxl_net_plan_out": (_I, [_P, _I, _I, _I, _I, _I]),
    "pxl_net_packed_bytes": (_Z, [_P]),
    "pxl_net_arena_bytes": (_Z, [_P]),
    "pxl_net_scratch_bytes": (_Z, [_P]),
    "pxl_net_set_sync": (_I, [_P, ALLREDUCE_FN, _P, _I]),
    "pxl_net_set_grad_sync": (_I, [_P, ALLREDUCE_FN, _P, _I, _L, _L]),
    "pxl_net_grad_buckets": (_I, [_P]),
    "pxl_net_set_update_hook": (_I, [_P, UPDATE_FN, _P, _L, _L, _L]),
    "pxl_net_update_buckets": (_I, [_P]),
    "pxl_net_pack_range": (_I, [_P, _P, _P, _I, _L, _L, _P]),
    "pxl_net_update_segments": (_I, [_P, _P, C.POINTER(UpdSeg), _I]),
    "pxl_sgd_ema_pack": (_I, [_L, _L, _P, _P, _P, _P, _I, C.POINTER(_L), C.POINTER(_F), C.POINTER(_P), _F, _F, _F, _P, _P, _I, _P, _P, _I, _P]),
    "pxl_net_tune": (_I, [_P, _P, _P, _P, _P, _Z, _P, _Z, _P]),
    "pxl_net_pack": (_I, [_P, _P, _P, _P]),
    "pxl_net_pack_parts": (_I, [_P, _P, _P, _I, _P]),
    "pxl_net_forward": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _Z, _I, _P]),
    "pxl_net_forward_pair": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _Z, _Z, _I, _I, _P]),
    "pxl_net_tune_pair": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _Z, _Z, _P]),
    "pxl_net_pairs": (_I, [_P]),
    "pxl_net_set_bn_repeat": (_I, [_P, _I]),
    "pxl_net_set_tune_dual": (_I, [_P, _I]),
    "pxl_net_pair_syncs": (_I, [_P]),
    "pxl_net_latent": (_I, [_P, _P, _P, _P]),
    "pxl_net_latent_shape": (_I, [_P, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)]),
    "pxl_net_read_tensor": (_I, [_P, _P, _I, _I, _P, _P, _P]),
    "pxl_net_tensor_bytes": (_Z, [_P, _I]),
    "pxl_net_tensor_shape": (_I, [_P, _I, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)]),
    "pxl_net_input_grad": (_I, [_P, _P, _P, _P]),
    "pxl_net_set_input_parts": (_I, [_P, _I, _P, _P]),
    "pxl_net_input_grad_parts": (_I, [_P, _P, _I, _P, _P, _P]),
    "pxl_stem_patches": (_I, [_I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "pxl_nchw_parts_to_nhwc": (_I, [_I, _I, _P, _P, _P, _I, _I, _I, _I, _P]),
    "pxl_nhwc_to_nchw_parts": (_I, [_I, _P, _I, _P, _P, _I, _I, _I, _I, _P]),
    "pxl_net_seed_latent_grad": (_I, [_P, _P, _Z, _P, _P]),
    "pxl_net_set_wgrad": (_I, [_P, _I]),
    "pxl_net_set_pack_dgrad": (_I, [_P, _I]),
    "pxl_net_profile": (_I, [_P, _I]),
    "pxl_net_profile_read": (_I, [_P, _I, C.POINTER(C.c_double), C.POINTER(C.c_long), C.POINTER(C.c_double)]),
    "pxl_net_profile_bytes": (_I, [_P, _I, C.POINTER(C.c_double)]),
    "pxl_net_head_loss_supported": (_I, [_P]),
    "pxl_net_head_forward": (_I, [_P, _P, _P, _P, _P]),
    "pxl_net_head_loss": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _F, _P, _Z, _P, _P]),
    "pxl_colsum_ordered": (_I, [_I, _I, _I, _I, _P, _P, _P]),
    "pxl_residual_bwd_reduce_rep": (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    "pxl_head_loss_ex": (_I, [_I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _I, _I, _I, _I, _F, _F, _P, _I, _I, _P, _P, _Z, _P, _P]),
    "pxl_hyper_set": (_I, [_P, _P, _I, _P]),
    "pxl_sgd_step_hp": (_I, [_L, _P, _P, _P, _P, _F, _F, _I, _P]),
    "pxl_ema_update_hp": (_I, [_L, _P, _P, _P, _P]),
    "pxl_head_loss_hp": (_I, [_I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _I, _I, _I, _I, _F, _P, _P, _P, _Z, _P, _P]),
    "pxl_net_head_loss_hp": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P, _P, _Z, _P, _P]),
    "pxl_net_backward_low": (_I, [_P, _P, _P, _P, _P, _Z, _P, _Z, _I, _P]),
    "pxl_cons_head_lds_bytes": (_Z, [_I, _I, _I]),
    "pxl_cons_head_workspace": (_Z, [_I, _I, _I, _I]),
    "pxl_cons_head_fwd": (_I, [_I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _Z, _P, _I, _P]),
    "pxl_cons_head_bwd": (_I, [_I, _I, _I, _I, _I, _I, _I, _I, _P, _Z, _P, _P, _P]),
    "pxl_net_cons_head_supported": (_I, [_P]),
    "pxl_net_cons_head_fwd": (_I, [_P, _P, _P, _P, _Z, _P, _P]),
    "pxl_net_cons_head_bwd": (_I, [_P, _P, _Z, _P, _P]),
    "pxl_net_backward": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _Z, _P, _Z, _I, _P]),
}

_lib = None


def lib():
    """Load (once) and return the ctypes handle; raises if the HIP library was not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PixelHipError(
                "libpixelhip.so is missing (%s). Build it with `python -m pixelssl_amd.build`; "
                "there is no CPU/PyTorch fallback for the accelerated path." % LIB_PATH)
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)
            fn.restype = res
            fn.argtypes = args
        _lib = h
    return _lib


class _CallProfile:
    """Proxy of the ctypes handle that accumulates host wall time and call counts per C-ABI symbol (bench.py --host-profile:
    where a host-paced step spends its time).  Off unless enable_call_profile() was called."""

    def __init__(self, h):
        self._h, self.stats = h, {}

    def __getattr__(self, name):
        import time
        fn, st = getattr(self._h, name), self.stats.setdefault(name, [0, 0.0])

        def timed(*a):
            t0 = time.perf_counter()
            r = fn(*a)
            st[1] += time.perf_counter() - t0
            st[0] += 1
            return r
        setattr(self, name, timed)
        return timed


def enable_call_profile():
    """Route every later lib() call through a timing proxy; returns its {symbol: [calls, seconds]} table."""
    global _lib
    h = lib()
    if not isinstance(h, _CallProfile):
        _lib = _CallProfile(h)
    return _lib.stats


def check(rc):
    if rc != 0:
        raise PixelHipError("libpixelhip error %d: %s" % (rc, lib().pxl_last_error().decode()))


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def dtype_code(dtype):
    if dtype in (torch.float32, "fp32", "f32", PXL_F32):
        return PXL_F32
    if dtype in (torch.bfloat16, "bf16", PXL_BF16):
        return PXL_BF16
    raise ValueError("unsupported engine dtype %r" % (dtype,))


def torch_dtype(code):
    return torch.float32 if code == PXL_F32 else torch.bfloat16
