"""ctypes binding of libvqwnet_hip.so (the C ABI declared in include/vqwnet_hip.h).

The library is built in-tree by `make -C medical-image-editing_amd/csrc` (or
__graft_entry__.build()).  There is NO fallback: if the shared object is missing
every operator raises — the product path never routes through PyTorch eager
kernels or the CPU oracle.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VQW_LIB_PATH") or os.path.join(os.path.dirname(_HERE), "lib", "libvqwnet_hip.so")   # override: A/B of two builds

c_p = ctypes.c_void_p
c_i = ctypes.c_int
c_l = ctypes.c_long
c_f = ctypes.c_float
c_d = ctypes.c_double
c_sz = ctypes.c_size_t

# name -> (restype, argtypes); mirrors include/vqwnet_hip.h one to one
SIGNATURES = {
    "vqw_last_error": (ctypes.c_char_p, []),
    "vqw_abi_version": (c_i, []),
    "vqw_set_conv_backend": (c_i, [c_i]),
    "vqw_profile_begin": (c_i, []),
    "vqw_profile_end": (c_i, [c_p]),
    "vqw_profile_families": (c_i, [ctypes.c_uint]),
    "vqw_conv2d_fwd": (c_i, [c_p, c_i, c_i, c_p, c_i, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "vqw_pack_dgrad_weights": (c_i, [c_p, c_p, c_i, c_i, c_i, c_p]),
    "vqw_conv2d_fwd_stats_parts": (c_i, [c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i]),
    "vqw_conv2d_fwd_stats": (c_i, [c_p, c_i, c_i, c_p, c_i, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "vqw_conv2d_wgrad_ws_bytes": (c_sz, [c_i, c_i, c_i, c_i, c_i, c_i, c_i]),
    "vqw_conv2d_wgrad": (c_i, [c_p, c_i, c_i, c_p, c_i, c_p, c_p, c_p, c_p, c_sz, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "vqw_conv3x3_up2_supported": (c_i, [c_i, c_i, c_i, c_i, c_i]),
    "vqw_conv3x3_up2_ws_bytes": (c_sz, [c_i, c_i]),
    "vqw_conv3x3_up2_prepare": (c_i, [c_p, c_p, c_sz, c_i, c_i, c_p]),
    "vqw_conv3x3_up2_fwd": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "vqw_conv3x3_up2_fwd_stats_parts": (c_i, [c_i, c_i, c_i, c_i, c_i]),
    "vqw_conv3x3_up2_fwd_stats": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    "vqw_conv3x3_up2_fwd_pair_supported": (c_i, [c_i] * 5),
    "vqw_conv3x3_up2_fwd_pair": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    "vqw_conv3x3_up2_dgrad": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    "vqw_conv3x3_up2_dgrad_acc_supported": (c_i, [c_i] * 5),
    "vqw_conv3x3_up2_dgrad_acc": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    "vqw_conv2d_fwd_acc_supported": (c_i, [c_i, c_i, c_i, c_i, c_i, c_i, c_i]),
    "vqw_conv2d_fwd_acc": (c_i, [c_p, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "vqw_conv3x3_wino_supported": (c_i, [c_i, c_i, c_i, c_i, c_i]),
    "vqw_conv3x3_wino_ws_bytes": (c_sz, [c_i, c_i]),
    "vqw_conv3x3_wino_prepare": (c_i, [c_p, c_p, c_sz, c_i, c_i, c_p]),
    "vqw_conv3x3_wino_prepare_dgrad": (c_i, [c_p, c_p, c_sz, c_i, c_i, c_p]),
    "vqw_conv3x3_wino_fwd": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "vqw_conv3x3_wino_masked_supported": (c_i, [c_i] * 5),
    "vqw_conv3x3_wino_fwd_masked": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    "vqw_conv3x3_wino_fwd_acc": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    "vqw_conv3x3_wino_split_supported": (c_i, [c_i] * 7),
    "vqw_conv3x3_wino_dil2_supported": (c_i, [c_i] * 5),
    "vqw_conv3x3_wino_dil2_stats_parts": (c_i, [c_i] * 5),
    "vqw_conv3x3_wino_dil2_fwd": (c_i, [c_p, c_p, c_p, c_p, c_p] + [c_i] * 7 + [c_p]),
    "vqw_conv3x3_wino_fwd_split": (c_i, [c_p, c_p, c_p, c_p, c_p] + [c_i] * 8 + [c_p]),
    "vqw_conv3x3_wino_split_padded_supported": (c_i, [c_i] * 8),
    "vqw_conv3x3_wino_fwd_split_padded": (c_i, [c_p, c_p, c_p, c_p, c_p] + [c_i] * 9 + [c_p]),
    "vqw_conv3x3_wino_fwd_inbwd_parts": (c_i, [c_i] * 5),
    "vqw_conv3x3_wino_fwd_inbwd": (c_i, [c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    "vqw_inorm_bwd_parts": (c_i, [c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_p]),
    "vqw_conv3x3_wino_fwd_stats_parts": (c_i, [c_i, c_i, c_i, c_i, c_i]),
    "vqw_conv3x3_wino_fwd_stats": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    "vqw_conv3x3_up2_wgrad_supported": (c_i, [c_i, c_i, c_i, c_i, c_i]),
    "vqw_conv3x3_up2_wgrad_ws_bytes": (c_sz, [c_i, c_i, c_i, c_i, c_i]),
    "vqw_conv3x3_up2_wgrad": (c_i, [c_p, c_p, c_p, c_p, c_p, c_sz, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "vqw_input_grad_gather": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_i, c_i, c_i, c_p]),
    "vqw_plane_ws_bytes": (c_sz, [c_i, c_i, c_i]),
    "vqw_inorm_fwd": (c_i, [c_p, c_p, c_i, c_i, c_p, c_p, c_sz, c_i, c_i, c_i, c_f, c_i, c_p]),
    "vqw_inorm_fwd_parts": (c_i, [c_p, c_p, c_i, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_i, c_p]),
    "vqw_inorm_stats": (c_i, [c_p, c_p, c_p, c_sz, c_i, c_i, c_i, c_f, c_p]),
    "vqw_inorm_stats_parts": (c_i, [c_p, c_i, c_p, c_i, c_i, c_i, c_f, c_p]),
    "vqw_inorm_stats_parts2": (c_i, [c_p, c_i, c_p, c_p, c_i, c_p, c_i, c_i, c_i, c_f, c_p]),
    "vqw_inorm_add_supported": (c_i, [c_i]),
    "vqw_inorm_add_fwd": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p]),
    "vqw_inorm_bwd": (c_i, [c_p, c_p, c_p, c_i, c_i, c_p, c_p, c_sz, c_i, c_i, c_i, c_i, c_p]),
    "vqw_inorm_bwd_pair": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_sz, c_i, c_i, c_i, c_p]),
    "vqw_res_tail_bwd_pair": (c_i, [c_p] * 11 + [c_sz, c_i, c_i, c_i, c_i, c_p]),
    "vqw_bn_partial_stats": (c_i, [c_p, c_p, c_p, c_sz, c_i, c_i, c_i, c_p]),
    "vqw_bn_stats_from_parts": (c_i, [c_p, c_p, c_i, c_i, c_d, c_p]),
    "vqw_bn_finalize": (c_i, [c_p, c_d, c_p, c_p, c_p, c_f, c_f, c_i, c_p]),
    "vqw_bn_finalize_parts": (c_i, [c_p, c_i, c_d, c_p, c_d, c_p, c_p, c_p, c_f, c_f, c_i, c_p]),
    "vqw_bn_eval_stats": (c_i, [c_p, c_p, c_p, c_f, c_i, c_p]),
    "vqw_spade_fwd": (c_i, [c_p, c_p, c_p, c_p, c_i, c_p, c_l, c_i, c_i, c_p]),
    "vqw_spade_fwd_res": (c_i, [c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_l, c_i, c_i, c_p]),
    "vqw_spade_fwd_res_norm_supported": (c_i, [c_l, c_i]),
    "vqw_spade_fwd_res_norm": (c_i, [c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_i, c_p, c_i, c_l, c_i, c_i, c_p]),
    "vqw_spade_bwd_reduce": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_sz, c_i, c_i, c_i, c_i, c_p]),
    "vqw_spade_bwd_apply": (c_i, [c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_d, c_p, c_l, c_i, c_i, c_i, c_p]),
    "vqw_add": (c_i, [c_p, c_p, c_p, c_l, c_i, c_p]),
    "vqw_relu_bwd": (c_i, [c_p, c_p, c_p, c_l, c_p]),
    "vqw_maxpool2_fwd": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_p]),
    "vqw_maxpool2_bwd": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p]),
    "vqw_res_tail_fwd": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p]),
    "vqw_res_tail_norm_fwd": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p]),
    "vqw_res_tail_bwd": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p]),
    "vqw_tanh_fwd": (c_i, [c_p, c_p, c_l, c_p]),
    "vqw_tanh_bwd": (c_i, [c_p, c_p, c_p, c_l, c_p]),
    "vqw_affine": (c_i, [c_p, c_p, c_f, c_f, c_l, c_p]),
    "vqw_mse_fwd": (c_i, [c_p, c_p, c_p, c_p, c_sz, c_l, c_p]),
    "vqw_mse_bwd": (c_i, [c_p, c_p, c_p, c_p, c_l, c_p]),
    "vqw_reduce_ws_bytes": (c_sz, [c_l]),
    "vqw_weighted_sum": (c_i, [c_p, c_p, c_i, c_p, c_p]),
    "vqw_weighted_sum_host": (c_i, [c_p, c_p, c_i, c_p, c_p]),
    "vqw_vq_ws_bytes": (c_sz, [c_l, c_i, c_i]),
    "vqw_vq_plan": (c_i, [c_i, c_i]),
    "vqw_vq_fwd": (c_i, [c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_sz, c_l, c_i, c_i, c_p]),
    "vqw_vq_ema_update": (c_i, [c_p, c_p, c_p, c_p, c_f, c_f, c_f, c_i, c_i, c_p]),
    "vqw_kmeans_update": (c_i, [c_p, c_p, c_p, c_p, c_sz, c_i, c_i, c_p]),
    "vqw_vq_lookup": (c_i, [c_p, c_p, c_p, c_p, c_p, c_l, c_i, c_i, c_p]),
    "vqw_vq_bwd": (c_i, [c_p, c_p, c_p, c_p, c_p, c_l, c_p]),
    "vqw_mask_scale": (c_i, [c_p, c_p, c_p, c_p, c_l, c_p]),
    "vqw_cross_ws_bytes": (c_sz, [c_i, c_i, c_l]),
    "vqw_cross_loss_fwd": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_sz, c_i, c_l, c_i, c_i, c_p]),
    "vqw_cross_loss_bwd": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_l, c_i, c_i, c_p]),
    "vqw_cross_loss_dense_fwd": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_sz, c_i, c_l, c_i, c_i, c_p]),
    "vqw_cross_loss_dense_bwd": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_l, c_i, c_i, c_p]),
    "vqw_codebook_losses": (c_i, [c_p, c_f, c_p, c_p, c_p, c_sz, c_i, c_i, c_p]),
    "vqw_onehot": (c_i, [c_p, c_p, c_i, c_l, c_i, c_p]),
    "vqw_flip_labels": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_p]),
    "vqw_warp_image": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p]),
    "vqw_warp_labels": (c_i, [c_p, c_i, c_p, c_p, c_i, c_i, c_i, c_p]),
    "vqw_photometric": (c_i, [c_p, c_p, c_p, c_p, c_i, c_l, c_p]),
    "vqw_gauss_blur": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    "vqw_sconv_fwd_ws_bytes": (c_sz, [c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i]),
    "vqw_sconv_fwd": (c_i, [c_p, c_p, c_p, c_p, c_p, c_sz, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_p]),
    "vqw_sconv_dgrad_ws_bytes": (c_sz, [c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i]),
    "vqw_sconv_dgrad": (c_i, [c_p, c_p, c_p, c_p, c_sz, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "vqw_sconv_wgrad_ws_bytes": (c_sz, [c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i]),
    "vqw_sconv_wgrad": (c_i, [c_p, c_p, c_p, c_p, c_p, c_sz, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "vqw_leaky_relu_bwd": (c_i, [c_p, c_p, c_p, c_f, c_l, c_p]),
    "vqw_bn_affine_fwd": (c_i, [c_p, c_p, c_p, c_p, c_p, c_l, c_i, c_f, c_p]),
    "vqw_bn_affine_bwd_reduce": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_sz, c_i, c_i, c_i, c_f, c_p]),
    "vqw_bn_affine_bwd_apply": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_d, c_p, c_p, c_p, c_l, c_i, c_f, c_i, c_i, c_p]),
    "vqw_hinge_fwd": (c_i, [c_p, c_l, c_i, c_p, c_p]),
    "vqw_hinge_bwd": (c_i, [c_p, c_l, c_i, c_p, c_p, c_p]),
    "vqw_window_mse_fwd": (c_i, [c_p, c_p, c_p, c_p, c_sz, c_l, c_f, c_f, c_f, c_f, c_p]),
    "vqw_window_mse_bwd": (c_i, [c_p, c_p, c_p, c_p, c_l, c_f, c_f, c_f, c_f, c_p]),
    "vqw_pixel_shuffle2": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    "vqw_dropblock_mask": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p]),
    "vqw_dropblock_apply": (c_i, [c_p, c_p, c_p, c_p, c_l, c_i, c_p]),
    "vqw_seg_ws_bytes": (c_sz, [c_i]),
    "vqw_seg_losses_fwd": (c_i, [c_p, c_p, c_p, c_p, c_p, c_sz, c_i, c_l, c_i, c_i, c_f, c_f, c_f, c_p]),
    "vqw_seg_losses_bwd": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_l, c_i, c_i, c_f, c_f, c_f, c_p]),
    "vqw_fold_defer": (c_i, [c_i]),
    "vqw_fold_pending": (c_i, []),
    "vqw_fold_table_bytes": (c_sz, []),
    "vqw_fold_discard": (c_i, []),
    "vqw_fold_flush_host": (c_i, [c_p, c_p, c_sz, c_p]),
    "vqw_adam_step": (c_i, [c_p, c_p, c_p, c_p, c_l, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_p]),
    "vqw_adam_multi": (c_i, [c_p, c_i, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_p]),
}

_lib = None


ABI_VERSION = 8


def load():
    """Load (once) and return the CDLL with prototypes set.  Raises if the library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libvqwnet_hip.so is not built (%s). Run `make -C medical-image-editing_amd/csrc` "
            "or __graft_entry__.build(); there is no CPU/eager fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError if the header and the library diverge
        fn.restype = res
        fn.argtypes = args
    if lib.vqw_abi_version() != ABI_VERSION:
        raise RuntimeError("libvqwnet_hip.so has ABI %d, the host code expects %d: rebuild it (make -C medical-image-editing_amd/csrc)"
                           % (lib.vqw_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(status, name="vqw"):
    if status != 0:
        msg = load().vqw_last_error()
        raise RuntimeError("%s failed (%d): %s" % (name, status, (msg or b"").decode()))
