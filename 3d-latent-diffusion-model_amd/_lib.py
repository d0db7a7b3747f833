"""ctypes binding of libldm3d.so (C ABI: include/ldm3d.h).  Fails loudly: there is no fallback path."""
from __future__ import annotations

import ctypes as C
import os
import threading

LDM_MAX_LEVELS = 8
_HERE = os.path.dirname(os.path.abspath(__file__))
# LDM3D_LIB: a diagnostic build of the same sources (tools/kstamps.py: in-kernel stamps); never a different implementation
LIB_PATH = os.environ.get("LDM3D_LIB") or os.path.join(_HERE, "libldm3d.so")


class UNetCfg(C.Structure):
    _fields_ = [("spatial_dims", C.c_int), ("in_channels", C.c_int), ("out_channels", C.c_int),
                ("num_levels", C.c_int),
                ("channels", C.c_int * LDM_MAX_LEVELS), ("attention_levels", C.c_int * LDM_MAX_LEVELS),
                ("num_head_channels", C.c_int * LDM_MAX_LEVELS), ("num_res_blocks", C.c_int * LDM_MAX_LEVELS),
                ("norm_num_groups", C.c_int), ("norm_eps", C.c_float)]


class VaeCfg(C.Structure):
    _fields_ = [("spatial_dims", C.c_int), ("in_channels", C.c_int), ("out_channels", C.c_int),
                ("latent_channels", C.c_int), ("num_levels", C.c_int),
                ("channels", C.c_int * LDM_MAX_LEVELS), ("num_res_blocks", C.c_int * LDM_MAX_LEVELS),
                ("attention_levels", C.c_int * LDM_MAX_LEVELS),
                ("norm_num_groups", C.c_int), ("norm_eps", C.c_float),
                ("with_encoder_nonlocal_attn", C.c_int), ("with_decoder_nonlocal_attn", C.c_int)]


# name -> (restype, argtypes); mirrors include/ldm3d.h one to one (tests/test_abi.py checks the header against it)
_P = C.c_void_p
_F = C.POINTER(C.c_float)
SIGNATURES = {
    "ldm_version": (C.c_int, []),
    "ldm_last_error": (C.c_char_p, []),
    "ldm_unet_create": (C.c_int, [C.POINTER(UNetCfg), C.POINTER(_P)]),
    "ldm_vae_create": (C.c_int, [C.POINTER(VaeCfg), C.POINTER(_P)]),
    "ldm_model_destroy": (None, [_P]),
    "ldm_model_num_params": (C.c_int, [_P]),
    "ldm_model_param_name": (C.c_char_p, [_P, C.c_int]),
    "ldm_model_param_ndim": (C.c_int, [_P, C.c_int]),
    "ldm_model_param_shape": (C.POINTER(C.c_int64), [_P, C.c_int]),
    "ldm_model_load_param": (C.c_int, [_P, C.c_char_p, _P, C.c_size_t]),
    "ldm_model_param_numel_total": (C.c_int64, [_P]),
    "ldm_unet_workspace_bytes": (C.c_size_t, [_P, C.c_int, C.c_int, C.c_int, C.c_int]),
    "ldm_unet_forward": (C.c_int, [_P, _P, C.c_int, _P, C.c_int, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int,
                                   _P, C.c_size_t, _P]),
    "ldm_model_param_offset": (C.c_int64, [_P, C.c_int]),
    "ldm_model_load_params_device": (C.c_int, [_P, C.POINTER(_P), C.c_int, _P]),
    "ldm_model_load_params_flat": (C.c_int, [_P, _P, _P]),
    "ldm_unet_train_workspace_bytes": (C.c_size_t, [_P, C.c_int, C.c_int, C.c_int, C.c_int]),
    "ldm_unet_train_forward": (C.c_int, [_P, _P, C.c_int, _P, C.c_int, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int,
                                         _P, C.c_size_t, _P]),
    "ldm_unet_train_backward": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_size_t, _P]),
    "ldm_vae_train_workspace_bytes": (C.c_size_t, [_P, C.c_int, C.c_int, C.c_int, C.c_int]),
    "ldm_vae_train_forward": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_size_t, _P]),
    "ldm_vae_train_backward": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_size_t, _P]),
    "ldm_grad_sq_norm": (C.c_int, [_P, C.c_int64, _P, _P]),
    "ldm_op_mse_loss": (C.c_int, [_P, _P, C.c_int64, _P, _P, _P]),
    "ldm_adam_step": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, _P,
                                C.c_float, _P]),
    "ldm_model_adam_step": (C.c_int, [_P, _P, _P, _P, _P, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, _P,
                                      C.c_float, _P]),
    "ldm_model_set_graph_mode": (C.c_int, [_P, C.c_int]),
    "ldm_model_set_precision": (C.c_int, [_P, C.c_int]),
    "ldm_model_get_precision": (C.c_int, [_P]),
    "ldm_model_tap_count": (C.c_int, [_P, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "ldm_model_tap_elems": (C.c_int64, [_P, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "ldm_model_tap_info": (C.c_int, [_P, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int,
                                     C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    "ldm_model_taps_workspace_bytes": (C.c_size_t, [_P, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "ldm_unet_forward_taps": (C.c_int, [_P, _P, C.c_int, _P, C.c_int, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P,
                                        _P, C.c_size_t, _P]),
    "ldm_vae_encode_taps": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, C.c_size_t, _P]),
    "ldm_vae_decode_taps": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, C.c_size_t, _P]),
    "ldm_vae_encode_workspace_bytes": (C.c_size_t, [_P, C.c_int, C.c_int, C.c_int, C.c_int]),
    "ldm_vae_decode_workspace_bytes": (C.c_size_t, [_P, C.c_int, C.c_int, C.c_int, C.c_int]),
    "ldm_vae_encode": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_size_t, _P]),
    "ldm_vae_decode": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_size_t, _P]),
    "ldm_ddpm_step": (C.c_int, [_P, _P, _P, _P, _P, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float,
                                C.c_float, C.c_int, _P]),
    "ldm_ddim_step": (C.c_int, [_P, _P, _P, _P, _P, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float,
                                C.c_float, C.c_int, _P]),
    "ldm_add_noise": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int64, _P]),
    "ldm_scale": (C.c_int, [_P, _P, C.c_int64, C.c_float, _P]),
    "ldm_sampler_create": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_uint64, C.POINTER(_P)]),
    "ldm_sampler_destroy": (None, [_P]),
    "ldm_sampler_reset": (C.c_int, [_P, _P, C.c_int, _P]),
    "ldm_sampler_step": (C.c_int, [_P, _P, _P, _P, C.c_int64, _P, C.c_int, _P]),
    "ldm_sampler_noise": (C.c_int, [_P, C.c_int, _P, C.c_int64, _P]),
    "ldm_unet_denoise_step": (C.c_int, [_P, _P, _P, C.c_int, _P, C.c_int, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int,
                                        _P, C.c_size_t, _P]),
    "ldm_op_conv3d": (C.c_int, [_P, C.c_int, _P, C.c_int, _P, _P, _P, C.c_int, _P, C.c_int, _P, _P, _P, C.c_int, _P, _P, _P,
                                C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                C.c_int, C.c_int, _P, C.c_size_t, _P]),
    "ldm_op_weight_flip_transpose": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "ldm_op_conv3d_wgrad": (C.c_int, [_P, C.c_int, _P, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "ldm_op_group_norm_bwd_scratch_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "ldm_op_group_norm_bwd": (C.c_int, [_P, _P, C.c_int, _P, C.c_int, _P, _P, C.c_int, C.c_float, C.c_int, _P, _P, _P, _P, _P, _P,
                                        C.c_int, C.c_int, _P, C.c_size_t, _P]),
    "ldm_op_scale_intensity_percentiles_scratch_bytes": (C.c_size_t, [C.c_int]),
    "ldm_op_scale_intensity_percentiles": (C.c_int, [_P, _P, C.c_int, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, _P, C.c_size_t, _P]),
    "ldm_op_im2col": (C.c_int, [_P, _P] + [C.c_int] * 10 + [_P]),
    "ldm_op_col2im": (C.c_int, [_P, _P] + [C.c_int] * 10 + [_P]),
    "ldm_op_leaky_relu": (C.c_int, [_P, _P, C.c_int64, C.c_float, _P]),
    "ldm_op_leaky_relu_bwd": (C.c_int, [_P, _P, _P, C.c_int64, C.c_float, _P]),
    "ldm_op_pack_ncdhw": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int64, _P]),
    "ldm_op_unpack_ndhwc": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int64, _P]),
    "ldm_op_conv3d_gn_scratch_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "ldm_op_conv3d_gn": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, C.c_int, C.c_float, C.c_int, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_size_t, _P]),
    "ldm_op_conv3d_fin_gn_scratch_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "ldm_op_conv3d_fin_gn": (C.c_int, [_P, C.c_int, _P, _P, _P, C.c_int, _P, _P, _P, C.c_int, C.c_float, C.c_int, _P, _P,
                                       C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_size_t, _P]),
    "ldm_op_group_norm_scratch_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "ldm_op_group_norm": (C.c_int, [_P, C.c_int, _P, C.c_int, _P, _P, C.c_int, C.c_float, C.c_int, _P, C.c_int, C.c_int,
                                    _P, C.c_size_t, _P]),
    "ldm_op_attention": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P]),
    "ldm_op_attention_hd": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "ldm_op_attention_bwd_hd": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "ldm_op_attention_train": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, _P]),
    "ldm_op_attention_bwd": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P]),
    "ldm_profile_start": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "ldm_profile_detail": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int]),
    "ldm_profile_stop": (C.c_int, [C.POINTER(C.c_double)]),
    "ldm_set_plan_trace": (C.c_int, [C.c_char_p]),
    "ldm_op_pack_ncdhw_f32": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int64, _P]),
    "ldm_op_unpack_ndhwc_f32": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int64, _P]),
    "ldm_op_im2col_f32": (C.c_int, [_P, _P] + [C.c_int] * 10 + [_P]),
    "ldm_op_col2im_f32": (C.c_int, [_P, _P] + [C.c_int] * 10 + [_P]),
    "ldm_op_leaky_relu_f32": (C.c_int, [_P, _P, C.c_int64, C.c_float, _P]),
    "ldm_op_leaky_relu_bwd_f32": (C.c_int, [_P, _P, _P, C.c_int64, C.c_float, _P]),
    "ldm_op_gemm_f32": (C.c_int, [_P, C.c_int, _P, _P, _P, C.c_int64, C.c_int, C.c_int, C.c_int, _P]),
    "ldm_debug_conv_block_slots": (C.c_int, [C.c_int]),
    "ldm_op_conv3d_block_stats_rows": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "ldm_op_conv3d_block": (C.c_int, [_P, C.c_int, _P, _P, _P, C.c_int, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "ldm_op_conv3d_block128": (C.c_int, [_P, C.c_int, _P, _P, _P, C.c_int, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "ldm_op_linear_f32x3": (C.c_int, [_P, C.c_int, _P, C.c_int, _P, _P, _P, _P, _P, C.c_int64, C.c_int, C.c_int, C.c_int, _P]),
    "ldm_op_gemm_wgrad_f32": (C.c_int, [_P, C.c_int, _P, C.c_int, _P, C.c_int, C.c_int64, C.c_int, _P]),
    "ldm_op_group_norm_f32_scratch_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "ldm_op_group_norm_f32": (C.c_int, [_P, C.c_int, _P, _P, C.c_int, C.c_float, C.c_int, _P, C.c_int, C.c_int, _P, C.c_size_t, _P]),
    "ldm_op_group_norm_bwd_f32": (C.c_int, [_P, _P, C.c_int, _P, _P, C.c_int, C.c_float, C.c_int, _P, _P, _P, C.c_int, C.c_int, _P, C.c_size_t, _P]),
    "ldm_debug_kstamps": (C.c_int, [C.POINTER(C.c_uint64), C.c_int, C.c_int]),
    "ldm_model_plan_conv_cfgs": (C.c_int, [_P, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int]),
    "ldm_comm_unique_id": (C.c_int, [C.c_char_p]),
    "ldm_comm_init": (C.c_int, [C.c_int, C.c_int, C.c_char_p, C.POINTER(_P)]),
    "ldm_comm_init_custom": (C.c_int, [C.c_int, C.c_int, _P, _P, C.POINTER(_P)]),
    "ldm_comm_allreduce": (C.c_int, [_P, _P, C.c_int64, C.c_int, C.c_int, _P]),
    "ldm_comm_broadcast": (C.c_int, [_P, _P, C.c_int64, C.c_int, C.c_int, _P]),
    "ldm_comm_barrier": (C.c_int, [_P, _P]),
    "ldm_comm_destroy": (None, [_P]),
    "ldm_comm_rank": (C.c_int, [_P]),
    "ldm_comm_world": (C.c_int, [_P]),
    "ldm_model_set_grad_sync": (C.c_int, [_P, _P]),
    "ldm_model_set_grad_wire": (C.c_int, [_P, C.c_int]),
    "ldm_model_grad_sync_pending": (C.c_int, [_P]),
    "ldm_comm_stats": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "ldm_comm_is_rccl": (C.c_int, [_P]),
    "ldm_comm_rccl_version": (C.c_int, []),
    "ldm_model_plan_launches": (C.c_int, [_P, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "ldm_model_grad_sync_trace": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int]),
    "ldm_model_grad_schedule": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int64),
                                          C.POINTER(C.c_int64), C.POINTER(C.c_int), C.c_int]),
}
# ldm_allreduce_fn (include/ldm3d.h): int fn(void* user, void* buf, int64_t count, int dtype, int op, void* stream)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, _P, _P, C.c_int64, C.c_int, C.c_int, _P)

_lib = None
_lock = threading.Lock()


class LdmError(RuntimeError):
    """Raised for every non-zero ldm_status (message from ldm_last_error)."""


def lib() -> C.CDLL:
    """Load libldm3d.so once.  Raises (never degrades) if it has not been built."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise LdmError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
                                   f"g.build()'` or `make -C {os.path.join(_HERE, 'csrc')}` (no CPU fallback exists)")
                h = C.CDLL(LIB_PATH)
                for name, (res, args) in SIGNATURES.items():
                    fn = getattr(h, name)            # AttributeError = ABI mismatch: fail loudly
                    fn.restype = res
                    fn.argtypes = args
                _lib = h
    return _lib


def check(status: int) -> None:
    if status != 0:
        msg = lib().ldm_last_error()
        raise LdmError(f"libldm3d status {status}: {msg.decode() if msg else '?'}")


def ptr(t) -> int | None:
    """Device/host address of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def current_stream() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream
