"""ctypes binding of libsonet_hip.so (the C ABI declared in include/sonet_hip.h).

There is no CPU fallback: if the library is missing, cannot be loaded, or the current device is not
a gfx950, every op raises.  PyTorch is used only for device memory and streams.
"""
import ctypes
import os

import torch  # noqa: F401  (loads libamdhip64.so first so the library binds to torch's HIP runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libsonet_hip.so")
# The VARIANTS build (make -C so-net_amd/csrc variants: tuning / ablation knobs read from SONET_* environment variables, measured-slower
# kernels kept as tested records) is never loaded unless asked for: SONET_HIP_LIB=<path> (tools/, tests/variants).
VARIANTS_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libsonet_hip_variants.so")
if os.environ.get("SONET_HIP_LIB"):
    LIB_PATH = os.environ["SONET_HIP_LIB"]

_vp = ctypes.c_void_p
_i = ctypes.c_int

# name -> argtypes; every function returns int status unless listed in _RESTYPES
SIGNATURES = {
    "sonet_abi_version": [],
    "sonet_build_arch": [],
    "sonet_last_error": [],
    "sonet_check_device": [],
    "sonet_range_log_set": [_vp],
    "sonet_pooled_wgrad_f32": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp],
    "sonet_bn_running_update_f32": [_vp, _vp, _vp, _vp, ctypes.c_float, ctypes.c_float, _i, _vp],
    "sonet_knn_prepare_f32": [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "sonet_pointmlp_h3_gather_f32": [_vp, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _vp],
    "sonet_planes_max_f32": [_vp, _vp, ctypes.c_longlong, _i, _i, _vp],
    "sonet_diag_mfma_f16_rate": [_i, _i, _vp, _vp, _vp],
    "sonet_index_max_f32": [_vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "sonet_index_max_bf16": [_vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "sonet_index_max_gather_f32": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "sonet_som_assign_f32": [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "sonet_som_assign_sort_ws_size": [_i, _i, _i, _i],
    "sonet_pointmlp_bf16_pool": [_vp, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "sonet_pointmlp_bf16_pool_xaff": [_vp, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp],
    "sonet_pointmlp_h3_segpool_ws_size": [_i, _i, _i],
    "sonet_pointmlp_h3_segpool_f32": [_vp, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp],
    "sonet_pointmlp_h3_stats_xaff_f32": [_vp, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp],
    "sonet_pointmlp_x3_bnb_f32": [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp],
    "sonet_pointmlp_x3_bnb_acc_f32": [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _vp],
    "sonet_wgrad_x3_xaff_f32": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp],
    "sonet_pooled_wgrad_xaff_f32": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp],
    "sonet_pack_multi": [_vp, _i, _i, _vp],
    "sonet_pack_multi_kc": [_i, _i],
    "sonet_bn_rider_set": [_vp, _vp, ctypes.c_float, ctypes.c_float, ctypes.c_float, _vp, _vp, _vp, _vp, _vp],
    "sonet_som_assign_sort_f32": [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sonet_som_assign_sort_det_f32": [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sonet_som_assign_sort_knn_f32": [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                      _vp, _i, _i, _i, _vp, _vp, _vp, _vp],
    "sonet_som_group_f32": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "sonet_som_mask_i32": [_vp, _vp, _i, _i, _i, _vp],
    "sonet_node_gather_f32": [_vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "sonet_knn_self_f32": [_vp, _vp, _i, _i, _i, _vp],
    "sonet_knn_group_f32": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp],
    "sonet_lastdim_max_f32": [_vp, _vp, ctypes.c_longlong, _i, _vp],
    "sonet_knn_gather_f32": [_vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "sonet_pointmlp_pack_size": [_i, _i],
    "sonet_pointmlp_pack_f32": [_vp, _vp, _i, _i, _vp],
    "sonet_pointmlp_f32": [_vp, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _vp],
    "sonet_pointmlp_bf16_pack_size": [_i, _i],
    "sonet_pointmlp_bf16_pack": [_vp, _vp, _i, _i, _vp],
    "sonet_pointmlp_bf16": [_vp, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _vp],
    "sonet_pointmlp_bf16_acc": [_vp, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _vp],
    "sonet_pointmlp_bf16_bnb": [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _vp],
    "sonet_pointmlp_bf16_gather": [_vp, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _vp],
    "sonet_index_max_gather_bf16": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "sonet_index_max_gather_p16": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "sonet_node_gather_lead_affine_act_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp],
    "sonet_node_gather_lead_affine_act_bf16": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp],
    "sonet_pointmlp_stats_ws_size": [_i, _i, _i],
    "sonet_pointmlp_h3_stats_f32": [_vp, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp],
    "sonet_pointmlp_x3_stats_f32": [_vp, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp],
    "sonet_pointmlp_bf16_stats_ws_size": [_i, _i, _i],
    "sonet_pointmlp_bf16_stats": [_vp, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp],
    "sonet_pointmlp_bf16_stats_xaff": [_vp, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp],
    "sonet_pointmlp_h3_nodeadd_f32": [_vp, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _vp, _vp, _i, _vp],
    "sonet_wgrad_x3_ws_size": [_i, _i, _i, _i],
    "sonet_wgrad_x3_f32": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "sonet_wgrad_bf16_ws_size": [_i, _i, _i, _i],
    "sonet_wgrad_bf16": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "sonet_wgrad_bf16_xaff": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp],
    "sonet_pointmlp_x3_pack_size": [_i, _i],
    "sonet_pointmlp_x3_pack": [_vp, _vp, _i, _i, _vp],
    "sonet_pointmlp_x3_f32": [_vp, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _vp],
    "sonet_pointmlp_h3_pack": [_vp, _vp, _i, _i, _vp],
    "sonet_pointmlp_h3_f32": [_vp, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _vp],
    "sonet_lastdim_argmax_f32": [_vp, _vp, _vp, ctypes.c_longlong, _i, _vp],
    "sonet_lastdim_argmax_bf16": [_vp, _vp, _vp, ctypes.c_longlong, _i, _vp],
    "sonet_lastdim_max_bwd_f32": [_vp, _vp, _vp, ctypes.c_longlong, _i, _vp],
    "sonet_lastdim_max_bwd_bf16": [_vp, _vp, _vp, ctypes.c_longlong, _i, _vp],
    "sonet_knn_gather_bwd_ws_size": [_i, _i, _i],
    "sonet_knn_gather_bwd_f32": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "sonet_knn_gather_bwd_bf16": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "sonet_p16_size": [_i, _i, _i],
    "sonet_p16_from_f32": [_vp, _vp, _i, _i, _i, _vp, _vp, _i, _vp],
    "sonet_p16_to_f32": [_vp, _vp, _i, _i, _i, _vp],
    "sonet_pointmlp_h3p_pack_size": [_i, _i],
    "sonet_pointmlp_h3p_pack": [_vp, _vp, _i, _i, _vp],
    "sonet_pointmlp_h3p_stats_ws_size": [_i, _i, _i],
    "sonet_pointmlp_h3p": [_vp, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp],
    "sonet_pointresnet_bf16_pack_size": [],
    "sonet_pointresnet_bf16_pack": [_vp, _vp, _vp, _vp, _i, _vp, _vp],
    "sonet_pointresnet_bf16": [_vp, _i, _vp, _vp, _vp, _i, _i, _vp],
    "sonet_pointresnet_bf16_pool_ws_size": [_i, _i, _i],
    "sonet_pointresnet_bf16_pool": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "sonet_pointresnet_pack_size": [],
    "sonet_pointresnet_pack": [_vp, _vp, _vp, _vp, _i, _vp, _vp],
    "sonet_pointresnet_fused_f32": [_vp, _i, _vp, _vp, _vp, _i, _i, _vp],
    "sonet_pointresnet_fused_p16_f32": [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "sonet_pointresnet_pool_ws_size": [_i, _i, _i],
    "sonet_pointresnet_fused_pool_f32": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "sonet_pointresnet_fused_pool_p16_f32": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "sonet_knn_stage_columns": [_i, _i, _i],
    "sonet_knn_stage_prepare_f32": [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "sonet_knn_stage_input_p16": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp],
    "sonet_pointmlp_h3p_gmax": [_vp, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp],
    "sonet_node_stage_columns": [_i, _i],
    "sonet_p16_flat_to_bcm_f32": [_vp, _vp, _i, _i, _i, _vp],
    "sonet_som_sort_group_f32": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sonet_pointwise_bwd_stats_f32": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "sonet_pointwise_bwd_apply_f32": [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "sonet_node_add_affine_act_f32": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "sonet_linear_act_f32": [_vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _vp],
    "sonet_pooled_dgrad_ws_size": [_i, _i, _i, _i],
    "sonet_pooled_wgrad_xbf16": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp],
    "sonet_pooled_wgrad_xaff_xbf16": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp],
    "sonet_pooled_dgrad_obf16": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "sonet_pooled_dgrad_f32": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "sonet_pooled_dgrad_mfma_bf16": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "sonet_bn_fwd_coeffs_f32": [_vp, _vp, _vp, _vp, ctypes.c_float, _i, _vp, _vp, _vp, _vp],
    "sonet_bn_bwd_coeffs_f32": [_vp, _vp, _vp, _vp, ctypes.c_double, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "sonet_channel_affine_act_out_f32": [_vp, _vp, _vp, _i, _vp, _i, _i, _i, _vp],
    "sonet_chunk_mean_f32": [_vp, _vp, ctypes.c_longlong, _i, _i, _vp],
    "sonet_pointmlp_x3_pack_strided": [_vp, ctypes.c_longlong, ctypes.c_longlong, _vp, _i, _i, _i, _vp],
    "sonet_pointmlp_h3_pack_strided": [_vp, ctypes.c_longlong, ctypes.c_longlong, _vp, _i, _i, _i, _vp],
    "sonet_pointmlp_bf16_pack_strided": [_vp, ctypes.c_longlong, ctypes.c_longlong, _vp, _i, _i, _i, _vp],
    "sonet_adam_chunk": [],
    "sonet_fc_max_rows": [],
    "sonet_fc_bn_act_fwd_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_float, ctypes.c_float, _i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "sonet_fc_bn_act_bwd_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "sonet_fc_dx_f32": [_vp, _vp, _i, _i, _i, _vp, _vp],
    "sonet_adam_multi_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _i, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, _vp],
    "sonet_channel_stats_f32": [_vp, _i, _i, _i, _vp, _vp, _vp, _vp],
    "sonet_channel_stats_bf16": [_vp, _i, _i, _i, _vp, _vp, _vp, _vp],
    "sonet_channel_affine_act_out_bf16": [_vp, _vp, _vp, _i, _vp, _i, _i, _i, _vp],
    "sonet_pointwise_bwd_stats_bf16": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "sonet_pointwise_bwd_apply_bf16": [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "sonet_channel_affine_act_f32": [_vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "sonet_chamfer_nn_f32": [_vp, _vp, _vp, _i, _i, _i, _vp],
}
_RESTYPES = {
    "sonet_build_arch": ctypes.c_char_p,
    "sonet_last_error": ctypes.c_char_p,
    "sonet_pointmlp_pack_size": ctypes.c_size_t,
    "sonet_pointmlp_stats_ws_size": ctypes.c_size_t,
    "sonet_pointmlp_h3_segpool_ws_size": ctypes.c_size_t,
    "sonet_pointmlp_bf16_stats_ws_size": ctypes.c_size_t,
    "sonet_wgrad_x3_ws_size": ctypes.c_size_t,
    "sonet_wgrad_bf16_ws_size": ctypes.c_size_t,
    "sonet_pointmlp_x3_pack_size": ctypes.c_size_t,
    "sonet_p16_size": ctypes.c_size_t,
    "sonet_knn_stage_columns": ctypes.c_size_t,
    "sonet_node_stage_columns": ctypes.c_size_t,
    "sonet_knn_gather_bwd_ws_size": ctypes.c_size_t,
    "sonet_pointmlp_h3p_pack_size": ctypes.c_size_t,
    "sonet_pointmlp_h3p_stats_ws_size": ctypes.c_size_t,
    "sonet_pointmlp_bf16_pack_size": ctypes.c_size_t,
    "sonet_pointresnet_pack_size": ctypes.c_size_t,
    "sonet_pointresnet_bf16_pack_size": ctypes.c_size_t,
    "sonet_pointresnet_bf16_pool_ws_size": ctypes.c_size_t,
    "sonet_pointresnet_pool_ws_size": ctypes.c_size_t,
    "sonet_pooled_dgrad_ws_size": ctypes.c_size_t,
    "sonet_pooled_dgrad_tail_ws_size": ctypes.c_size_t,
    "sonet_som_assign_sort_ws_size": ctypes.c_size_t,
    "sonet_chamfer_nn2_ws_size": ctypes.c_size_t,
}

# entry points that only the variants build exports (bound when present)
VARIANT_SIGNATURES = {
    "sonet_chamfer_nn2_ws_size": [_i, _i, _i],
    "sonet_chamfer_nn2_f32": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "sonet_pointmlp_h3_kmax_f32": [_vp, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp],
    "sonet_pooled_dgrad_tail_ws_size": [_i, _i, _i],
    "sonet_pooled_dgrad_tail_f32": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp],
}

_lib = None


class SonetHipError(RuntimeError):
    pass


def load():
    """Load the library (once).  Raises SonetHipError when it is not built / not loadable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SonetHipError(
            "libsonet_hip.so is not built (%s).  Build it with `make -C so-net_amd/csrc` or "
            "`python -c 'import __graft_entry__ as g; g.build()'`.  There is no CPU fallback." % LIB_PATH)
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise SonetHipError("cannot load %s: %s" % (LIB_PATH, e))
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header / library mismatch
        fn.argtypes = args
        fn.restype = _RESTYPES.get(name, ctypes.c_int)
    for name, args in VARIANT_SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.argtypes = args
            fn.restype = _RESTYPES.get(name, ctypes.c_int)
    _lib = lib
    return lib


def last_error():
    return load().sonet_last_error().decode("utf-8", "replace")


def check(status, what=""):
    if status != 0:
        raise SonetHipError("%s failed (status %d): %s" % (what or "libsonet_hip", status, last_error()))


_device_ok = {}


def require_device(device):
    """The ops run only on an MI355X: fail loudly otherwise.  (Checked once per device: torch.cuda.is_available() alone costs 2-3 us per
    call -- an environment lookup -- and an op wrapper runs a hundred times per training step.)"""
    if device.index in _device_ok:
        return
    if not torch.cuda.is_available():
        raise SonetHipError("sonet_hip needs an MI355X (gfx950) GPU; torch.cuda.is_available() is False "
                            "and there is no CPU fallback")
    idx = device.index if device.index is not None else torch.cuda.current_device()
    ok = _device_ok.get(idx)
    if ok is None:
        with torch.cuda.device(idx):
            check(load().sonet_check_device(), "sonet_check_device")
        _device_ok[idx] = True


def stream_ptr():
    """The current HIP stream of the current device (what torch.cuda.current_stream().cuda_stream returns, without building a Stream
    object: 10 us -> 0.4 us, 100 calls per training step)."""
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


class on_device:
    """``with torch.cuda.device(dev)`` for a ``torch.device`` that is known to be a CUDA device, without the argument parsing
    (4-5 us per use, 50 uses per training step)."""
    __slots__ = ("idx", "prev")

    def __init__(self, dev):
        self.idx = dev.index if dev.index is not None else -1

    def __enter__(self):
        self.prev = torch.cuda._exchange_device(self.idx)

    def __exit__(self, *exc):
        torch.cuda._maybe_exchange_device(self.prev)
        return False


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None
