"""ctypes view of the C ABI in include/smhip.h.  No compute happens in Python."""
from __future__ import annotations

import ctypes
import sys
import os

from . import build as _build

c_double_p = ctypes.POINTER(ctypes.c_double)
c_float_p = ctypes.POINTER(ctypes.c_float)
c_int32_p = ctypes.POINTER(ctypes.c_int32)


class IcpOptions(ctypes.Structure):
    _fields_ = [("max_iteration", ctypes.c_int32), ("dist_outlier_ratio", ctypes.c_float),
                ("early_exit", ctypes.c_int32), ("nn_mode", ctypes.c_int32), ("grid_cell", ctypes.c_float),
                ("grid_max_ring", ctypes.c_int32), ("check_every", ctypes.c_int32),
                ("use_ball", ctypes.c_int32), ("exact_matches", ctypes.c_int32), ("ball_radius", ctypes.c_float),
                ("ball_cap_factor", ctypes.c_float), ("no_certify", ctypes.c_int32), ("no_lds_table", ctypes.c_int32),
                ("no_overlap", ctypes.c_int32), ("overlap_streams", ctypes.c_int32),
                ("split_after", ctypes.c_int32), ("nn_epsilon", ctypes.c_float), ("no_fused_sums", ctypes.c_int32), ("no_single_kernel", ctypes.c_int32)]


class IcpStats(ctypes.Structure):
    _fields_ = [("iterations", ctypes.c_int32), ("kept", ctypes.c_int32), ("limit_d2", ctypes.c_double),
                ("fallback_queries", ctypes.c_int32), ("status", ctypes.c_int32), ("hard_queries", ctypes.c_int32),
                ("refined_iterations", ctypes.c_int32), ("searched_queries", ctypes.c_int32), ("fused_iterations", ctypes.c_int32)]


class IcpProfile(ctypes.Structure):
    _fields_ = [("ms_prepare", ctypes.c_double), ("ms_find_closests", ctypes.c_double),
                ("ms_error_elements", ctypes.c_double), ("ms_solve", ctypes.c_double),
                ("launches_find_closests", ctypes.c_int32), ("launches_error_elements", ctypes.c_int32),
                ("launches_solve", ctypes.c_int32), ("launches_nn_main", ctypes.c_int32),
                ("ms_nn_main", ctypes.c_double), ("pairs_nn_main", ctypes.c_double),
                ("ms_nn_certify", ctypes.c_double), ("pairs_nn_certify", ctypes.c_double),
                ("launches_nn_certify", ctypes.c_int32), ("split_after_used", ctypes.c_int32),
                ("ms_nn_listed", ctypes.c_double), ("pairs_nn_listed", ctypes.c_double), ("ms_nn_refine", ctypes.c_double),
                ("pairs_error_elements", ctypes.c_double), ("launches_nn_listed", ctypes.c_int32), ("launches_nn_refine", ctypes.c_int32)]


class NdtOptions(ctypes.Structure):
    _fields_ = [("resolution", ctypes.c_float), ("step_size", ctypes.c_float), ("outlier_ratio", ctypes.c_float),
                ("transformation_epsilon", ctypes.c_float), ("max_iterations", ctypes.c_int32),
                ("min_points_per_voxel", ctypes.c_int32), ("min_covar_eigvalue_mult", ctypes.c_float),
                ("reserved", ctypes.c_int32 * 5)]


class NdtStats(ctypes.Structure):
    _fields_ = [("iterations", ctypes.c_int32), ("derivative_calls", ctypes.c_int32), ("voxels", ctypes.c_int32),
                ("status", ctypes.c_int32), ("trans_probability", ctypes.c_double), ("pairs_last", ctypes.c_double)]


class NdtGicpOptions(ctypes.Structure):
    _fields_ = [("voxel_resolution", ctypes.c_float), ("using_voxel_filter", ctypes.c_int32), ("use_ndt", ctypes.c_int32),
                ("ndt_transformation_epsilon", ctypes.c_float), ("ndt_step_size", ctypes.c_float),
                ("ndt_resolution", ctypes.c_float), ("ndt_max_iterations", ctypes.c_int32),
                ("gicp_max_iterations", ctypes.c_int32), ("gicp_rotation_epsilon", ctypes.c_double),
                ("gicp_transformation_epsilon", ctypes.c_double), ("gicp_epsilon", ctypes.c_double),
                ("gicp_corr_dist_threshold", ctypes.c_double), ("gicp_max_inner_iterations", ctypes.c_int32),
                ("gicp_k_correspondences", ctypes.c_int32), ("gicp_search_cell", ctypes.c_float),
                ("reserved", ctypes.c_int32 * 3)]


class NdtGicpStats(ctypes.Structure):
    _fields_ = [("ok", ctypes.c_int32), ("n_source", ctypes.c_int32), ("n_target", ctypes.c_int32),
                ("ndt_iterations", ctypes.c_int32), ("gicp_iterations", ctypes.c_int32),
                ("gicp_function_evaluations", ctypes.c_int32), ("gicp_correspondences", ctypes.c_int32),
                ("reserved", ctypes.c_int32), ("ndt_score", ctypes.c_double), ("gicp_score", ctypes.c_double)]


class FilterDesc(ctypes.Structure):
    _fields_ = [("type", ctypes.c_int32), ("axis_index", ctypes.c_int32), ("seed", ctypes.c_uint32),
                ("reserved", ctypes.c_int32), ("p", ctypes.c_float * 6)]


class MrvmSettings(ctypes.Structure):
    _fields_ = [("prob_threshold", ctypes.c_float), ("high_resolution", ctypes.c_float), ("hit_prob", ctypes.c_float),
                ("miss_prob", ctypes.c_float), ("z_offset", ctypes.c_float), ("max_point_num_in_cell", ctypes.c_int32),
                ("use_max_intensity", ctypes.c_int32), ("reserved", ctypes.c_int32)]


# name -> (restype, argtypes): every symbol include/smhip.h declares
SIGNATURES = {
    "smhip_version": (ctypes.c_int, []),
    "smhip_status_string": (ctypes.c_char_p, [ctypes.c_int]),
    "smhip_device_count": (ctypes.c_int, []),
    "smhip_create": (ctypes.c_int, [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                    ctypes.POINTER(ctypes.c_void_p)]),
    "smhip_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "smhip_last_error": (ctypes.c_char_p, [ctypes.c_void_p]),
    "smhip_icp_default_options": (None, [ctypes.POINTER(IcpOptions)]),
    "smhip_icp_set_options": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(IcpOptions)]),
    "smhip_synchronize": (ctypes.c_int, [ctypes.c_void_p]),
    "smhip_set_source_f64": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, c_double_p, ctypes.c_int]),
    "smhip_set_target_f64": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, c_double_p, c_double_p, ctypes.c_int]),
    "smhip_set_source_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, c_float_p, ctypes.c_int, ctypes.c_int]),
    "smhip_reserve_batch_workspaces": (ctypes.c_int, [ctypes.c_void_p]),
    "smhip_set_sources_f32_batch": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, c_int32_p, ctypes.POINTER(c_float_p), c_int32_p]),
    "smhip_set_target_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, c_float_p, ctypes.c_int, c_float_p,
                                            ctypes.c_int, ctypes.c_int]),
    "smhip_copy_slot": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]),
    "smhip_icp_align": (ctypes.c_int, [ctypes.c_void_p, c_double_p, c_double_p, c_double_p, ctypes.POINTER(IcpStats)]),
    "smhip_icp_align_batch": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, c_double_p, c_double_p, c_double_p,
                                             ctypes.POINTER(IcpStats)]),
    "smhip_icp_align_range": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, c_double_p, c_double_p, c_double_p,
                                             ctypes.POINTER(IcpStats)]),
    "smhip_icp_trimmed_score": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, c_double_p, ctypes.c_float, c_double_p, c_int32_p]),
    "smhip_prepare_target_from_target": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, c_int32_p]),
    "smhip_sample_source": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_uint32, c_int32_p]),
    "smhip_set_target_cache": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "smhip_icp_forget_search_history": (ctypes.c_int, [ctypes.c_void_p]),
    "smhip_icp_single_launch_counts": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]),
    "smhip_get_capacity": (ctypes.c_int, [ctypes.c_void_p, c_int32_p, c_int32_p, c_int32_p]),
    "smhip_get_cloud_sizes": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, c_int32_p, c_int32_p, c_int32_p]),
    "smhip_icp_enqueue_batch": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, c_double_p]),
    "smhip_icp_fetch_batch": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, c_double_p, c_double_p,
                                             ctypes.POINTER(IcpStats)]),
    "smhip_icp_export_results_device": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    "smhip_calculate_normals_f64": (ctypes.c_int, [c_double_p, ctypes.c_int, c_double_p, c_double_p, c_int32_p]),
    "smhip_prepare_target_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, c_float_p, ctypes.c_int, ctypes.c_int, c_int32_p]),
    "smhip_prepare_target_from_source": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, c_int32_p]),
    "smhip_prepare_targets_from_sources": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, c_int32_p, c_int32_p, c_int32_p]),
    "smhip_get_target_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, c_float_p, c_float_p, ctypes.c_int]),
    "smhip_icp_get_matches": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, c_int32_p, c_float_p, ctypes.c_int]),
    "smhip_icp_find_closests": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, c_double_p, c_int32_p, c_float_p,
                                               ctypes.c_int]),
    "smhip_ndt_default_options": (None, [ctypes.POINTER(NdtOptions)]),
    "smhip_ndt_set_options": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(NdtOptions)]),
    "smhip_ndt_align": (ctypes.c_int, [ctypes.c_void_p, c_double_p, c_double_p, c_double_p, ctypes.POINTER(NdtStats)]),
    "smhip_ndt_align_batch": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, c_double_p, c_double_p, c_double_p, ctypes.POINTER(NdtStats)]),
    "smhip_ndt_build_voxels": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]),
    "smhip_ndt_get_voxels": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, c_int32_p, c_int32_p, c_double_p, c_float_p, c_float_p]),
    "smhip_ndt_compute_derivatives": (ctypes.c_int, [ctypes.c_void_p, c_double_p, ctypes.c_int, c_double_p, c_double_p, c_double_p]),
    "smhip_ndt_time_derivatives": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_double_p, c_double_p]),
    "smhip_ndt_gicp_default_options": (None, [ctypes.POINTER(NdtGicpOptions)]),
    "smhip_ndt_gicp_set_options": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(NdtGicpOptions)]),
    "smhip_ndt_gicp_set_source_f32": (ctypes.c_int, [ctypes.c_void_p, c_float_p, ctypes.c_int, ctypes.c_int]),
    "smhip_ndt_gicp_set_target_f32": (ctypes.c_int, [ctypes.c_void_p, c_float_p, ctypes.c_int, ctypes.c_int]),
    "smhip_ndt_gicp_align": (ctypes.c_int, [ctypes.c_void_p, c_double_p, c_double_p, c_double_p, ctypes.POINTER(NdtGicpStats)]),
    "smhip_ndt_gicp_jobs": (ctypes.c_int, [ctypes.c_void_p]),
    "smhip_ndt_gicp_set_source_f32_job": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, c_float_p, ctypes.c_int, ctypes.c_int]),
    "smhip_ndt_gicp_set_target_f32_job": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, c_float_p, ctypes.c_int, ctypes.c_int]),
    "smhip_ndt_gicp_align_batch": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, c_double_p, c_double_p, c_double_p, ctypes.POINTER(NdtGicpStats)]),
    "smhip_gicp_align": (ctypes.c_int, [ctypes.c_void_p, c_double_p, c_double_p, c_double_p, ctypes.POINTER(NdtGicpStats)]),
    "smhip_ndt_gicp_get_downsampled": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, c_float_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]),
    "smhip_gicp_evaluate": (ctypes.c_int, [ctypes.c_void_p, c_double_p, c_double_p, c_double_p, c_double_p]),
    "smhip_gicp_get_covariances": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, c_double_p, ctypes.c_int]),
    "smhip_filter_default": (None, [ctypes.c_int, ctypes.POINTER(FilterDesc)]),
    "smhip_filter_config_valid": (ctypes.c_int, [ctypes.POINTER(FilterDesc)]),
    "smhip_filter_chain_f32": (ctypes.c_int, [ctypes.c_void_p, c_float_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(FilterDesc),
                                              ctypes.c_int, ctypes.POINTER(ctypes.c_int)]),
    "smhip_filter_get_output": (ctypes.c_int, [ctypes.c_void_p, c_float_p, c_int32_p, ctypes.c_int]),
    "smhip_filter_output_to_source": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "smhip_mrvm_default_settings": (None, [ctypes.POINTER(MrvmSettings)]),
    "smhip_mrvm_create": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(MrvmSettings), ctypes.POINTER(ctypes.c_void_p)]),
    "smhip_mrvm_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "smhip_mrvm_last_error": (ctypes.c_char_p, [ctypes.c_void_p]),
    "smhip_mrvm_set_offset_z": (None, [ctypes.c_void_p, ctypes.c_float]),
    "smhip_mrvm_insert_f32": (ctypes.c_int, [ctypes.c_void_p, c_float_p, ctypes.c_int, ctypes.c_int, c_float_p]),
    "smhip_mrvm_voxel_count": (ctypes.c_int, [ctypes.c_void_p, c_int32_p]),
    "smhip_mrvm_set_max_table_log2": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "smhip_mrvm_table_log2": (ctypes.c_int, [ctypes.c_void_p]),
    "smhip_mrvm_output": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_float, c_float_p, ctypes.c_int, c_int32_p]),
    "smhip_mrvm_output_ex": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_float, ctypes.c_int, c_float_p, ctypes.c_int, c_int32_p]),
    "smhip_mrvm_last_skipped": (ctypes.c_int, [ctypes.c_void_p, c_int32_p]),
    "smhip_mrvm_dump": (ctypes.c_int, [ctypes.c_void_p, c_int32_p, ctypes.POINTER(ctypes.c_uint8), c_int32_p, c_int32_p, c_float_p, ctypes.c_int, c_int32_p]),
    "smhip_icp_enable_profile": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "smhip_icp_get_profile": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(IcpProfile)]),
    "smhip_icp_get_search_counts": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_uint32)]),
}

_LIB = None


def library_path() -> str:
    return _build.LIB_PATH


def load_library():
    """dlopen staticmapping_amd/lib/libsmhip.so.  Fails loudly: there is no CPU fallback."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: the HIP extension has not been built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
            "There is deliberately no CPU fallback for the registration hot path.")
    # (One process, one HIP runtime: PyTorch-ROCm ships its own libamdhip64 / libhsa-runtime64 under the same SONAMEs.  A
    # process that uses both must import torch BEFORE the first call into this package -- bench.py and tests/conftest.py do --
    # or torch's later initialisation finds "No HIP GPUs are available"; tools/runtime_order_probe.py shows both orders.)
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)        # AttributeError = header/library drift, also loud
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib
