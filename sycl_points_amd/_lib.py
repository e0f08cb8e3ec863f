"""ctypes loader for libsycl_points_amd.so (the C ABI declared in include/sycl_points_amd.h).

The product path has no CPU fallback: if the library is missing this module raises, loudly.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# SP_AMD_LIB: another build of the same library (A/B measurements on one box, scratch/ab.sh); the tests and bench.py never set it
LIB_PATH = os.environ.get("SP_AMD_LIB") or os.path.join(_HERE, "lib", "libsycl_points_amd.so")
CSRC = os.path.join(_HERE, "csrc")

SP_OK, SP_ERR_INVALID_ARGUMENT, SP_ERR_RUNTIME, SP_ERR_HIP = 0, 1, 2, 3
COMM_ID_BYTES = 128  # SP_COMM_ID_BYTES


class SpError(RuntimeError):
    """Mirrors the C++ exceptions of the reference: code 1 -> std::invalid_argument, 2 -> std::runtime_error,
    3 -> device error (sycl::exception from wait_and_throw)."""

    def __init__(self, code, msg):
        super().__init__(f"[sycl_points_amd] error {code}: {msg}")
        self.code = code


class Linearized(C.Structure):  # sp_linearized
    _fields_ = [("H", C.c_float * 36), ("b", C.c_float * 6), ("error", C.c_float), ("inlier", C.c_uint32),
                ("inlier_lo", C.c_float), ("inlier_hi", C.c_float), ("pad", C.c_float * 2)]


class FactorParams(C.Structure):  # sp_factor_params
    _fields_ = [("reg_type", C.c_int), ("robust_type", C.c_int), ("max_correspondence_distance", C.c_float),
                ("robust_scale", C.c_float), ("genz_alpha", C.c_float), ("genz_planarity_threshold", C.c_float),
                ("rotation_constraint_enable", C.c_int), ("rotation_constraint_weight", C.c_float),
                ("rotation_robust_scale", C.c_float)]


class DegenerateRegParams(C.Structure):  # sp_degenerate_reg_params
    _fields_ = [("type", C.c_int), ("rot_eigenvalue_threshold", C.c_float), ("trans_eigenvalue_threshold", C.c_float),
                ("base_factor", C.c_float)]


class MapPriorParams(C.Structure):  # sp_map_prior_params
    _fields_ = [("enabled", C.c_int), ("rot_vel_sigma", C.c_float), ("trans_vel_sigma", C.c_float),
                ("rot_base_sigma", C.c_float), ("trans_base_sigma", C.c_float)]


class MapPriorState(C.Structure):  # sp_map_prior_state
    _fields_ = [("has_prior", C.c_int), ("omega", C.c_float * 36), ("T_pred_inv", C.c_float * 16)]


class GnParams(C.Structure):  # sp_gn_params
    _fields_ = [("lambda_", C.c_float), ("crit_rotation", C.c_float), ("crit_translation", C.c_float)]


class OptParams(C.Structure):  # sp_opt_params
    _fields_ = [("method", C.c_int), ("max_iterations", C.c_int), ("crit_rotation", C.c_float), ("crit_translation", C.c_float),
                ("gn_lambda", C.c_float), ("lm_max_inner_iterations", C.c_int), ("lm_lambda_factor", C.c_float),
                ("lm_init_lambda", C.c_float), ("lm_max_lambda", C.c_float), ("lm_min_lambda", C.c_float),
                ("dl_initial_radius", C.c_float), ("dl_min_radius", C.c_float), ("dl_max_radius", C.c_float),
                ("dl_eta1", C.c_float), ("dl_eta2", C.c_float), ("dl_gamma_decrease", C.c_float),
                ("dl_gamma_increase", C.c_float)]


class OptLogEntry(C.Structure):  # sp_opt_log_entry
    _fields_ = [("level", C.c_uint16), ("iteration", C.c_uint16), ("trials", C.c_uint16), ("accepted", C.c_uint16),
                ("damping", C.c_float), ("error", C.c_float)]


OPT_MAX_LEVELS, OPT_LOG_ENTRIES = 8, 64


class AlignResult(C.Structure):  # sp_align_result
    _fields_ = [("T", C.c_float * 16), ("T_lin", C.c_float * 16), ("H", C.c_float * 36), ("b", C.c_float * 6),
                ("error", C.c_float), ("error_raw", C.c_float), ("inlier", C.c_uint32), ("iterations", C.c_uint32),
                ("converged", C.c_uint32), ("status", C.c_uint32), ("linearizations", C.c_uint32), ("trials", C.c_uint32),
                ("searched", C.c_uint32), ("damping", C.c_float), ("log_entries", C.c_uint32), ("pad", C.c_uint32 * 3),
                ("log", OptLogEntry * OPT_LOG_ENTRIES)]


assert C.sizeof(Linearized) == 192
assert C.sizeof(OptParams) == 68 and C.sizeof(OptLogEntry) == 16 and C.sizeof(AlignResult) == 4 * (74 + 14) + 16 * 64

_vp, _sz, _f, _i = C.c_void_p, C.c_size_t, C.c_float, C.c_int

# name -> (restype, argtypes); every symbol include/sycl_points_amd.h declares
SIGNATURES = {
    "sp_abi_version": (_i, []),
    "sp_stream_retired": (None, [_vp]),
    "sp_last_error": (C.c_char_p, []),
    "sp_device_count": (_i, []),
    "sp_set_device": (_i, [_i]),
    "sp_knn_bruteforce_workspace_bytes": (_sz, [_sz, _sz, _sz]),
    "sp_knn_bruteforce_set_pass_a": (_i, [_i]),
    "sp_knn_bruteforce": (_i, [_vp, _sz, _vp, _sz, _sz, _vp, _vp, _vp, _sz, _vp]),
    "sp_kdtree_create": (_i, [_vp, _sz, _sz, _vp, C.POINTER(_vp)]),
    "sp_kdtree_destroy": (None, [_vp]),
    "sp_kdtree_size": (_sz, [_vp]),
    "sp_kdtree_search": (_i, [_vp, _vp, _sz, _sz, _vp, _i, _vp, _vp, _vp]),
    "sp_kdtree_radius_search": (_i, [_vp, _vp, _sz, _sz, _f, _vp, _i, _vp, _vp, _vp]),
    "sp_kdtree_remove_by_flags": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "sp_grid_create": (_i, [_vp, _sz, _f, _f, _vp, C.POINTER(_vp)]),
    "sp_grid_create_adaptive": (_i, [_vp, _sz, _f, _vp, C.POINTER(_vp)]),
    "sp_grid_create_bounded": (_i, [_vp, _sz, _vp, _f, _f, _vp, _vp]),
    "sp_grid_destroy": (None, [_vp]),
    "sp_grid_size": (_sz, [_vp]),
    "sp_grid_order": (_i, [_vp, _vp, _vp]),
    "sp_grid_cell_size": (_f, [_vp]),
    "sp_grid_max_cell_points": (C.c_uint32, [_vp]),
    "sp_grid_search": (_i, [_vp, _vp, _sz, _sz, _vp, _i, _vp, _vp, _vp]),
    "sp_grid_radius_search": (_i, [_vp, _vp, _sz, _sz, _f, _vp, _i, _vp, _vp, _vp]),
    "sp_grid_remove_by_flags": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "sp_grid_self_workspace_bytes": (_sz, [_vp]),
    "sp_grid_self_knn": (_i, [_vp, _sz, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sp_grid_self_knn_range": (_i, [_vp, _sz, _sz, _sz, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sp_grid_gather_rows": (_i, [_vp, _vp, _sz, _sz, _sz, _vp, _vp]),
    "sp_grid_scatter_rows": (_i, [_vp, _vp, _sz, _sz, _sz, _vp, _vp]),
    "sp_cov_estimate": (_i, [_vp, _sz, _vp, _sz, _vp, _vp]),
    "sp_cov_estimate_robust": (_i, [_vp, _sz, _vp, _sz, _i, _f, _f, _sz, _vp, _vp]),
    "sp_cov_normalize": (_i, [_vp, _sz, _vp, _vp]),
    "sp_normals_from_knn": (_i, [_vp, _sz, _vp, _sz, _vp, _vp]),
    "sp_normals_from_cov": (_i, [_vp, _vp, _sz, _vp, _vp]),
    "sp_cov_update_plane": (_i, [_vp, _sz, _vp, _vp]),
    "sp_bvh_create": (_i, [_vp, _sz, _vp, C.POINTER(_vp)]),
    "sp_bvh_destroy": (None, [_vp]),
    "sp_bvh_size": (_sz, [_vp]),
    "sp_bvh_search": (_i, [_vp, _vp, _sz, _sz, _vp, _i, _vp, _vp, _vp]),
    "sp_bvh_self_knn": (_i, [_vp, _sz, _vp, _vp, _vp]),
    "sp_bvh_radius_search": (_i, [_vp, _vp, _sz, _sz, _f, _vp, _i, _vp, _vp, _vp]),
    "sp_bvh_remove_by_flags": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "sp_bvh_export_points": (_i, [_vp, _vp, _vp]),
    "sp_voxel_keys": (_i, [_vp, _sz, _f, _vp, _vp]),
    "sp_voxel_downsample_workspace_bytes": (_sz, [_sz]),
    "sp_voxel_downsample": (_i, [_vp, _sz, _f, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sp_voxel_key_box": (_i, [_vp, _sz, _f, _vp, _vp]),
    "sp_voxel_downsample_boxed": (_i, [_vp, _sz, _f, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sp_voxel_downsample_report": (_i, [_vp, _sz, _f, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sp_transform": (_i, [_vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp]),
    "sp_box_filter_flags": (_i, [_vp, _sz, _f, _f, _vp, _vp]),
    "sp_compact_workspace_bytes": (_sz, [_sz]),
    "sp_compact_by_flags": (_i, [_vp, _sz, _sz, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sp_compact_by_flags_multi": (_i, [_vp, _vp, _vp, _i, _sz, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sp_gather_rows_multi": (_i, [_vp, _vp, _vp, _i, _vp, _sz, _vp]),
    "sp_box_filter_compact_multi": (_i, [_vp, _sz, _f, _f, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sp_gicp_workspace_bytes": (_sz, [_sz]),
    "sp_gicp_linearize": (_i, [_vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _i, C.POINTER(FactorParams), _vp, _vp, _sz, _vp]),
    "sp_gicp_error": (_i, [_vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _i, C.POINTER(FactorParams), _vp, _vp, _sz, _vp]),
    "sp_icp_robust_weights": (_i, [_vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _i, C.POINTER(FactorParams), _vp, _vp]),
    "sp_genz_counts": (_i, [_vp, _vp, _vp, _sz, _f, _f, _vp, _vp]),
    "sp_gicp_target_create": (_i, [_vp, _vp, _sz, _vp, C.POINTER(_vp)]),
    "sp_gicp_target_create_plain": (_i, [_vp, _vp, _sz, _vp, C.POINTER(_vp)]),
    "sp_gicp_target_certify": (_i, [_vp, _vp, _vp]),
    "sp_gicp_target_has_certificates": (_i, [_vp]),
    "sp_gicp_target_update": (_i, [_vp, _vp, _vp]),
    "sp_gicp_target_prepare": (_i, [_vp, _vp, _i, _vp]),
    "sp_gicp_error_prepared": (_i, [_vp, _vp, _vp, _vp, _i, C.POINTER(FactorParams), _vp, _vp, _sz, _vp]),
    "sp_gicp_target_destroy": (None, [_vp]),
    "sp_gicp_source_create": (_i, [_sz, C.POINTER(_vp)]),
    "sp_gicp_source_prepare": (_i, [_vp, _vp, _vp, _vp, _sz, _vp, _i, _i, _vp]),
    "sp_gicp_source_destroy": (None, [_vp]),
    "sp_gicp_iteration_fused": (_i, [_vp, _vp, _vp, _i, C.POINTER(FactorParams), _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sp_gicp_align_optimize": (_i, [_vp, _vp, _vp, C.POINTER(FactorParams), C.POINTER(OptParams), C.POINTER(C.c_float), _i, _vp, _vp,
                                    _sz, _vp]),
    "sp_gicp_source_set_persistent": (_i, [_vp, _i]),
    "sp_gicp_source_set_wave_per_point": (_i, [_vp, _i]),
    "sp_gicp_align_fused": (_i, [_vp, _vp, _vp, C.POINTER(FactorParams), _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sp_gicp_align_step": (_i, [_vp, _vp, _vp, C.POINTER(FactorParams), _vp, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sp_gicp_align_linearization_pose": (_i, [_vp, _i, _vp, _vp]),
    "sp_xchg_create": (_i, [_i, _i, C.POINTER(_vp)]),
    "sp_xchg_handle": (_i, [_vp, _vp]),
    "sp_xchg_connect": (_i, [_vp, _vp]),
    "sp_xchg_set_timeout_ms": (_i, [_vp, C.c_uint]),
    "sp_xchg_rank": (_i, [_vp]),
    "sp_xchg_world": (_i, [_vp]),
    "sp_xchg_destroy": (None, [_vp]),
    "sp_gicp_align_direct": (_i, [_vp, _vp, _vp, C.POINTER(FactorParams), _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sp_gicp_align_status": (_i, [_vp, _i, _vp]),
    "sp_gicp_align_rows": (_vp, [_vp, _i, _vp]),
    "sp_gicp_align_row": (_vp, [_vp, _i, _vp]),
    "sp_comm_unique_id": (_i, [_vp]),
    "sp_comm_create": (_i, [_vp, _i, _i, C.POINTER(_vp)]),
    "sp_comm_destroy": (None, [_vp]),
    "sp_comm_rank": (_i, [_vp]),
    "sp_comm_world": (_i, [_vp]),
    "sp_allreduce_rows": (_i, [_vp, _vp, _i, _vp]),
    "sp_allreduce_f32": (_i, [_vp, _vp, _sz, _vp]),
    "sp_allgather": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "sp_gicp_align_sharded": (_i, [_vp, _vp, _vp, C.POINTER(FactorParams), _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sp_gicp_align_finish": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sp_gn_update": (_i, [_vp, _vp, _f, _f, _f, _vp, _vp]),
    "sp_gn_update_host": (_i, [_vp, _vp, _f, _f, _f, _vp]),
    "sp_se3_exp_host": (None, [_vp, _vp]),
    "sp_rigid_mul_host": (None, [_vp, _vp, _vp]),
    "sp_ldlt6_solve_host": (_i, [_vp, _vp, _vp]),
    "sp_dogleg_step_host": (None, [_vp, _vp, _f, _vp, _vp, _vp]),
    "sp_se3_log_host": (None, [_vp, _vp]),
    "sp_degenerate_regularize_host": (_i, [_vp, _vp, _vp, C.c_uint32, _vp, _vp]),
    "sp_map_prior_update_host": (_i, [_vp, _vp, _f, C.c_uint32, _vp, _vp, _vp]),
    "sp_map_prior_apply_host": (_f, [_vp, _vp, _vp, _vp, _vp]),
    "sp_vhm_create": (_i, [_f, _vp, C.POINTER(_vp)]),
    "sp_vhm_destroy": (None, [_vp]),
    "sp_vhm_set": (_i, [_vp, _i, _f]),
    "sp_vhm_get": (_f, [_vp, _i]),
    "sp_vhm_info": (_sz, [_vp, _i]),
    "sp_vhm_clear": (_i, [_vp, _vp]),
    "sp_vhm_add_point_cloud": (_i, [_vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    "sp_vhm_downsampling": (_i, [_vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _sz, C.POINTER(_sz), _vp]),
    "sp_vhm_overlap_ratio": (_i, [_vp, _vp, _sz, _vp, C.POINTER(_f), _vp]),
    "sp_vhm_remove_old_data": (_i, [_vp, _vp]),
}


# csrc/sp_internal.h: per-handle measurement / tuning switches, exported for tests/, bench.py and scratch/ — not part of the
# C ABI, so not in the table above (tests/test_cabi.py checks that the public header does not mention them)
INTERNAL_SIGNATURES = {
    "sp_internal_source_option": (_i, [_vp, _i, _i]),
    "sp_internal_grid_option": (_i, [_vp, _i, _i]),
    "sp_internal_bvh_option": (_i, [_vp, _i, _i]),
    "sp_internal_grid_small_build": (_i, [_i]),
    "sp_internal_align_searched_log": (_vp, [_vp, _vp]),
    "sp_internal_radix_sort_workspace_bytes": (_sz, [_sz]),
    "sp_internal_radix_sort_u32": (_i, [_vp, _vp, _vp, _vp, _sz, C.c_uint, _vp, _sz, _vp, _vp]),
}
VOXEL_BOX_SHARDS, VOXEL_BOX_SHARD_STRIDE = 16, 32  # SP_VOXEL_BOX_SHARDS, SP_VOXEL_BOX_SHARD_STRIDE
INTERNAL_OPTION = {"stage_mask": 0, "reuse": 1, "fast_nn": 2, "self_knn_mode": 3, "persistent": 4, "persistent_from": 5,
                   "bvh_self_heap": 6, "bvh_sort_queries": 7, "grid_sort_queries": 8, "opt_wave_query": 9, "opt_fuse_trials": 10}


def build(force=False):
    """Compile every HIP source for gfx950 into sycl_points_amd/lib/libsycl_points_amd.so (hipcc cross-compiles
    without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    srcs.append(os.path.join(_HERE, "..", "include", "sycl_points_amd.h"))
    stale = (not os.path.exists(LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", CSRC, "-j8", "-s"])
    return LIB_PATH


_lib = None


def lib():
    """The loaded library with typed entry points. torch must be imported first so that its bundled HIP runtime
    (same SONAME, libamdhip64.so.7) is the one both sides share."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(there is no CPU fallback for the HIP path)")
        import torch  # noqa: F401  (loads libamdhip64 before our library resolves it)

        L = C.CDLL(LIB_PATH)
        for name, (res, args) in list(SIGNATURES.items()) + list(INTERNAL_SIGNATURES.items()):
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        if L.sp_abi_version() != 6:
            raise ImportError("libsycl_points_amd.so ABI version mismatch")
        _lib = L
    return _lib


def check(code):
    if code != SP_OK:
        raise SpError(code, lib().sp_last_error().decode())
