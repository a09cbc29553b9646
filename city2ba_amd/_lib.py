"""ctypes binding of include/city2ba_hip.h (the C-ABI shared library built from csrc/).

There is NO CPU fallback: if the HIP library is missing this module raises at import of the
first symbol, and every compute entry point fails with C2B_ERR_NO_DEVICE when no GPU is visible.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libcity2ba_hip.so")

OK = 0
ERR_INVALID_ARGUMENT = -1
ERR_INDEX_OUT_OF_RANGE = -2
ERR_HIP = -3
ERR_OOM = -4
ERR_NO_DEVICE = -5
ERR_RCCL = -6
COMM_ID_BYTES = 128
ABI_VERSION = 6            # include/city2ba_hip.h: C2B_ABI_VERSION
CAMBLK_DOUBLES = 32
STATS_DOUBLES = 20

_vp = C.c_void_p
_i64 = C.c_int64
_u64 = C.c_uint64
_d = C.c_double
_int = C.c_int

# name -> (restype, argtypes).  Kept in one table so tests can check every symbol the header declares.
SIGNATURES = {
    "c2b_version": (C.c_char_p, []),
    "c2b_abi_version": (_int, []),
    "c2b_last_error": (C.c_char_p, []),
    "c2b_device_count": (_int, [C.POINTER(_int)]),
    "c2b_workspace_bytes": (_i64, [_i64]),
    "c2b_workspace_init": (_int, [_vp, _vp]),
    "c2b_workspace_selfcheck": (_int, [_vp, _vp, C.POINTER(_i64)]),
    "c2b_cameras_from_bal": (_int, [_vp, _i64, _vp, _vp]),
    "c2b_cameras_to_bal": (_int, [_vp, _i64, _vp, _vp]),
    "c2b_camblk_doubles": (_i64, [_i64]),
    "c2b_camblk_from_state": (_int, [_vp, _i64, _vp, _i64, _vp, _vp]),
    "c2b_camblk_from_bal": (_int, [_vp, _i64, _vp, _i64, _vp, _vp]),
    "c2b_cameras_from_position_direction": (_int, [_vp, _vp, _i64, _vp, _vp]),
    "c2b_project_world": (_int, [_vp, _vp, _vp, _i64, _vp, _vp]),
    "c2b_to_world": (_int, [_vp, _vp, _vp, _i64, _vp, _vp]),
    "c2b_cameras_transform": (_int, [_vp, _vp, _vp, _i64, _vp]),
    "c2b_points_pad": (_int, [_vp, _i64, _vp, _vp]),
    "c2b_points_unpad": (_int, [_vp, _i64, _vp, _vp]),
    "c2b_expand_rows": (_int, [_vp, _i64, _i64, _i64, _vp, _vp]),
    "c2b_project": (_int, [_vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    "c2b_reprojection_error_sum": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _d, _vp, _vp, _vp]),
    "c2b_rows_tiles_bytes": (_i64, [_i64]),
    "c2b_rows_pack": (_int, [_vp, _i64, _i64, _vp, _vp]),
    "c2b_project_rows": (_int, [_vp, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp]),
    "c2b_reprojection_error_sum_rows": (_int, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _d, _vp, _vp, _vp]),
    "c2b_visibility_rows": (_int, [_vp, _vp, _vp, _i64, _vp, _vp, _i64, _d, _vp, _vp, _vp]),
    "c2b_visibility_rows_bits": (_int, [_vp, _vp, _vp, _i64, _vp, _vp, _i64, _d, _vp, _vp, _vp]),
    "c2b_reprojection_error_sums2_rows": (_int, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "c2b_add_noise_observations_error_sums2_rows": (_int, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _i64, _d, _u64, _vp, _vp, _vp]),
    "c2b_jacobian_stream_policy": (_int, [_i64, _i64, _i64]),
    "c2b_jacobian_tiles_per_wave": (_int, [_i64]),
    "c2b_jacobian_launch_shape": (_int, [_i64, _d, _vp, _vp]),
    "c2b_jacobian_outputs_store_rate": (_int, [_vp, _vp]),
    "c2b_jacobian_outputs_set_store_rate": (_int, [_vp, _d]),
    "c2b_residual_jacobian_rows_placed": (_int, [_vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _d, _vp, _vp, _vp]),
    "c2b_residual_jacobian_rows": (_int, [_vp, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _d, _vp, _vp, _vp]),
    "c2b_residual_jacobian": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _d, _vp, _vp]),
    "c2b_error_sum_finish": (_int, [_vp, _i64, _vp, _vp]),
    "c2b_residual_jacobian_sum": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _d, _vp, _vp, _vp]),
    "c2b_host_alloc": (_int, [C.POINTER(_vp), _i64]),
    "c2b_host_free": (None, [_vp]),
    "c2b_jacobian_outputs_alloc": (_int, [_i64, _int, _d, _vp, C.POINTER(_vp)]),
    "c2b_jacobian_outputs_pointers": (_int, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)]),
    "c2b_jacobian_outputs_log": (_int, [_vp, _vp, _int, C.POINTER(_int), C.POINTER(_int)]),
    "c2b_jacobian_outputs_free": (None, [_vp]),
    "c2b_calib_store_pattern": (_int, [_i64, _vp, _vp, _vp, _vp]),
    "c2b_calib_copy": (_int, [_vp, _vp, _i64, _vp]),
    "c2b_visibility_pairs": (_int, [_vp, _vp, _vp, _vp, _i64, _d, _vp, _vp, _vp]),
    "c2b_visibility_dense_tiles": (_i64, [_i64]),
    "c2b_visibility_dense_count": (_int, [_vp, _i64, _vp, _i64, _d, _vp, _vp, _vp, _vp]),
    "c2b_visibility_dense_fill": (_int, [_vp, _i64, _vp, _i64, _d, _vp, _vp, _vp, _vp, _vp]),
    "c2b_occlusion_filter": (_int, [_vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _vp]),
    "c2b_bvh_build": (_int, [_vp, _i64, C.POINTER(_vp)]),
    "c2b_bvh_sizes": (_int, [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_int)]),
    "c2b_bvh_copy": (_int, [_vp, _vp, _vp, _vp]),
    "c2b_bvh_free": (None, [_vp]),
    "c2b_occlusion_filter_bvh": (_int, [_vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp]),
    "c2b_stats": (_int, [_vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp]),
    "c2b_stats_partial_pass1": (_int, [_vp, _vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _vp, _vp]),
    "c2b_stats_partial_pass2": (_int, [_vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp]),
    "c2b_stats_combine_shares": (_int, [_vp, _int, _vp]),
    "c2b_stats_finish_shares": (_int, [_vp, _int, _i64, _vp]),
    "c2b_comm_backend": (C.c_char_p, []),
    "c2b_comm_unique_id": (_int, [_vp]),
    "c2b_comm_init_rank": (_int, [_vp, _int, _int, _int, C.POINTER(_vp)]),
    "c2b_comm_init_all": (_int, [_int, _vp, _vp]),
    "c2b_comm_group_start": (_int, []),
    "c2b_comm_group_end": (_int, []),
    "c2b_comm_info": (_int, [_vp, C.POINTER(_int), C.POINTER(_int), C.POINTER(_int)]),
    "c2b_comm_all_reduce_sum_f64": (_int, [_vp, _vp, _i64, _vp]),
    "c2b_comm_all_gather_f64": (_int, [_vp, _vp, _i64, _vp, _vp]),
    "c2b_comm_destroy": (None, [_vp]),
    "c2b_stats_sharded": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _i64, _vp, _vp, _vp]),
    "c2b_add_drift_sharded": (_int, [_vp, _i64, _i64, _vp, _i64, _vp, _int, _d, _d, _d, _d, _d, _d, _u64, _vp]),
    "c2b_add_noise_entities_sharded": (_int, [_vp, _i64, _i64, _vp, _i64, _vp, _d, _d, _d, _u64, _vp]),
    "c2b_add_drift": (_int, [_vp, _i64, _vp, _i64, _vp, _d, _d, _d, _d, _d, _d, _u64, _vp]),
    "c2b_add_drift_normalized": (_int, [_vp, _i64, _vp, _i64, _vp, _d, _d, _d, _u64, _vp]),
    "c2b_add_noise_entities": (_int, [_vp, _i64, _vp, _i64, _vp, _d, _d, _d, _u64, _vp]),
    "c2b_add_noise_observations": (_int, [_vp, _i64, _i64, _d, _u64, _vp]),
    "c2b_add_sin_noise": (_int, [_vp, _i64, _vp, _i64, _vp, _d, _d, _d, _d, _d, _d, _d, _d, _vp]),
    "c2b_convert_f64_to_f32": (_int, [_vp, _i64, _vp, _vp]),
    "c2b_convert_f32_to_f64": (_int, [_vp, _i64, _vp, _vp]),
    "c2b_stats_f32": (_int, [_vp, _i64, _vp, _i64, _vp, _vp, _vp]),
    "c2b_add_drift_f32": (_int, [_vp, _i64, _vp, _i64, _vp, _d, _d, _d, _d, _d, _d, _u64, _vp]),
    "c2b_add_drift_normalized_f32": (_int, [_vp, _i64, _vp, _i64, _vp, _d, _d, _d, _u64, _vp]),
    "c2b_add_noise_entities_f32": (_int, [_vp, _i64, _vp, _i64, _vp, _d, _d, _d, _u64, _vp]),
    "c2b_add_sin_noise_f32": (_int, [_vp, _i64, _vp, _i64, _vp, _d, _d, _d, _d, _d, _d, _d, _d, _vp]),
    "c2b_partition_cameras": (_int, [_vp, _i64, _int, _vp]),
    "c2b_synthetic_grid_sizes": (_int, [_i64, _i64, _i64, C.POINTER(_i64), C.POINTER(_i64)]),
    "c2b_synthetic_grid_layout": (_int, [_i64, _i64, _i64, _d, _d, _d, _d, _vp, _vp, _vp]),
    "c2b_synthetic_line_layout": (_int, [_i64, _i64, _d, _d, _d, _d, _vp, _vp, _vp]),
    "c2b_candidate_pairs": (_int, [_vp, _i64, _vp, _i64, _d, _i64, _i64, _int, _d, _d, _int, C.POINTER(_vp)]),
    "c2b_pairs_count": (_i64, [_vp]),
    "c2b_pairs_cam_idx": (_vp, [_vp]),
    "c2b_pairs_pt_idx": (_vp, [_vp]),
    "c2b_pairs_free": (None, [_vp]),
    "c2b_obj_load": (_int, [C.c_char_p, C.POINTER(_vp)]),
    "c2b_obj_model_count": (_i64, [_vp]),
    "c2b_obj_model_name": (C.c_char_p, [_vp, _i64]),
    "c2b_obj_model_sizes": (_int, [_vp, _i64, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_int)]),
    "c2b_obj_model_copy": (_int, [_vp, _i64, _vp, _vp]),
    "c2b_obj_move_to_origin": (_int, [_vp, _i64]),
    "c2b_obj_triangles": (_int, [_vp, _i64, _vp, C.POINTER(_i64)]),
    "c2b_obj_free": (None, [_vp]),
    "c2b_generate_cameras_path": (_int, [_vp, _i64, _i64, _d, _u64, _vp, _vp]),
    "c2b_generate_cameras_poisson": (_int, [_vp, _i64, _i64, _d, _d, _u64, _i64, _vp, _vp, C.POINTER(_i64)]),
    "c2b_generate_cameras_poisson_bvh": (_int, [_vp, _i64, _vp, _i64, _d, _d, _u64, _i64, _vp, _vp, C.POINTER(_i64)]),
    "c2b_modify_intrinsics": (_int, [_vp, _i64, _vp, _vp, _u64]),
    "c2b_generate_world_points": (_int, [_vp, _i64, _vp, _i64, _i64, _d, _u64, _vp, C.POINTER(_i64)]),
    "c2b_cull": (_int, [C.POINTER(_i64), _vp, _int, C.POINTER(_i64), _vp, _vp, _vp, _vp, _int]),
    "c2b_add_incorrect_correspondences": (_int, [_i64, _vp, _vp, _vp, _d, _u64]),
    "c2b_drop_features": (_int, [_i64, _vp, _vp, _vp, _d, _u64]),
    "c2b_split_landmarks": (_int, [C.POINTER(_i64), _vp, _i64, _i64, _vp, _d, _u64]),
    "c2b_join_landmarks": (_int, [_i64, _vp, _i64, _vp, _d, _u64]),
    "c2b_bal_read": (_int, [C.c_char_p, C.POINTER(_vp)]),
    "c2b_bal_sizes": (_int, [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    "c2b_bal_copy": (_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "c2b_bal_close": (None, [_vp]),
    "c2b_bal_write": (_int, [C.c_char_p, _i64, _vp, _i64, _vp, _vp, _vp, _vp]),
    "c2b_bal_read_as": (_int, [C.c_char_p, _int, C.POINTER(_vp)]),
    "c2b_bal_write_as": (_int, [C.c_char_p, _int, _i64, _vp, _i64, _vp, _vp, _vp, _vp]),
    "c2b_format_f64": (_int, [_i64, _vp, _vp, _i64, C.POINTER(_i64)]),
    "c2b_parse_f64": (_int, [C.c_char_p, _i64, _i64, _vp, _vp]),
    "c2b_ply_write": (_int, [C.c_char_p, _i64, _vp, _i64, _vp, _vp, _vp]),
    "c2b_problem_create": (_int, [_int, C.POINTER(_vp)]),
    "c2b_problem_destroy": (None, [_vp]),
    "c2b_problem_options_init": (None, [_vp]),
    "c2b_problem_set_options": (_int, [_vp, _vp]),
    "c2b_problem_get_options": (_int, [_vp, _vp]),
    "c2b_host_set_io_threads": (None, [_int]),
    "c2b_problem_upload": (_int, [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp]),
    "c2b_problem_upload_bal": (_int, [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp]),
    "c2b_problem_synthetic_grid_layout": (_int, [_vp, _i64, _i64, _i64, _d, _d, _d, _d]),
    "c2b_problem_synthetic_line_layout": (_int, [_vp, _i64, _i64, _d, _d, _d, _d]),
    "c2b_problem_sizes": (_int, [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    "c2b_problem_download": (_int, [_vp, _vp, _vp, _vp]),
    "c2b_problem_download_bal": (_int, [_vp, _vp]),
    "c2b_problem_write": (_int, [_vp, C.c_char_p, _int]),
    "c2b_problem_read": (_int, [_vp, C.c_char_p, _int]),
    "c2b_problem_from_position_direction": (_int, [_vp, _i64, _vp, _vp, _vp]),
    "c2b_problem_centers": (_int, [_vp, _vp]),
    "c2b_problem_project": (_int, [_vp, _vp]),
    "c2b_problem_total_reprojection_error": (_int, [_vp, _d, C.POINTER(_d)]),
    "c2b_problem_total_reprojection_error_sharded": (_int, [_vp, _vp, _d, C.POINTER(_d)]),
    "c2b_problem_total_reprojection_errors_l1_l2": (_int, [_vp, C.POINTER(_d), C.POINTER(_d)]),
    "c2b_problem_total_reprojection_errors_l1_l2_sharded": (_int, [_vp, _vp, C.POINTER(_d), C.POINTER(_d)]),
    "c2b_problem_add_noise_errors_l1_l2": (_int, [_vp, _d, _d, _d, _d, _u64, C.POINTER(_d), C.POINTER(_d)]),
    "c2b_problem_add_noise_errors_l1_l2_sharded": (_int, [_vp, _vp, _d, _d, _d, _d, _u64, C.POINTER(_d), C.POINTER(_d)]),
    "c2b_problem_residual_jacobian": (_int, [_vp, _vp, _vp, _vp]),
    "c2b_problem_residual_jacobian_device": (_int, [_vp, _int, C.POINTER(_vp), C.POINTER(_d)]),
    "c2b_problem_stats": (_int, [_vp, _vp]),
    "c2b_problem_visibility_pairs": (_int, [_vp, _i64, _vp, _vp, _d, _vp, _vp]),
    "c2b_problem_cull": (_int, [_vp, _int]),
    "c2b_problem_largest_connected_component": (_int, [_vp, _int]),
    "c2b_problem_remove_singletons": (_int, [_vp]),
    "c2b_largest_connected_component": (_int, [C.POINTER(_i64), _vp, _int, C.POINTER(_i64), _vp, _vp, _vp, _vp, _int]),
    "c2b_remove_singletons": (_int, [C.POINTER(_i64), _vp, _int, C.POINTER(_i64), _vp, _vp, _vp, _vp]),
    "c2b_problem_adopt_visibility": (_int, [_vp]),
    "c2b_problem_download_graph": (_int, [_vp, _vp, _vp]),
    "c2b_problem_export_device": (_int, [_vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "c2b_problem_visibility_pairs_compact": (_int, [_vp, _i64, _vp, _vp, _d, _vp]),
    "c2b_problem_visibility_within_distance": (_int, [_vp, _d, _int, _d, _d, _vp]),
    "c2b_problem_visibility_dense": (_int, [_vp, _d, _vp]),
    "c2b_problem_visibility_dense_fetch": (_int, [_vp, _vp, _vp]),
    "c2b_problem_visibility_dense_occlude": (_int, [_vp, _vp, _i64, _vp]),
    "c2b_problem_visibility_dense_occlude_bvh": (_int, [_vp, _vp, _vp]),
    "c2b_problem_generate_world_points": (_int, [_vp, _vp, _i64, _i64, _d, _u64, C.POINTER(_i64)]),
    "c2b_problem_add_drift": (_int, [_vp, _d, _d, _d, _vp, _u64]),
    "c2b_problem_add_drift_normalized": (_int, [_vp, _d, _d, _d, _u64]),
    "c2b_problem_add_noise": (_int, [_vp, _d, _d, _d, _d, _u64]),
    "c2b_problem_add_sin_noise": (_int, [_vp, _vp, _vp, _d, _d]),
    "c2b_problem_set_shard": (_int, [_vp, _i64, _i64, _i64]),
    "c2b_problem_stats_sharded": (_int, [_vp, _vp, _vp]),
    "c2b_problem_add_drift_sharded": (_int, [_vp, _vp, _d, _d, _d, _vp, _u64]),
    "c2b_problem_add_noise_sharded": (_int, [_vp, _vp, _d, _d, _d, _d, _u64]),
    "c2b_problem_add_sin_noise_sharded": (_int, [_vp, _vp, _vp, _vp, _d, _d]),
}


class City2baError(RuntimeError):
    """Mirrors the reference's failure modes: bad arguments / the asserts of
    src/baproblem.rs:345-346, 365-369 (status -2) / device errors."""

    def __init__(self, status, message):
        super().__init__("city2ba_hip status %d: %s" % (status, message))
        self.status = status


_lib = None


def _share_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64.  libcity2ba_hip.so links ROCm's; with
    both loaded a process holds two HIP runtimes and whichever initialises second can fail ("no ROCm-capable
    device", "No HIP GPUs are available") depending on import order.  When torch is installed its runtime is
    therefore opened first, globally, so that this library binds to the same copy torch will use.  Hosts without
    torch (the Rust binding) simply get ROCm's runtime."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    libdir = os.path.join(os.path.dirname(spec.origin), "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError:
                pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "city2ba_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
        _share_torch_hip_runtime()
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(L, name)          # AttributeError here = header/library mismatch
            f.restype = res
            f.argtypes = args
        if L.c2b_abi_version() != ABI_VERSION:       # the number this table of signatures was written against (C2B_ABI_VERSION)
            raise ImportError("city2ba_amd: %s speaks ABI %d, this binding %d -- rebuild the library" % (LIB_PATH, L.c2b_abi_version(), ABI_VERSION))
        _lib = L
    return _lib


def check(status):
    if status != OK:
        raise City2baError(status, lib().c2b_last_error().decode("utf-8", "replace"))


def device_count():
    n = _int(0)
    rc = lib().c2b_device_count(C.byref(n))
    return n.value if rc == OK else 0
