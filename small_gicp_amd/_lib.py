"""ctypes binding of libsmall_gicp_amd.so (the C-ABI declared in include/small_gicp_amd.h).

The product path has NO CPU fallback: if the library is missing or there is no gfx950 device, calls raise.
"""
import ctypes as C
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SGA_LIB_PATH") or os.path.join(_HERE, "lib", "libsmall_gicp_amd.so")

SGA_OK = 0
ICP, PLANE_ICP, GICP = 0, 1, 2
ROBUST_NONE, ROBUST_HUBER, ROBUST_CAUCHY = 0, 1, 2
LEVENBERG_MARQUARDT, GAUSS_NEWTON = 0, 1
MATH_FP32, MATH_FP64 = 0, 1
ACCUM_DOUBLES = 30


class SgaError(RuntimeError):
    pass


class FactorParams(C.Structure):
    _fields_ = [("factor_kind", C.c_int), ("robust_kind", C.c_int), ("robust_c", C.c_double), ("max_dist_sq", C.c_double), ("math_mode", C.c_int)]


class RegistrationSettingC(C.Structure):
    _fields_ = [
        ("factor", FactorParams),
        ("optimizer", C.c_int),
        ("max_iterations", C.c_int),
        ("max_inner_iterations", C.c_int),
        ("init_lambda", C.c_double),
        ("lambda_factor", C.c_double),
        ("gn_lambda", C.c_double),
        ("translation_eps", C.c_double),
        ("rotation_eps", C.c_double),
        ("verbose", C.c_int),
        ("restrict_dof_lambda", C.c_double),
        ("restrict_dof_mask", C.c_double * 6),
    ]


class ResultC(C.Structure):
    _fields_ = [
        ("T_target_source", C.c_double * 16),
        ("converged", C.c_int),
        ("iterations", C.c_uint64),
        ("num_inliers", C.c_uint64),
        ("H", C.c_double * 36),
        ("b", C.c_double * 6),
        ("error", C.c_double),
    ]


REJECTOR_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_size_t, C.POINTER(C.c_int64), C.POINTER(C.c_float), C.POINTER(C.c_ubyte))
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_size_t)  # sga_allreduce_fn
LINEARIZE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint64))
ERROR_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double))

# every symbol include/small_gicp_amd.h and include/small_gicp_amd_debug.h declare: (name, restype, argtypes)
_vp, _dp, _fp = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_float)
_pvp = C.POINTER(C.c_void_p)
SYMBOLS = [
    ("sga_last_error", C.c_char_p, []),
    ("sga_version", C.c_char_p, []),
    ("sga_device_count", C.c_int, []),
    ("sga_allocator_stats", None, [C.POINTER(C.c_uint64)]),
    ("sga_context_create", C.c_int, [C.c_int, _pvp]),
    ("sga_context_create_on_stream", C.c_int, [C.c_int, _vp, _pvp]),
    ("sga_context_destroy", C.c_int, [_vp]),
    ("sga_context_synchronize", C.c_int, [_vp]),
    ("sga_context_stream", _vp, [_vp]),
    ("sga_cloud_create_f32", C.c_int, [_vp, _fp, _fp, _fp, C.c_size_t, _pvp]),
    ("sga_cloud_create_f64", C.c_int, [_vp, _dp, _dp, _dp, C.c_size_t, _pvp]),
    ("sga_cloud_create_f32_origin", C.c_int, [_vp, _fp, _fp, _fp, C.c_size_t, _dp, _pvp]),
    ("sga_cloud_create_f64_origin", C.c_int, [_vp, _dp, _dp, _dp, C.c_size_t, _dp, _pvp]),
    ("sga_cloud_origin", C.c_int, [_vp, _dp]),
    ("sga_index_origin", C.c_int, [_vp, _dp]),
    ("sga_choose_origin", None, [_dp, _dp, _dp]),
    ("sga_cloud_slice", C.c_int, [_vp, _vp, C.c_size_t, C.c_size_t, _pvp]),
    ("sga_cloud_destroy", C.c_int, [_vp]),
    ("sga_cloud_size", C.c_int, [_vp, C.POINTER(C.c_size_t)]),
    ("sga_cloud_has", C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("sga_cloud_download", C.c_int, [_vp, _vp, _fp, _fp, _fp]),
    ("sga_cloud_download_f64", C.c_int, [_vp, _vp, _dp, _fp, _fp]),
    ("sga_voxelgrid_sampling", C.c_int, [_vp, _vp, C.c_double, _pvp]),
    ("sga_estimate_normals_covariances", C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int]),
    ("sga_index_build_kdtree", C.c_int, [_vp, _vp, _pvp]),
    ("sga_index_build_gaussian_voxelmap", C.c_int, [_vp, _vp, C.c_double, _pvp]),
    ("sga_index_create_voxelmap_from_voxels", C.c_int, [_vp, C.c_double, C.c_void_p, _dp, _dp, C.c_size_t, _pvp]),
    ("sga_index_create_flatmap_from_voxels", C.c_int, [_vp, C.c_double, C.c_void_p, C.c_void_p, _dp, _dp, C.c_int, C.c_size_t, _pvp]),
    ("sga_index_refresh_attributes", C.c_int, [_vp, _vp, _vp]),
    ("sga_index_clone", C.c_int, [_vp, _vp, _pvp]),
    ("sga_index_destroy", C.c_int, [_vp]),
    ("sga_index_size", C.c_int, [_vp, C.POINTER(C.c_size_t)]),
    ("sga_index_voxelmap_download", C.c_int, [_vp, _vp, C.POINTER(C.c_int32), _fp, _fp, C.POINTER(C.c_uint32)]),
    ("sga_voxelmap_create", C.c_int, [_vp, C.c_double, _pvp]),
    ("sga_voxelmap_insert", C.c_int, [_vp, _vp, _vp, _dp]),
    ("sga_voxelmap_set_lru", C.c_int, [_vp, C.c_uint32, C.c_uint32]),
    ("sga_flatmap_create", C.c_int, [_vp, C.c_double, _pvp]),
    ("sga_flatmap_set_setting", C.c_int, [_vp, C.c_double, C.c_uint32]),
    ("sga_voxelmap_set_search_offsets", C.c_int, [_vp, C.c_int]),
    ("sga_flatmap_download", C.c_int, [_vp, _vp, C.POINTER(C.c_int32), C.POINTER(C.c_uint32), _fp, _fp]),
    ("sga_index_knn", C.c_int, [_vp, _vp, _fp, C.c_size_t, C.c_int, C.c_double, C.POINTER(C.c_int64), _fp]),
    ("sga_index_knn_f64", C.c_int, [_vp, _vp, _dp, C.c_size_t, C.c_int, C.c_double, C.POINTER(C.c_int64), _dp]),
    ("sga_factor_params_default", None, [C.POINTER(FactorParams)]),
    ("sga_problem_create", C.c_int, [_vp, _vp, _vp, _dp, _pvp]),
    ("sga_problem_create_from_index", C.c_int, [_vp, _vp, _vp, _dp, _pvp]),
    ("sga_problem_destroy", C.c_int, [_vp]),
    ("sga_linearize", C.c_int, [_vp, _vp, C.POINTER(FactorParams), _dp, _dp, _dp, _dp, C.POINTER(C.c_uint64)]),
    ("sga_error", C.c_int, [_vp, _vp, C.POINTER(FactorParams), _dp, _dp]),
    ("sga_linearize_async", C.c_int, [_vp, _vp, C.POINTER(FactorParams), _dp, _vp]),
    ("sga_error_async", C.c_int, [_vp, _vp, C.POINTER(FactorParams), _dp, _vp]),
    ("sga_comm_unique_id", C.c_int, [C.POINTER(C.c_ubyte)]),
    ("sga_comm_init", C.c_int, [_vp, C.c_int, C.c_int, C.POINTER(C.c_ubyte)]),
    ("sga_comm_destroy", C.c_int, [_vp]),
    ("sga_comm_init_callback", C.c_int, [_vp, C.c_int, C.c_int, ALLREDUCE_FN, _vp]),
    ("sga_error_model_eval", C.c_int, [_dp, _dp, _dp, _dp]),
    ("sga_unpack_accumulator", None, [_dp, _dp, _dp, _dp, C.POINTER(C.c_uint64)]),
    ("sga_problem_set_rejector", C.c_int, [_vp, REJECTOR_FN, _vp]),
    ("sga_linearize_per_point", C.c_int, [_vp, _vp, C.POINTER(FactorParams), _dp, _dp, C.POINTER(C.c_ubyte)]),
    ("sga_problem_get_factors", C.c_int, [_vp, _vp, C.POINTER(C.c_int64), _fp]),
    ("sga_context_set_stream_ordered", C.c_int, [_vp, C.c_int]),
    ("sga_context_set_profiling", C.c_int, [_vp, C.c_int]),
    ("sga_context_get_kernel_ms", C.c_int, [_vp, _dp, C.POINTER(C.c_uint64), _dp, C.POINTER(C.c_uint64)]),
    ("sga_context_get_comm_ms", C.c_int, [_vp, _dp, C.POINTER(C.c_uint64)]),
    ("sga_context_get_search_ms", C.c_int, [_vp, _dp, C.POINTER(C.c_uint64)]),
    ("sga_context_get_pass_ms", C.c_int, [_vp, _dp, C.POINTER(C.c_uint64), _dp, C.POINTER(C.c_uint64), _dp]),
    ("sga_set_warm_limit", None, [C.c_double]),
    ("sga_set_error_model", None, [C.c_int]),
    ("sga_problem_set_search_stats", C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    ("sga_problem_get_search_stats", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    ("sga_problem_get_sorted_points", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    ("sga_problem_get_grid_stats", C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    ("sga_set_grid_mode", None, [C.c_int, C.c_longlong]),
    ("sga_host_alloc", C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    ("sga_host_free", C.c_int, [C.c_void_p]),
    ("sga_index_spacing", C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    ("sga_set_knn_wave_max", None, [C.c_longlong]),
    ("sga_debug_timer_start", C.c_int, [C.c_void_p]),
    ("sga_debug_timer_stop", C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    ("sga_debug_shard_frame_pack", None, [C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    ("sga_debug_shard_frame_agree", C.c_int, [C.POINTER(C.c_double)]),
    ("sga_debug_kd_trips", C.c_int, [C.c_void_p]),
    ("sga_debug_kd_wave_times", C.c_int, [C.c_void_p, C.c_int]),
    ("sga_set_search_mode", None, [C.c_int, C.c_int, C.c_int]),
    ("sga_get_warm_limit", C.c_double, []),
    ("sga_problem_get_pass_stats", C.c_int, [_vp, _vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("sga_registration_setting_default", None, [C.POINTER(RegistrationSettingC)]),
    ("sga_align", C.c_int, [_vp, _vp, _vp, _dp, C.POINTER(RegistrationSettingC), C.POINTER(ResultC)]),
    ("sga_align_problem", C.c_int, [_vp, _vp, _dp, C.POINTER(RegistrationSettingC), C.POINTER(ResultC)]),
    ("sga_multi_create", C.c_int, [C.POINTER(C.c_int), C.c_int, _pvp]),
    ("sga_multi_destroy", C.c_int, [_vp]),
    ("sga_multi_num_devices", C.c_int, [_vp]),
    ("sga_multi_set_target_f64", C.c_int, [_vp, _dp, _dp, _dp, C.c_size_t]),
    ("sga_multi_set_target_f32", C.c_int, [_vp, _fp, _fp, _fp, C.c_size_t]),
    ("sga_multi_set_source_f32", C.c_int, [_vp, _fp, _fp, _fp, C.c_size_t, _dp]),
    ("sga_multi_set_target_f32_origin", C.c_int, [_vp, _fp, _fp, _fp, C.c_size_t, _dp]),
    ("sga_multi_set_source_f32_origin", C.c_int, [_vp, _fp, _fp, _fp, C.c_size_t, _dp, _dp]),
    ("sga_multi_set_target_voxels", C.c_int, [_vp, C.c_double, C.c_void_p, _dp, _dp, C.c_size_t]),
    ("sga_multi_set_search_offsets", C.c_int, [_vp, C.c_int]),
    ("sga_multi_set_rejector", C.c_int, [_vp, C.c_void_p, _vp]),
    ("sga_multi_set_target_flat_voxels", C.c_int, [_vp, C.c_double, C.c_void_p, C.c_void_p, _dp, _dp, C.c_int, C.c_size_t]),
    ("sga_multi_set_source_f64", C.c_int, [_vp, _dp, _dp, _dp, C.c_size_t, _dp]),
    ("sga_multi_linearize", C.c_int, [_vp, C.POINTER(FactorParams), _dp, _dp, _dp, _dp, C.POINTER(C.c_uint64)]),
    ("sga_multi_error", C.c_int, [_vp, C.POINTER(FactorParams), _dp, _dp]),
    ("sga_multi_align", C.c_int, [_vp, _dp, C.POINTER(RegistrationSettingC), C.POINTER(ResultC)]),
    ("sga_multi_reset_search_state", C.c_int, [_vp]),
    ("sga_multi_get_factors", C.c_int, [_vp, C.c_void_p, C.c_void_p]),
    ("sga_optimize", C.c_int, [C.POINTER(RegistrationSettingC), _dp, LINEARIZE_FN, ERROR_FN, _vp, C.POINTER(ResultC)]),
    ("sga_se3_exp", None, [_dp, _dp]),
]

_LIB = None


def load():
    """Load the shared library (raises if it has not been built: `python -c 'import __graft_entry__ as g; g.build()'` or `make lib`)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise SgaError(f"{LIB_PATH} is missing: build it with `make lib` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # torch bundles its own libamdhip64.so with the same SONAME; if torch is (or will be) in this process it must be
    # loaded first so that exactly one HIP runtime is mapped.
    if "torch" in sys.modules or os.environ.get("SGA_WITH_TORCH", "0") == "1":
        import torch  # noqa: F401
    # contexts are streams, and streams that share a hardware queue serialise (csrc/context.hip: 4 queues per device unless asked otherwise;
    # read once, when the HIP runtime initialises — so before anything of it runs)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, restype, argtypes in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = restype
        fn.argtypes = argtypes
    _LIB = lib
    return lib


def check(rc):
    if rc != SGA_OK:
        raise SgaError(f"small_gicp_amd error {rc}: {load().sga_last_error().decode()}")
