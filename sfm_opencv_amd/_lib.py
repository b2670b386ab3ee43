"""ctypes loader for sfm_opencv_amd/libsfmhip.so (the C-ABI declared in include/sfmhip.h).

The library is built in-tree by sfm_opencv_amd/csrc/Makefile (hipcc --offload-arch=gfx950).  There is no
CPU fallback: if the library is missing, or no GPU is visible when a context is requested, this raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SFMHIP_LIB: another build of the same library (experiments/: the -DSFMHIP_EXPERIMENTS build with its A/B knobs); never a fallback
LIB_PATH = os.environ.get("SFMHIP_LIB") or os.path.join(_HERE, "libsfmhip.so")

OK, E_ARG, E_HIP, E_COMM, E_NUMERIC, E_NODEVICE = 0, -1, -2, -3, -4, -5


class SfmHipError(RuntimeError):
    pass


class BAOptions(C.Structure):
    _fields_ = [("max_num_iterations", C.c_int),
                ("initial_trust_region_radius", C.c_double), ("max_trust_region_radius", C.c_double),
                ("min_trust_region_radius", C.c_double), ("min_relative_decrease", C.c_double),
                ("min_lm_diagonal", C.c_double), ("max_lm_diagonal", C.c_double),
                ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double),
                ("parameter_tolerance", C.c_double), ("huber_delta", C.c_double),
                ("jacobi_scaling", C.c_int), ("fix_first_camera", C.c_int), ("fix_intrinsics", C.c_int),
                ("verbose", C.c_int), ("linearizer", C.c_int), ("solver", C.c_int)]


class BASummary(C.Structure):
    _fields_ = [("termination", C.c_int), ("iterations", C.c_int), ("successful_steps", C.c_int),
                ("num_residuals", C.c_int), ("initial_cost", C.c_double), ("final_cost", C.c_double),
                ("final_radius", C.c_double), ("final_gradient_max_norm", C.c_double),
                ("total_time_s", C.c_double), ("preprocessor_time_s", C.c_double),
                ("minimizer_time_s", C.c_double), ("postprocessor_time_s", C.c_double)]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)

# every symbol include/sfmhip.h declares (tests check that the library exports all of them)
SYMBOLS = [
    "sfmhip_create", "sfmhip_destroy", "sfmhip_set_stream", "sfmhip_synchronize", "sfmhip_trim", "sfmhip_last_error",
    "sfmhip_version", "sfmhip_set_kernel_timing", "sfmhip_match_kernel_ms",
    "sfmhip_descset_create_l2_host", "sfmhip_descset_create_l2_dev",
    "sfmhip_descset_create_hamming2_host", "sfmhip_descset_create_hamming2_dev",
    "sfmhip_descsets_create_l2_host", "sfmhip_descsets_create_hamming2_host",
    "sfmhip_descset_destroy", "sfmhip_descset_refresh", "sfmhip_descsets_refresh", "sfmhip_descset_info",
    "sfmhip_knn2_dev", "sfmhip_knn2_l2_f32", "sfmhip_knn2_hamming2_u8", "sfmhip_ratio_filter",
    "sfmhip_match_features_l2", "sfmhip_match_features_hamming2",
    "sfmhip_match_pairs_dev", "sfmhip_match_pairs", "sfmhip_l2_distance_matrix_dev", "sfmhip_selftest_exact_sqrt",
    "sfmhip_triangulate2_f32", "sfmhip_triangulate2_f32_dev", "sfmhip_triangulate2_matches_dev",
    "sfmhip_triangulate_tracks", "sfmhip_reprojection_errors",
    "sfmhip_ba_default_options", "sfmhip_ba_solve", "sfmhip_ba_create", "sfmhip_ba_destroy",
    "sfmhip_ba_set_allreduce", "sfmhip_ba_run", "sfmhip_ba_iterate", "sfmhip_ba_reset",
    "sfmhip_ba_get_params", "sfmhip_ba_reduced_system", "sfmhip_ba_phase_ms", "sfmhip_ba_debug_table",
    "sfmhip_estimate_normals",
    "sfmhip_rccl_available", "sfmhip_rccl_get_unique_id", "sfmhip_rccl_comm_create", "sfmhip_rccl_comm_destroy",
    "sfmhip_ba_set_rccl", "sfmhip_rccl_allreduce_f64", "sfmhip_ba_solve_multi", "sfmhip_match_pairs_multi", "sfmhip_debug_fail_allocations",
]

_lib = None


def load():
    """Load libsfmhip.so; raises SfmHipError if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SfmHipError(f"{LIB_PATH} not found: build it with `make -C sfm_opencv_amd/csrc` "
                          "(or __graft_entry__.build()); there is no CPU fallback")
    # PyTorch-ROCm bundles its own libamdhip64.so.7; two HIP runtimes in one process cannot both own the GPU.
    # Importing torch first makes the dynamic linker resolve our DT_NEEDED libamdhip64.so.7 to that copy.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    vp, i32, f32, f64, sz = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_size_t
    sig = {
        "sfmhip_create": (i32, [i32, C.POINTER(vp)]),
        "sfmhip_destroy": (None, [vp]),
        "sfmhip_set_stream": (i32, [vp, vp]),
        "sfmhip_synchronize": (i32, [vp]),
        "sfmhip_trim": (i32, [vp]),
        "sfmhip_last_error": (C.c_char_p, [vp]),
        "sfmhip_version": (C.c_char_p, []),
        "sfmhip_set_kernel_timing": (i32, [vp, i32]),
        "sfmhip_match_kernel_ms": (i32, [vp, C.POINTER(f64)]),
        "sfmhip_descset_create_l2_host": (i32, [vp, vp, i32, i32, sz, C.POINTER(vp)]),
        "sfmhip_descset_create_l2_dev": (i32, [vp, vp, i32, i32, sz, C.POINTER(vp)]),
        "sfmhip_descset_create_hamming2_host": (i32, [vp, vp, i32, i32, sz, C.POINTER(vp)]),
        "sfmhip_descset_create_hamming2_dev": (i32, [vp, vp, i32, i32, sz, C.POINTER(vp)]),
        "sfmhip_descsets_create_l2_host": (i32, [vp, C.POINTER(vp), vp, i32, C.POINTER(sz), i32, C.POINTER(vp)]),
        "sfmhip_descsets_create_hamming2_host": (i32, [vp, C.POINTER(vp), vp, i32, C.POINTER(sz), i32, C.POINTER(vp)]),
        "sfmhip_descset_destroy": (None, [vp]),
        "sfmhip_descset_refresh": (i32, [vp]),
        "sfmhip_descsets_refresh": (i32, [vp, C.POINTER(vp), i32]),
        "sfmhip_descset_info": (i32, [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]),
        "sfmhip_knn2_dev": (i32, [vp, vp, vp, vp, vp, i32]),
        "sfmhip_knn2_l2_f32": (i32, [vp, vp, i32, vp, i32, i32, sz, sz, vp, vp]),
        "sfmhip_knn2_hamming2_u8": (i32, [vp, vp, i32, vp, i32, i32, sz, sz, vp, vp]),
        "sfmhip_ratio_filter": (i32, [vp, vp, i32, f64, f32, f32, vp, C.POINTER(i32)]),
        "sfmhip_match_features_l2": (i32, [vp, vp, i32, vp, i32, i32, sz, sz, vp, C.POINTER(i32)]),
        "sfmhip_match_features_hamming2": (i32, [vp, vp, i32, vp, i32, i32, sz, sz, vp, C.POINTER(i32)]),
        "sfmhip_match_pairs_dev": (i32, [vp, C.POINTER(vp), i32, vp, i32, f64, f32, f32, vp, i32, vp]),
        "sfmhip_match_pairs": (i32, [vp, C.POINTER(vp), i32, vp, i32, f64, f32, f32, vp, i32, vp]),
        "sfmhip_l2_distance_matrix_dev": (i32, [vp, vp, vp, vp, sz, i32]),
        "sfmhip_selftest_exact_sqrt": (i32, [vp, C.POINTER(i32)]),
        "sfmhip_triangulate2_f32": (i32, [vp, vp, vp, vp, vp, i32, vp, vp]),
        "sfmhip_triangulate2_f32_dev": (i32, [vp, vp, vp, vp, vp, i32, vp, vp]),
        "sfmhip_triangulate2_matches_dev": (i32, [vp, vp, vp, vp, vp, vp, i32, vp, vp]),
        "sfmhip_triangulate_tracks": (i32, [vp, vp, vp, i32, vp, vp, vp, i32, i32, vp, vp]),
        "sfmhip_reprojection_errors": (i32, [vp, vp, vp, i32, vp, i32, vp, vp, vp, i32, vp]),
        "sfmhip_ba_default_options": (None, [C.POINTER(BAOptions)]),
        "sfmhip_ba_solve": (i32, [vp, vp, vp, i32, vp, i32, vp, vp, vp, i32, C.POINTER(BAOptions), C.POINTER(BASummary)]),
        "sfmhip_ba_create": (i32, [vp, vp, vp, i32, vp, i32, vp, vp, vp, i32, C.POINTER(BAOptions), C.POINTER(vp)]),
        "sfmhip_ba_destroy": (None, [vp]),
        "sfmhip_ba_set_allreduce": (i32, [vp, ALLREDUCE_FN, vp, i32, i32]),
        "sfmhip_ba_run": (i32, [vp, C.POINTER(BASummary)]),
        "sfmhip_ba_iterate": (i32, [vp, i32, C.POINTER(BASummary)]),
        "sfmhip_ba_reset": (i32, [vp]),
        "sfmhip_ba_get_params": (i32, [vp, vp, vp, vp]),
        "sfmhip_ba_reduced_system": (i32, [vp, f64, vp, vp, C.POINTER(i32), C.POINTER(f64)]),
        "sfmhip_ba_phase_ms": (i32, [vp, C.POINTER(f64)]),
        "sfmhip_ba_debug_table": (i32, [vp, C.c_char_p, vp, sz, C.POINTER(sz)]),
        "sfmhip_estimate_normals": (i32, [vp, vp, i32, i32, vp]),
        "sfmhip_rccl_available": (i32, []),
        "sfmhip_rccl_get_unique_id": (i32, [vp]),
        "sfmhip_rccl_comm_create": (i32, [vp, vp, i32, i32, C.POINTER(vp)]),
        "sfmhip_rccl_comm_destroy": (i32, [vp]),
        "sfmhip_ba_set_rccl": (i32, [vp, vp, i32, i32]),
        "sfmhip_rccl_allreduce_f64": (i32, [vp, vp, vp, sz]),
        "sfmhip_ba_solve_multi": (i32, [C.POINTER(vp), i32, vp, vp, i32, vp, i32, vp, vp, vp, i32, C.POINTER(BAOptions), C.POINTER(BASummary)]),
        "sfmhip_match_pairs_multi": (i32, [C.POINTER(vp), i32, i32, C.POINTER(vp), vp, i32, C.POINTER(sz), i32, vp, i32, f64, f32, f32, vp, i32, vp]),
        "sfmhip_debug_fail_allocations": (i32, [vp, i32]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
