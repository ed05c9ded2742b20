"""ctypes binding of libicp_mi355x.so (C ABI declared in include/icp_mi355x.h).

The product path has NO CPU fallback: if the shared library is missing or no MI355X is visible, loading / context
creation raises (`IcpLibraryError`), it never silently computes somewhere else.
"""
import ctypes as C
import os
from typing import Optional

import numpy as np

__all__ = ["IcpLibraryError", "IcpConfig", "IcpRegisterResult", "load_library", "library_path", "EXPORTED_SYMBOLS",
           "SCHEMES", "COSTS", "MEM_HOST", "MEM_DEVICE", "TARGETS_ALL", "TARGETS_SKIP_NULL", "STATUS_MESSAGES"]

MEM_HOST, MEM_DEVICE = 0, 1
TARGETS_ALL, TARGETS_SKIP_NULL = 0, 1
# RIGID_ALIGNMENT modes of the reference (slam/odometry/alignment.py:200-208) -> icp_cost
COSTS = {"point_to_plane_gauss_newton": 0, "point_to_point_gauss_newton": 1}

ICP_OK = 0
ICP_ERR_INVALID_ARGUMENT = -1
ICP_ERR_HIP = -2
ICP_ERR_INVALID_JACOBIAN = -3
ICP_ERR_EMPTY_MAP = -4
ICP_ERR_NO_DEVICE = -5
ICP_ERR_EXCHANGE = -6
BATCH_MAX_SEQUENCES = 32  # ICP_BATCH_MAX_SEQUENCES

STATUS_MESSAGES = {
    ICP_ERR_INVALID_ARGUMENT: "invalid argument",
    ICP_ERR_HIP: "HIP runtime error",
    ICP_ERR_INVALID_JACOBIAN: "Invalid Jacobian in Gauss Newton minimization",
    ICP_ERR_EMPTY_MAP: "the local map is empty",
    ICP_ERR_NO_DEVICE: "no MI355X (gfx950) device visible — the MI355X ICP path has no CPU fallback",
    ICP_ERR_EXCHANGE: "multi-GPU exchange: a peer did not deliver its normal equations in time",
}

# names of the reference's `_LS_SCHEME` members (slam/common/optimization.py:210-226) -> icp_scheme
SCHEMES = {"default": 0, "least_square": 0, "huber": 1, "exp": 2, "neighborhood": 3, "geman_mcclure": 4,
           "square_geman_mcclure": 5, "cauchy": 6}


class IcpLibraryError(RuntimeError):
    pass


class IcpConfig(C.Structure):
    _fields_ = [("height", C.c_int32), ("width", C.c_int32), ("up_fov", C.c_float), ("down_fov", C.c_float),
                ("max_num_alignments", C.c_int32), ("threshold_delta_pose", C.c_float), ("scheme", C.c_int32),
                ("sigma", C.c_float), ("local_map_size", C.c_int32), ("num_neighbors_normals", C.c_int32),
                ("cell_size", C.c_float), ("max_rings", C.c_int32), ("device", C.c_int32), ("poll_every", C.c_int32)]


class IcpRegisterResult(C.Structure):
    _fields_ = [("pose", C.c_float * 16), ("params", C.c_float * 6), ("iterations", C.c_int32),
                ("converged", C.c_int32), ("status", C.c_int32), ("num_targets", C.c_int32),
                ("normals_computed", C.c_int64)]


_P = C.c_void_p
_I64 = C.c_int64
_INT = C.c_int

# every symbol include/icp_mi355x.h declares: name -> (restype, argtypes)
EXPORTED_SYMBOLS = {
    "icp_default_config": (None, [C.POINTER(IcpConfig)]),
    "icp_create": (_INT, [C.POINTER(IcpConfig), C.POINTER(_P)]),
    "icp_destroy": (None, [_P]),
    "icp_last_error": (C.c_char_p, [_P]),
    "icp_version": (C.c_char_p, []),
    "icp_set_stream": (_INT, [_P, _P]),
    "icp_synchronize": (_INT, [_P]),
    "icp_set_option": (_INT, [_P, C.c_char_p, C.c_double]),
    "icp_set_alignment": (_INT, [_P, C.c_int32, C.c_float, C.c_int32, C.c_float]),
    "icp_set_cost": (_INT, [_P, C.c_int32]),
    "icp_exchange_create": (_INT, [_P, C.c_int32, C.c_int32, _P]),
    "icp_exchange_connect": (_INT, [_P, _P]),
    "icp_exchange_destroy": (_INT, [_P]),
    "icp_map_normals_owned": (_INT, [_P, C.c_int32, C.c_int32, _P]),
    "icp_map_normals_install": (_INT, [_P, _P]),
    "icp_project": (_INT, [_P, _P, _I64, _INT, _P, _P, _INT]),
    "icp_project_rows": (_INT, [_P, _P, _I64, _P, _P]),
    "icp_project_pixels": (_INT, [_P, _P, _I64, _INT, _P, _P, _INT]),
    "icp_kitti_correct_scan": (_INT, [_P, _P, _I64, _INT, _INT, _P, _INT]),
    "icp_grid_sample": (_INT, [_P, _P, _I64, _INT, C.c_double, _P, _P, C.POINTER(_I64), _INT]),
    "icp_voxel_hash": (_INT, [_P, _P, _I64, _INT, C.c_double, _P, _P, _INT]),
    "icp_grid_sample_f64": (_INT, [_P, _P, _I64, _INT, C.c_double, _P, _P, C.POINTER(_I64), _INT]),
    "icp_grid_sample_padded": (_INT, [_P, _P, _I64, C.c_double, _P, _P, _P]),
    "icp_grid_sample_padded_f64": (_INT, [_P, _P, _I64, C.c_double, _P, _P, _P]),
    "icp_distort": (_INT, [_P, _P, _P, _I64, _INT, _P, _P, _INT]),
    "icp_map_init": (_INT, [_P]),
    "icp_map_set": (_INT, [_P, _P, _I64, _INT]),
    "icp_map_update": (_INT, [_P, _P, _P, _I64, _INT, _INT, C.POINTER(_I64)]),
    "icp_map_stage_cloud": (_INT, [_P, _P, _I64, _INT, _INT]),
    "icp_map_update_staged": (_INT, [_P, _P, C.POINTER(_I64)]),
    "icp_map_update_vertex_map": (_INT, [_P, _P, _P, _INT, C.POINTER(_I64)]),
    "icp_map_size": (_I64, [_P]),
    "icp_map_num_clouds": (_INT, [_P]),
    "icp_handoff_fallbacks": (_INT, [_P]),
    "icp_map_get": (_INT, [_P, _P, _INT]),
    "icp_nearest_neighbor_search": (_INT, [_P, _P, _I64, _INT, _P, _P, _P, _INT]),
    "icp_last_neighbors": (_INT, [_P, _P, _P, _INT]),
    "icp_compute_normal_map": (_INT, [_P, _P, _INT, _INT, _P, _INT]),
    "icp_compute_neighbors": (_INT, [_P, _P, _P, _P, _INT, _INT, _INT, _P, _P, _INT]),
    "icp_pmap_init": (_INT, [_P]),
    "icp_pmap_update": (_INT, [_P, _P, _P, _INT, _INT]),
    "icp_pmap_num_maps": (_INT, [_P]),
    "icp_pmap_get_model": (_INT, [_P, _P, _P, _INT]),
    "icp_pmap_nearest_neighbor_search": (_INT, [_P, _P, _I64, _INT, _P, C.POINTER(_I64), _INT]),
    "icp_pmap_register": (_INT, [_P, _P, _I64, _INT, _INT, _P, C.POINTER(IcpRegisterResult), _P, _P]),
    "icp_align_point_to_plane": (_INT, [_P, _P, _P, _P, _I64, _INT, _P, _P, _P, _P, _P]),
    "icp_voxel_statistics": (_INT, [_P, _P, _I64, _INT, C.c_double, _P, _P, _P, _P, _P, _P, _P, _INT]),
    "icp_align_point_to_point": (_INT, [_P, _P, _P, _I64, _INT, _P, _P, _P, _P, _P, _P]),
    "icp_weighted_procrustes": (_INT, [_P, _P, _P, _P, _I64, _INT, _P]),
    "icp_compact_targets": (_INT, [_P, _P, _I64, _INT, _P, _I64]),
    "icp_register": (_INT, [_P, _P, _I64, _INT, _INT, _P, C.POINTER(IcpRegisterResult), _P, _P]),
    "icp_register_begin": (_INT, [_P, _P, _I64, _INT, _INT, _P]),
    "icp_register_launch": (_INT, [_P, _P, _I64, _INT, _INT, _P]),
    "icp_register_launch_from_last": (_INT, [_P, _P, _I64, _INT, _INT]),
    "icp_iteration_accumulate": (_INT, [_P]),
    "icp_iteration_solve": (_INT, [_P]),
    "icp_register_end": (_INT, [_P, C.POINTER(IcpRegisterResult), _P, _P]),
    "icp_batch_create": (_INT, [_P, C.c_int32, C.POINTER(_P)]),
    "icp_batch_destroy": (None, [_P]),
    "icp_batch_last_error": (C.c_char_p, [_P]),
    "icp_batch_set_stream": (_INT, [_P, _P]),
    "icp_batch_register_launch": (_INT, [_P, _P, _P, _INT, _INT, _P, _INT]),
    "icp_batch_project": (_INT, [_P, _P, _P, _P]),
    "icp_batch_map_update": (_INT, [_P]),
    "icp_batch_register_end": (_INT, [_P, _P, _P, _P]),
    "icp_normal_equations_ptr": (_P, [_P]),
    "icp_set_normal_equations_buffer": (_INT, [_P, _P]),
    "icp_profile_enable": (_INT, [_P, _INT]),
    "icp_profile_read_iterations": (_INT, [_P, _P, _P, C.c_int32]),
    "icp_profile_event_floor": (_INT, [_P, C.c_int32, C.POINTER(C.c_double)]),
    "icp_profile_read": (_INT, [_P, C.POINTER(C.c_double), C.POINTER(_I64), C.POINTER(C.c_double),
                                C.POINTER(C.c_double)]),
}

_LIB: Optional[C.CDLL] = None


def library_path() -> str:
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "_lib", "libicp_mi355x.so")


def load_library() -> C.CDLL:
    """Loads libicp_mi355x.so (once).  torch is imported first so that the HIP runtime torch ships
    (same SONAME libamdhip64.so.7) is the single runtime in the process."""
    global _LIB
    if _LIB is not None:
        return _LIB
    import torch  # noqa: F401  (must precede the dlopen, see docstring)
    path = library_path()
    if not os.path.exists(path):
        raise IcpLibraryError(f"{path} is missing: build it with `python __graft_entry__.py` "
                              f"(make -C pylidar-slam_amd/csrc). There is no CPU fallback.")
    try:
        lib = C.CDLL(path)
    except OSError as e:
        raise IcpLibraryError(f"cannot load {path}: {e}") from e
    for name, (restype, argtypes) in EXPORTED_SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise IcpLibraryError(f"{path} does not export {name}") from e
        fn.restype = restype
        fn.argtypes = argtypes
    _LIB = lib
    return lib


def as_f32_rows(a: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.ndim != 2 or a.shape[1] != 3:
        raise AssertionError(f"expected an [N, 3] array, got {a.shape}")
    return a
