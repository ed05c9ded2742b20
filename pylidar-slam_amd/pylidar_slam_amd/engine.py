"""`IcpContext`: the Python handle on one `icp_ctx` of libicp_mi355x.so.

Accepts numpy arrays (host memory, staged by the library) and torch-ROCm tensors (device memory, zero-copy through
`tensor.data_ptr()`); returns numpy arrays or torch tensors accordingly.  torch is used for device memory and streams
only — every computation happens in the HIP kernels behind the C ABI.
"""
import ctypes as C
from dataclasses import dataclass
from typing import Optional, Tuple, Union

import numpy as np
import torch

from . import _lib
from ._lib import (COSTS, IcpConfig, IcpLibraryError, IcpRegisterResult, MEM_DEVICE, MEM_HOST, SCHEMES,
                   STATUS_MESSAGES, TARGETS_ALL, TARGETS_SKIP_NULL)

Array = Union[np.ndarray, torch.Tensor]
_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)  # (device index) -> raw hipStream_t of torch's current stream


class InvalidJacobianError(RuntimeError):
    """RuntimeError("Invalid Jacobian in Gauss Newton minimization") of slam/common/optimization.py:336."""


class ExchangeTimeoutError(RuntimeError):
    """A peer rank did not deliver its normal equations within `exchange_timeout_ms` (in-library multi-GPU exchange)."""


@dataclass
class RegisterResult:
    pose: np.ndarray  # [4,4] f32
    params: np.ndarray  # [6] f32
    iterations: int
    converged: bool
    num_targets: int
    normals_computed: int
    losses: np.ndarray  # [iterations] f64
    dx: np.ndarray  # [iterations, 6] f32


def _ptr_mem(a: Optional[Array]) -> Tuple[Optional[int], int, object]:
    """(pointer, mem kind, keep-alive object) of an [.., 3]-float32 contiguous array / tensor."""
    if a is None:
        return None, MEM_HOST, None
    if isinstance(a, torch.Tensor):
        t = a
        if t.dtype != torch.float32 or not t.is_contiguous():
            t = t.to(torch.float32).contiguous()
        if t.is_cuda:
            return t.data_ptr(), MEM_DEVICE, t
        n = t.numpy()
        return n.ctypes.data, MEM_HOST, n
    n = np.ascontiguousarray(a, dtype=np.float32)
    return n.ctypes.data, MEM_HOST, n


def _on_device(*arrays) -> bool:
    return any(isinstance(a, torch.Tensor) and a.is_cuda for a in arrays)


def _pose16(m) -> "C.Array":
    a = np.ascontiguousarray(np.asarray(m, dtype=np.float32).reshape(4, 4))
    return (C.c_float * 16)(*a.reshape(-1).tolist())


class IcpContext:
    def __init__(self, height: int = 64, width: int = 1024, up_fov: float = 3.0, down_fov: float = -24.0,
                 max_num_alignments: int = 100, threshold_delta_pose: float = 1.0e-4, scheme: str = "default",
                 sigma: float = 0.5, local_map_size: int = 20, num_neighbors_normals: int = 10,
                 cell_size: float = 0.0, max_rings: int = 2, device: int = 0, poll_every: int = 4):
        self._lib = _lib.load_library()
        cfg = IcpConfig()
        self._lib.icp_default_config(C.byref(cfg))
        if scheme not in SCHEMES:
            raise AssertionError(f"unknown weighting scheme {scheme}")
        cfg.height, cfg.width, cfg.up_fov, cfg.down_fov = int(height), int(width), float(up_fov), float(down_fov)
        cfg.max_num_alignments, cfg.threshold_delta_pose = int(max_num_alignments), float(threshold_delta_pose)
        cfg.scheme, cfg.sigma = SCHEMES[scheme], float(sigma)
        cfg.local_map_size, cfg.num_neighbors_normals = int(local_map_size), int(num_neighbors_normals)
        cfg.cell_size, cfg.max_rings, cfg.device, cfg.poll_every = float(cell_size), int(max_rings), int(device), \
            int(poll_every)
        self.config = cfg
        self.device = torch.device("cuda", int(device))
        self._device_index = int(device)
        handle = C.c_void_p()
        rc = self._lib.icp_create(C.byref(cfg), C.byref(handle))
        if rc != 0:
            raise IcpLibraryError(f"icp_create failed: {STATUS_MESSAGES.get(rc, rc)}")
        self._h = handle
        self._neq_tensor: Optional[torch.Tensor] = None

    # ------------------------------------------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._lib.icp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc == 0:
            return
        msg = self._lib.icp_last_error(self._h).decode() or STATUS_MESSAGES.get(rc, str(rc))
        if rc == _lib.ICP_ERR_INVALID_JACOBIAN:
            raise InvalidJacobianError("Invalid Jacobian in Gauss Newton minimization")
        if rc == _lib.ICP_ERR_INVALID_ARGUMENT:
            raise AssertionError(msg)
        if rc == _lib.ICP_ERR_EXCHANGE:
            raise ExchangeTimeoutError(msg)
        raise RuntimeError(f"libicp_mi355x: {msg} ({rc})")

    def use_torch_stream(self):
        """Enqueue on torch's current HIP stream of this device (the library orders a switch of streams against the work
        it enqueued on the previous one)."""
        # (the raw handle straight from torch's bookkeeping where the build exposes it: the public query builds a Stream
        # object, 3-10 us a call — and every call that takes a device tensor asks)
        stream = _RAW_STREAM(self._device_index) if _RAW_STREAM is not None else \
            torch.cuda.current_stream(self.device).cuda_stream
        if stream != getattr(self, "_bound_stream", None):
            self._check(self._lib.icp_set_stream(self._h, C.c_void_p(stream)))
            self._bound_stream = stream

    def _bind(self, *arrays):
        """Device tensors come from (and results go to) torch's current stream: make sure the library enqueues there."""
        if _on_device(*arrays):
            self.use_torch_stream()

    def synchronize(self):
        self._check(self._lib.icp_synchronize(self._h))

    def set_cost(self, mode: str):
        """Alignment mode of the registration loop: a RIGID_ALIGNMENT member name of the reference
        ("point_to_plane_gauss_newton" | "point_to_point_gauss_newton")."""
        if mode not in COSTS:
            raise AssertionError(f"unknown alignment mode {mode}")
        self._check(self._lib.icp_set_cost(self._h, COSTS[mode]))

    def set_option(self, name: str, value: float):
        """MI355X-side tuning option (see `icp_set_option` in include/icp_mi355x.h); never changes a result."""
        self._check(self._lib.icp_set_option(self._h, name.encode(), float(value)))

    def set_alignment(self, scheme: str, sigma: float, max_num_alignments: int, threshold_delta_pose: float):
        self._check(self._lib.icp_set_alignment(self._h, SCHEMES[scheme], float(sigma), int(max_num_alignments),
                                                float(threshold_delta_pose)))
        self.config.scheme, self.config.sigma = SCHEMES[scheme], float(sigma)
        self.config.max_num_alignments = int(max_num_alignments)
        self.config.threshold_delta_pose = float(threshold_delta_pose)

    # ---- projection --------------------------------------------------------------------------------------------------
    def project(self, points: Array, with_index: bool = False, out: Optional[torch.Tensor] = None):
        """Vertex map [3, H, W] (same kind as the input: numpy in -> numpy out, cuda tensor in -> cuda tensor out)."""
        self._bind(points)
        p, mem, keep = _ptr_mem(points)
        n = int(keep.shape[0]) if keep is not None else 0
        h, w = self.config.height, self.config.width
        if mem == MEM_DEVICE:
            vmap = out if out is not None else torch.empty((3, h, w), dtype=torch.float32, device=keep.device)
            idx = torch.empty((h, w), dtype=torch.int32, device=keep.device) if with_index else None
            self._check(self._lib.icp_project(self._h, p, n, mem, vmap.data_ptr(),
                                              idx.data_ptr() if idx is not None else None, MEM_DEVICE))
        else:
            vmap = np.empty((3, h, w), dtype=np.float32)
            idx = np.empty((h, w), dtype=np.int32) if with_index else None
            self._check(self._lib.icp_project(self._h, p, n, mem, vmap.ctypes.data,
                                              idx.ctypes.data if idx is not None else None, MEM_HOST))
        return (vmap, idx) if with_index else vmap

    def project_rows(self, points: torch.Tensor):
        """Device-resident projection: (vertex map [3, H, W], the same pixels as rows [H * W, 3]) from one launch — the rows
        are `vmap.permute(1, 2, 0).reshape(-1, 3)` without the transposing copy."""
        self._bind(points)
        p, mem, keep = _ptr_mem(points)
        if mem != MEM_DEVICE:
            raise AssertionError("project_rows takes a device tensor")
        h, w = self.config.height, self.config.width
        vmap = torch.empty((3, h, w), dtype=torch.float32, device=keep.device)
        rows = torch.empty((h * w, 3), dtype=torch.float32, device=keep.device)
        self._check(self._lib.icp_project_rows(self._h, p, int(keep.shape[0]), vmap.data_ptr(), rows.data_ptr()))
        return vmap, rows

    def project_pixels(self, points: np.ndarray):
        p, mem, keep = _ptr_mem(points)
        n = int(keep.shape[0])
        rows, cols = np.empty(n, np.float32), np.empty(n, np.float32)
        self._check(self._lib.icp_project_pixels(self._h, p, n, mem, rows.ctypes.data, cols.ctypes.data, MEM_HOST))
        return rows, cols

    def kitti_correct_scan(self, scan: np.ndarray) -> np.ndarray:
        """`KITTIOdometrySequence.correct_scan`: [N,4] (or [N,3]) float32 -> corrected xyz [N,3] float64."""
        a = np.ascontiguousarray(scan, dtype=np.float32)
        if a.ndim != 2 or a.shape[1] < 3:
            raise AssertionError(f"expected [N, >=3] rows, got {a.shape}")
        out = np.empty((a.shape[0], 3), np.float64)
        self._check(self._lib.icp_kitti_correct_scan(self._h, a.ctypes.data, int(a.shape[0]), int(a.shape[1]), MEM_HOST,
                                                     out.ctypes.data, MEM_HOST))
        return out

    # ---- grid sampling -----------------------------------------------------------------------------------------------
    def voxel_hash(self, points: np.ndarray, voxel_size: float):
        p, mem, keep = _ptr_mem(points)
        n = int(keep.shape[0])
        vox, hashes = np.empty((n, 3), np.int64), np.empty(n, np.int64)
        self._check(self._lib.icp_voxel_hash(self._h, p, n, mem, float(voxel_size), vox.ctypes.data,
                                             hashes.ctypes.data, MEM_HOST))
        return vox, hashes

    def grid_sample(self, points: Array, voxel_size: float):
        """(sample points [V,3], indices [V] int64), ordered by ascending voxel hash."""
        self._bind(points)
        p, mem, keep = _ptr_mem(points)
        n = int(keep.shape[0])
        count = C.c_int64(0)
        if mem == MEM_DEVICE:
            idx = torch.empty(max(n, 1), dtype=torch.int64, device=keep.device)
            pts = torch.empty((max(n, 1), 3), dtype=torch.float32, device=keep.device)
            self._check(self._lib.icp_grid_sample(self._h, p, n, mem, float(voxel_size), idx.data_ptr(),
                                                  pts.data_ptr(), C.byref(count), MEM_DEVICE))
            return pts[:count.value], idx[:count.value]
        idx = np.empty(max(n, 1), np.int64)
        pts = np.empty((max(n, 1), 3), np.float32)
        self._check(self._lib.icp_grid_sample(self._h, p, n, mem, float(voxel_size), idx.ctypes.data,
                                              pts.ctypes.data, C.byref(count), MEM_HOST))
        return pts[:count.value].copy(), idx[:count.value].copy()

    def grid_sample_padded(self, points: "torch.Tensor", voxel_size: float):
        """The grid sample of the device-resident pipeline (`icp_grid_sample_padded[_f64]`): a cuda tensor in, and —
        without any synchronisation — (points [n,3] with the V samples first and NaN rows behind them, indices [n] int64
        with -1 behind the samples, V as a 0-dim int32 cuda tensor) out."""
        self._bind(points)
        if not (isinstance(points, torch.Tensor) and points.is_cuda):
            raise AssertionError("grid_sample_padded takes a cuda tensor (the device-resident pipeline)")
        f64 = points.dtype == torch.float64
        t = points.contiguous() if f64 else points.to(torch.float32).contiguous()
        n = int(t.shape[0])
        idx = torch.empty(max(n, 1), dtype=torch.int64, device=t.device)
        out = torch.empty((max(n, 1), 3), dtype=t.dtype, device=t.device)
        count = torch.empty((), dtype=torch.int32, device=t.device)
        fn = self._lib.icp_grid_sample_padded_f64 if f64 else self._lib.icp_grid_sample_padded
        self._check(fn(self._h, t.data_ptr(), n, float(voxel_size), idx.data_ptr(), out.data_ptr(), count.data_ptr()))
        return out[:n], idx[:n], count

    def voxel_statistics(self, points: np.ndarray, voxel_size: float, with_normal_distribution: bool = True):
        """`Voxelization.filter` (slam/preprocessing.py:63-98): dict with voxel_coordinates [n,3] i64, voxel_hashes [n]
        i64, voxel_indices [n] i64 and — with_normal_distribution — voxel_sizes [V] i64, voxel_means [V,3] f32,
        voxel_covariances [V,3,3] f32 (voxels by ascending hash)."""
        p, mem, keep = _ptr_mem(points)
        if mem != MEM_HOST:
            raise AssertionError("voxel_statistics takes a host array (the reference filter is numpy-only)")
        n = int(keep.shape[0])
        vox, hashes, ids = np.empty((n, 3), np.int64), np.empty(n, np.int64), np.empty(n, np.int64)
        nv = C.c_int64(0)
        sizes = means = covs = None
        if with_normal_distribution:
            sizes, means, covs = np.empty(n, np.int64), np.empty((n, 3), np.float32), np.empty((n, 3, 3), np.float32)
        self._check(self._lib.icp_voxel_statistics(
            self._h, p, n, mem, float(voxel_size), vox.ctypes.data, hashes.ctypes.data, ids.ctypes.data, C.byref(nv),
            sizes.ctypes.data if sizes is not None else None, means.ctypes.data if means is not None else None,
            covs.ctypes.data if covs is not None else None, MEM_HOST))
        out = {"voxel_coordinates": vox, "voxel_hashes": hashes, "voxel_indices": ids, "num_voxels": int(nv.value)}
        if with_normal_distribution:
            v = int(nv.value)
            out.update(voxel_sizes=sizes[:v].copy(), voxel_means=means[:v].copy(), voxel_covariances=covs[:v].copy())
        return out

    def grid_sample_f64(self, points: Array, voxel_size: float):
        """float64 cloud (the output of `distort`) -> (sample points [V,3] f64, indices [V] int64); a cuda tensor stays
        on the device (zero-copy in, device tensors out)."""
        self._bind(points)
        count = C.c_int64(0)
        if isinstance(points, torch.Tensor) and points.is_cuda:
            t = points.to(torch.float64).contiguous()
            n = int(t.shape[0])
            idx = torch.empty(max(n, 1), dtype=torch.int64, device=t.device)
            out = torch.empty((max(n, 1), 3), dtype=torch.float64, device=t.device)
            self._check(self._lib.icp_grid_sample_f64(self._h, t.data_ptr(), n, MEM_DEVICE, float(voxel_size),
                                                      idx.data_ptr(), out.data_ptr(), C.byref(count), MEM_DEVICE))
            return out[:count.value], idx[:count.value]
        pts64 = np.ascontiguousarray(points, dtype=np.float64)
        n = int(pts64.shape[0])
        idx = np.empty(max(n, 1), np.int64)
        out = np.empty((max(n, 1), 3), np.float64)
        self._check(self._lib.icp_grid_sample_f64(self._h, pts64.ctypes.data, n, MEM_HOST, float(voxel_size),
                                                  idx.ctypes.data, out.ctypes.data, C.byref(count), MEM_HOST))
        return out[:count.value].copy(), idx[:count.value].copy()

    # ---- de-skew -----------------------------------------------------------------------------------------------------
    def distort(self, points: Array, timestamps: Array, rel_pose):
        """`Distortion.filter`: [N,3] f32 points + [N] f64 timestamps + 4x4 initial motion -> [N,3] f64.  cuda tensors in
        -> cuda tensor out (nothing crosses PCIe but the 4x4 pose)."""
        self._bind(points)
        pose = np.ascontiguousarray(np.asarray(rel_pose, dtype=np.float64).reshape(4, 4))
        if isinstance(points, torch.Tensor) and points.is_cuda:
            pts = points.to(torch.float32).contiguous()
            ts = torch.as_tensor(timestamps).to(pts.device, torch.float64).reshape(-1).contiguous()
            if pts.ndim != 2 or pts.shape[1] != 3 or ts.shape[0] != pts.shape[0]:
                raise AssertionError(f"expected [N,3] points and [N] timestamps, got {tuple(pts.shape)} / {tuple(ts.shape)}")
            out = torch.empty((pts.shape[0], 3), dtype=torch.float64, device=pts.device)
            self._check(self._lib.icp_distort(self._h, pts.data_ptr(), ts.data_ptr(), int(pts.shape[0]), MEM_DEVICE,
                                              pose.ctypes.data, out.data_ptr(), MEM_DEVICE))
            return out
        pts = np.ascontiguousarray(points, dtype=np.float32)
        ts = np.ascontiguousarray(np.asarray(timestamps).reshape(-1), dtype=np.float64)
        if pts.ndim != 2 or pts.shape[1] != 3 or ts.shape[0] != pts.shape[0]:
            raise AssertionError(f"expected [N,3] points and [N] timestamps, got {pts.shape} / {ts.shape}")
        out = np.empty((pts.shape[0], 3), np.float64)
        self._check(self._lib.icp_distort(self._h, pts.ctypes.data, ts.ctypes.data, int(pts.shape[0]), MEM_HOST,
                                          pose.ctypes.data, out.ctypes.data, MEM_HOST))
        return out

    # ---- local map ---------------------------------------------------------------------------------------------------
    def map_init(self):
        self._check(self._lib.icp_map_init(self._h))

    def map_set(self, points: Array):
        self._bind(points)
        p, mem, keep = _ptr_mem(points)
        self._check(self._lib.icp_map_set(self._h, p, int(keep.shape[0]), mem))

    def map_update(self, rel_pose, new_points: Optional[Array] = None, skip_null: bool = False) -> int:
        self._bind(new_points)
        p, mem, keep = _ptr_mem(new_points)
        n = int(keep.shape[0]) if keep is not None else 0
        if keep is not None and n == 0:
            # an empty cloud still counts as a cloud in the reference's bookkeeping; give the library a valid pointer
            keep = np.zeros((1, 3), np.float32)
            p = keep.ctypes.data
            mem = MEM_HOST
        ins = C.c_int64(0)
        # rel_pose None: the device-resident pose of the last registration (no host round trip)
        self._check(self._lib.icp_map_update(self._h, _pose16(rel_pose) if rel_pose is not None else None, p, n, mem,
                                             TARGETS_SKIP_NULL if skip_null else TARGETS_ALL, C.byref(ins)))
        return int(ins.value)

    def map_stage_cloud(self, new_points: Array, skip_null: bool = False) -> None:
        """Prepares the cloud a later `map_update_staged` inserts (icp_map_stage_cloud): its valid rows are compacted on the
        device and their count is sent to the host without waiting — called before a registration, the update after it
        needs no host round trip of its own."""
        self._bind(new_points)
        p, mem, keep = _ptr_mem(new_points)
        n = int(keep.shape[0])
        self._check(self._lib.icp_map_stage_cloud(self._h, p if n else None, n, mem,
                                                  TARGETS_SKIP_NULL if skip_null else TARGETS_ALL))

    def map_update_staged(self, rel_pose) -> int:
        ins = C.c_int64(0)
        self._check(self._lib.icp_map_update_staged(self._h, _pose16(rel_pose) if rel_pose is not None else None,
                                                    C.byref(ins)))
        return int(ins.value)

    def map_update_vertex_map(self, rel_pose, vmap: Array) -> int:
        self._bind(vmap)
        p, mem, keep = _ptr_mem(vmap)
        ins = C.c_int64(0)
        self._check(self._lib.icp_map_update_vertex_map(self._h, _pose16(rel_pose), p, mem, C.byref(ins)))
        return int(ins.value)

    def map_size(self) -> int:
        return int(self._lib.icp_map_size(self._h))

    def map_num_clouds(self) -> int:
        return int(self._lib.icp_map_num_clouds(self._h))

    def handoff_fallbacks(self) -> int:
        """Registrations finished on per-iteration launches behind a timed-out hand-off (`icp_handoff_fallbacks`)."""
        return int(self._lib.icp_handoff_fallbacks(self._h))

    def map_points(self) -> np.ndarray:
        out = np.empty((self.map_size(), 3), np.float32)
        if out.shape[0]:
            self._check(self._lib.icp_map_get(self._h, out.ctypes.data, MEM_HOST))
        return out

    def nearest_neighbor_search(self, points: Array, with_normals: bool = True, with_index: bool = False):
        self._bind(points)
        p, mem, keep = _ptr_mem(points)
        n = int(keep.shape[0])
        if mem == MEM_DEVICE:
            nb = torch.empty((n, 3), dtype=torch.float32, device=keep.device)
            nm = torch.empty((n, 3), dtype=torch.float32, device=keep.device) if with_normals else None
            ix = torch.empty(n, dtype=torch.int32, device=keep.device) if with_index else None
            self._check(self._lib.icp_nearest_neighbor_search(
                self._h, p, n, mem, nb.data_ptr(), nm.data_ptr() if nm is not None else None,
                ix.data_ptr() if ix is not None else None, MEM_DEVICE))
        else:
            nb = np.empty((n, 3), np.float32)
            nm = np.empty((n, 3), np.float32) if with_normals else None
            ix = np.empty(n, np.int32) if with_index else None
            self._check(self._lib.icp_nearest_neighbor_search(
                self._h, p, n, mem, nb.ctypes.data, nm.ctypes.data if nm is not None else None,
                ix.ctypes.data if ix is not None else None, MEM_HOST))
        return nb, nm, ix

    def last_neighbors(self, n: int):
        """(original map index per target, [3,4] pose) of the LAST iteration of the last registration, read back from the
        library's exact nearest-neighbour cache (test support: `icp_last_neighbors`)."""
        ix = np.empty(int(n), np.int32)  # n = the number of target rows of that registration
        pose = np.empty(12, np.float32)
        self._check(self._lib.icp_last_neighbors(self._h, ix.ctypes.data, pose.ctypes.data, MEM_HOST))
        return ix, pose.reshape(3, 4)

    # ---- projective local map --------------------------------------------------------------------
    def _planar(self, vmap: Array):
        """[3,H,W] float32 contiguous -> (pointer, mem, keep-alive)."""
        h, w = self.config.height, self.config.width
        if isinstance(vmap, torch.Tensor):
            t = vmap.to(torch.float32).contiguous()
            if tuple(t.shape[-3:]) != (3, h, w):
                raise AssertionError(f"expected a [3,{h},{w}] vertex map, got {tuple(t.shape)}")
            if t.is_cuda:
                return t.data_ptr(), MEM_DEVICE, t
            a = t.numpy()
            return a.ctypes.data, MEM_HOST, a
        a = np.ascontiguousarray(vmap, dtype=np.float32)
        if a.shape[-3:] != (3, h, w):
            raise AssertionError(f"expected a [3,{h},{w}] vertex map, got {a.shape}")
        return a.ctypes.data, MEM_HOST, a

    def compute_normal_map(self, vmap: Array, kernel_size: int = 5):
        p, mem, keep = self._planar(vmap)
        h, w = self.config.height, self.config.width
        if mem == MEM_DEVICE:
            out = torch.empty((3, h, w), dtype=torch.float32, device=keep.device)
            self._check(self._lib.icp_compute_normal_map(self._h, p, mem, int(kernel_size), out.data_ptr(), MEM_DEVICE))
            return out
        out = np.empty((3, h, w), np.float32)
        self._check(self._lib.icp_compute_normal_map(self._h, p, mem, int(kernel_size), out.ctypes.data, MEM_HOST))
        return out

    def compute_neighbors(self, vm_target: np.ndarray, vm_reference: np.ndarray,
                          reference_fields: Optional[np.ndarray] = None):
        h, w = self.config.height, self.config.width
        t = np.ascontiguousarray(vm_target, dtype=np.float32).reshape(3, h, w)
        r = np.ascontiguousarray(vm_reference, dtype=np.float32).reshape(-1, 3, h, w)
        k = r.shape[0]
        f = fo = None
        c = 0
        if reference_fields is not None:
            f = np.ascontiguousarray(reference_fields, dtype=np.float32).reshape(k, -1, h, w)
            c = f.shape[1]
            fo = np.empty((c, h, w), np.float32)
        nb = np.empty((3, h, w), np.float32)
        self._check(self._lib.icp_compute_neighbors(self._h, t.ctypes.data, r.ctypes.data,
                                                    f.ctypes.data if f is not None else None, k, c, MEM_HOST,
                                                    nb.ctypes.data, fo.ctypes.data if fo is not None else None,
                                                    MEM_HOST))
        return nb, fo

    def pmap_init(self):
        self._check(self._lib.icp_pmap_init(self._h))

    def pmap_update(self, rel_pose, vmap: Optional[Array] = None, normals_kernel_size: int = 5):
        if vmap is None:
            self._check(self._lib.icp_pmap_update(self._h, _pose16(rel_pose), None, MEM_HOST, int(normals_kernel_size)))
            return
        p, mem, keep = self._planar(vmap)
        self._check(self._lib.icp_pmap_update(self._h, _pose16(rel_pose), p, mem, int(normals_kernel_size)))

    def pmap_num_maps(self) -> int:
        return int(self._lib.icp_pmap_num_maps(self._h))

    def pmap_model(self):
        """(`_model_vmap` [K,3,H,W], `_model_nmap` [K,3,H,W]) as numpy arrays."""
        k, h, w = self.pmap_num_maps(), self.config.height, self.config.width
        v4 = np.empty((k, h * w, 4), np.float32)
        n4 = np.empty((k, h * w, 4), np.float32)
        if k:
            self._check(self._lib.icp_pmap_get_model(self._h, v4.ctypes.data, n4.ctypes.data, MEM_HOST))
        to_maps = lambda a: np.ascontiguousarray(a[:, :, :3].reshape(k, h, w, 3).transpose(0, 3, 1, 2))
        return to_maps(v4), to_maps(n4)

    def pmap_nearest_neighbor_search(self, points: Array):
        """(neighbour points, neighbour normals, new target points), each [n,3], matched pixels in pixel order; cuda
        tensors for cuda points, numpy arrays otherwise."""
        self._bind(points)
        p, mem, keep = _ptr_mem(points)
        n = int(keep.shape[0])
        npix = self.config.height * self.config.width
        count = C.c_int64(0)
        if mem == MEM_DEVICE:
            rows = torch.empty((npix, 9), dtype=torch.float32, device=keep.device)
            self._check(self._lib.icp_pmap_nearest_neighbor_search(self._h, p, n, mem, rows.data_ptr(), C.byref(count),
                                                                   MEM_DEVICE))
            r = rows[:count.value]
            return r[:, 0:3].contiguous(), r[:, 3:6].contiguous(), r[:, 6:9].contiguous()
        rows = np.empty((npix, 9), np.float32)
        self._check(self._lib.icp_pmap_nearest_neighbor_search(self._h, p, n, mem, rows.ctypes.data, C.byref(count),
                                                               MEM_HOST))
        r = rows[:count.value]
        return r[:, 0:3].copy(), r[:, 3:6].copy(), r[:, 6:9].copy()

    def pmap_register(self, points: Array, init_pose=None, skip_null: bool = False) -> RegisterResult:
        self._bind(points)
        p, mem, keep = _ptr_mem(points)
        n = int(keep.shape[0])
        cap = max(1, int(self.config.max_num_alignments))
        losses = (C.c_double * cap)()
        dxs = (C.c_float * (6 * cap))()
        res = IcpRegisterResult()
        init = _pose16(init_pose if init_pose is not None else np.eye(4))
        self._check(self._lib.icp_pmap_register(self._h, p, n, mem, TARGETS_SKIP_NULL if skip_null else TARGETS_ALL,
                                                init, C.byref(res), losses, dxs))
        return self._result(res, losses, dxs)

    # ---- alignment ---------------------------------------------------------------------------------------------------
    def _residual_buffer(self, n: int, mem: int, like, with_residuals: bool):
        if not with_residuals:
            return None, None
        if mem == MEM_DEVICE:
            buf = torch.empty(n, dtype=torch.float32, device=like.device)
            return buf, buf.data_ptr()
        buf = np.empty(n, np.float32)
        return buf, buf.ctypes.data

    def align_point_to_plane(self, ref_points: Array, tgt_points: Array, ref_normals: Array,
                             with_residuals: bool = False):
        """One Gauss-Newton point-to-plane step: (pose [4,4], dx [6], loss, normal equations [32] f64[, residuals [n]
        = (w r)^2 per row, numpy / cuda tensor like the inputs])."""
        self._bind(ref_points, tgt_points, ref_normals)
        r, mem_r, kr = _ptr_mem(ref_points)
        t, mem_t, kt = _ptr_mem(tgt_points)
        nn, mem_n, kn = _ptr_mem(ref_normals)
        if not (mem_r == mem_t == mem_n):
            raise AssertionError("ref / tgt / normals must live in the same memory space")
        n = int(kr.shape[0])
        if not (kt.shape[0] == n and kn.shape[0] == n):
            raise AssertionError("ref / tgt / normals must have the same number of rows")
        dx = (C.c_float * 6)()
        pose = (C.c_float * 16)()
        loss = C.c_double(0)
        neq = (C.c_double * 32)()
        res, res_ptr = self._residual_buffer(n, mem_r, kr, with_residuals)
        self._check(self._lib.icp_align_point_to_plane(self._h, r, t, nn, n, mem_r, dx, pose, C.byref(loss), neq,
                                                       res_ptr))
        out = (np.array(pose, np.float32).reshape(4, 4), np.array(dx, np.float32), float(loss.value),
               np.array(neq, np.float64))
        return out + (res,) if with_residuals else out

    def align_point_to_point(self, ref_points: Array, tgt_points: Array, x0=None, with_residuals: bool = False):
        """One Gauss-Newton point-to-point step linearised at x0 ([6] or None = zeros):
        (pose [4,4], params [6] = x0 + dx, loss, normal equations [32] f64[, residuals [n]])."""
        self._bind(ref_points, tgt_points)
        r, mem_r, kr = _ptr_mem(ref_points)
        t, mem_t, kt = _ptr_mem(tgt_points)
        if mem_r != mem_t:
            raise AssertionError("ref / tgt must live in the same memory space")
        n = int(kr.shape[0])
        if kt.shape[0] != n:
            raise AssertionError("ref / tgt must have the same number of rows")
        x = (C.c_float * 6)(*[float(v) for v in np.asarray(x0, np.float32).reshape(6)]) if x0 is not None else None
        params = (C.c_float * 6)()
        pose = (C.c_float * 16)()
        loss = C.c_double(0)
        neq = (C.c_double * 32)()
        res, res_ptr = self._residual_buffer(n, mem_r, kr, with_residuals)
        self._check(self._lib.icp_align_point_to_point(self._h, r, t, n, mem_r, x, params, pose, C.byref(loss), neq,
                                                       res_ptr))
        out = (np.array(pose, np.float32).reshape(4, 4), np.array(params, np.float32), float(loss.value),
               np.array(neq, np.float64))
        return out + (res,) if with_residuals else out

    def weighted_procrustes(self, tgt_points: Array, ref_points: Array, weights=None) -> np.ndarray:
        """`weighted_procrustes` (slam/common/registration.py:15-74): [4,4] float64 transform target -> reference."""
        t, mem_t, kt = _ptr_mem(tgt_points)
        r, mem_r, kr = _ptr_mem(ref_points)
        if mem_r != mem_t or kt.shape[0] != kr.shape[0]:
            raise AssertionError("target / reference must have the same shape and memory space")
        w, keep_w = None, None
        if weights is not None:
            if isinstance(weights, torch.Tensor):
                keep_w = weights.reshape(-1).to(torch.float32).contiguous()
                if (MEM_DEVICE if keep_w.is_cuda else MEM_HOST) != mem_t:
                    raise AssertionError("weights must live where the points live")
                w = keep_w.data_ptr()
            else:
                if mem_t != MEM_HOST:
                    raise AssertionError("weights must live where the points live")
                keep_w = np.ascontiguousarray(np.asarray(weights, np.float32).reshape(-1))
                w = keep_w.ctypes.data
            if keep_w.shape[0] != kt.shape[0]:
                raise AssertionError("one weight per point")
        out = (C.c_double * 16)()
        self._check(self._lib.icp_weighted_procrustes(self._h, t, r, w, int(kt.shape[0]), mem_t, out))
        return np.array(out, np.float64).reshape(4, 4)

    def compact_targets(self, rows: torch.Tensor, cap: int, skip_null: bool = True) -> torch.Tensor:
        """[cap,3] device tensor: the rows of `rows` [n,3] (a cuda tensor) that a registration with this masking would
        use, in order, at its head, null rows behind them — so registering it with skip_null walks `cap` rows instead of
        n.  `cap` must bound the number of passing rows (the caller's knowledge: a vertex map built from N points has at
        most N non-null pixels).  No host round trip."""
        if not (isinstance(rows, torch.Tensor) and rows.is_cuda):
            raise AssertionError("compact_targets works on device tensors")
        self._bind(rows)
        t = rows if rows.dtype == torch.float32 and rows.is_contiguous() else rows.to(torch.float32).contiguous()
        out = torch.empty((max(int(cap), 1), 3), dtype=torch.float32, device=t.device)
        self._check(self._lib.icp_compact_targets(self._h, t.data_ptr(), int(t.shape[0]),
                                                  TARGETS_SKIP_NULL if skip_null else TARGETS_ALL, out.data_ptr(),
                                                  int(cap)))
        return out[:int(cap)]

    # ---- registration ------------------------------------------------------------------------------------------------
    def _result(self, res: IcpRegisterResult, losses, dxs) -> RegisterResult:
        k = int(res.iterations)
        return RegisterResult(np.array(res.pose, np.float32).reshape(4, 4), np.array(res.params, np.float32), k,
                              bool(res.converged), int(res.num_targets), int(res.normals_computed),
                              np.array(losses[:k], np.float64), np.array(dxs, np.float32).reshape(-1, 6)[:k])

    def register(self, points: Array, init_pose=None, skip_null: bool = False) -> RegisterResult:
        self._bind(points)
        p, mem, keep = _ptr_mem(points)
        n = int(keep.shape[0])
        cap = max(1, int(self.config.max_num_alignments))
        losses = (C.c_double * cap)()
        dxs = (C.c_float * (6 * cap))()
        res = IcpRegisterResult()
        init = _pose16(init_pose if init_pose is not None else np.eye(4))
        self._check(self._lib.icp_register(self._h, p, n, mem, TARGETS_SKIP_NULL if skip_null else TARGETS_ALL, init,
                                           C.byref(res), losses, dxs))
        return self._result(res, losses, dxs)

    # ---- multi-GPU: exchange of the normal equations inside the library ----------------------------------------------
    def exchange_create(self, rank: int, world: int) -> bytes:
        """Allocates this rank's inbox; returns its 64-byte IPC handle, to be all-gathered by the caller."""
        buf = C.create_string_buffer(64)
        self._check(self._lib.icp_exchange_create(self._h, int(rank), int(world), buf))
        return bytes(buf.raw)

    def exchange_connect(self, handles):
        """`handles`: the `world` handles in rank order.  From here on every registration on this context exchanges its
        normal equations with the peers inside the library (all ranks must issue the same registrations)."""
        blob = b"".join(bytes(h) for h in handles)
        self._check(self._lib.icp_exchange_connect(self._h, C.c_char_p(blob)))

    def exchange_destroy(self):
        self._check(self._lib.icp_exchange_destroy(self._h))

    # ---- multi-GPU: map-sharded normals ------------------------------------------------------------------------------
    def map_normals_owned(self, rank: int, world: int) -> torch.Tensor:
        """[M,4] float32 device tensor: (nx, ny, nz, 1) at the original index of every map point whose spatial bucket
        this rank owns, zeros elsewhere — to be summed over the ranks and handed to `map_normals_install`."""
        self.use_torch_stream()
        out = torch.empty((self.map_size(), 4), dtype=torch.float32, device=self.device)
        self._check(self._lib.icp_map_normals_owned(self._h, int(rank), int(world), out.data_ptr()))
        return out

    def map_normals_install(self, normals_by_index: torch.Tensor):
        self.use_torch_stream()
        t = normals_by_index.to(self.device, torch.float32).contiguous()
        if tuple(t.shape) != (self.map_size(), 4):
            raise AssertionError(f"expected [{self.map_size()}, 4] normals, got {tuple(t.shape)}")
        self._check(self._lib.icp_map_normals_install(self._h, t.data_ptr()))

    # ---- multi-GPU seam ----------------------------------------------------------------------------------------------
    def normal_equations_tensor(self) -> torch.Tensor:
        """A torch-owned [32] f64 device vector installed as the context's normal-equation buffer, so that
        `torch.distributed.all_reduce` (RCCL) can sum it in place between accumulate() and solve()."""
        self.use_torch_stream()  # the collective that sums this vector is ordered on torch's stream
        if self._neq_tensor is None:
            self._neq_tensor = torch.zeros(32, dtype=torch.float64, device=self.device)
            self._check(self._lib.icp_set_normal_equations_buffer(self._h, self._neq_tensor.data_ptr()))
        return self._neq_tensor

    def register_begin(self, points: Array, init_pose=None, skip_null: bool = False):
        self._bind(points)
        p, mem, keep = _ptr_mem(points)
        self._keep_targets = [keep]
        init = _pose16(init_pose if init_pose is not None else np.eye(4))
        self._check(self._lib.icp_register_begin(self._h, p, int(keep.shape[0]), mem,
                                                 TARGETS_SKIP_NULL if skip_null else TARGETS_ALL, init))

    def register_launch(self, points: Array, init_pose=None, skip_null: bool = False):
        """Enqueue a whole registration without waiting; `register_end()` later blocks on it alone, so work enqueued in
        between (e.g. `map_update(None)`) overlaps the host's wait.  `init_pose="last"`: the initial guess is the pose of
        the previous registration, read on the device (constant-velocity initialisation without a host round trip) — with
        it up to two registrations may be in flight, `register_end()` returning them oldest first."""
        self._bind(points)
        p, mem, keep = _ptr_mem(points)
        self._keep_targets = (getattr(self, "_keep_targets", None) or [])[-1:] + [keep]
        mode = TARGETS_SKIP_NULL if skip_null else TARGETS_ALL
        if isinstance(init_pose, str):
            if init_pose != "last":
                raise AssertionError(f"unknown initial pose {init_pose!r}")
            self._check(self._lib.icp_register_launch_from_last(self._h, p, int(keep.shape[0]), mem, mode))
            return
        init = _pose16(init_pose if init_pose is not None else np.eye(4))
        self._check(self._lib.icp_register_launch(self._h, p, int(keep.shape[0]), mem, mode, init))

    def iteration_accumulate(self):
        self._check(self._lib.icp_iteration_accumulate(self._h))

    def iteration_solve(self):
        self._check(self._lib.icp_iteration_solve(self._h))

    def register_end(self) -> RegisterResult:
        cap = max(1, int(self.config.max_num_alignments))
        losses = (C.c_double * cap)()
        dxs = (C.c_float * (6 * cap))()
        res = IcpRegisterResult()
        self._check(self._lib.icp_register_end(self._h, C.byref(res), losses, dxs))
        return self._result(res, losses, dxs)

    def raise_for_status(self, rc: int):
        """Maps a status code of the C ABI to the exception the reference raises (public form of the internal check)."""
        self._check(rc)

    # ---- profiling ---------------------------------------------------------------------------------------------------
    def profile_enable(self, mask: int = 1):
        """bit mask of the kernels timed with HIP events: 1 search, 2 reduction, 4 normals (0 = off)."""
        self._check(self._lib.icp_profile_enable(self._h, int(mask)))

    def profile_read(self):
        s, n, r, m = C.c_double(0), C.c_int64(0), C.c_double(0), C.c_double(0)
        self._check(self._lib.icp_profile_read(self._h, C.byref(s), C.byref(n), C.byref(r), C.byref(m)))
        return {"search_ms": s.value, "search_launches": int(n.value), "reduce_ms": r.value, "normals_ms": m.value}

    def profile_read_iterations(self, cap: int = 64):
        """(ms, launches) of the search kernel by ICP iteration index (`icp_profile_read_iterations`)."""
        ms = np.zeros(cap, np.float64)
        n = np.zeros(cap, np.int64)
        self._check(self._lib.icp_profile_read_iterations(self._h, ms.ctypes.data, n.ctypes.data, int(cap)))
        return ms, n

    def profile_event_floor(self, samples: int = 200) -> float:
        """What an event pair adds (us, median) to the launch it brackets on this context's stream: measured around a kernel
        that spins for exactly 20 us, minus those 20 us (`icp_profile_event_floor`)."""
        v = C.c_double(0)
        self._check(self._lib.icp_profile_event_floor(self._h, int(samples), C.byref(v)))
        return float(v.value)


class IcpBatch:
    """B independent sequences advanced by ONE launch per ICP iteration (`icp_batch_*`, include/icp_mi355x.h): B
    `IcpContext`s of one device — each with its own local map, scan and registration state, i.e. B instances of the
    reference's `ICPFrameToModel` (slam/odometry/icp_odometry.py:248-299 registers one sequence, one frame at a time) —
    whose registrations are enqueued together by one host thread.  Per sequence the same poses, bit for bit, as
    `IcpContext.register_launch` / `map_update(None)` / `register_end` on that context alone."""

    def __init__(self, contexts):
        self.contexts = list(contexts)
        if not self.contexts:
            raise AssertionError("a batch needs at least one context")
        self._lib = self.contexts[0]._lib
        arr = (C.c_void_p * len(self.contexts))(*[c._h for c in self.contexts])
        handle = C.c_void_p()
        rc = self._lib.icp_batch_create(arr, len(self.contexts), C.byref(handle))
        if rc != 0:
            raise AssertionError(f"icp_batch_create failed ({STATUS_MESSAGES.get(rc, rc)}): 1..{_lib.BATCH_MAX_SEQUENCES} distinct "
                                 "contexts of one device")
        self._h = handle
        self._keep = None

    def close(self):
        if getattr(self, "_h", None):
            self._lib.icp_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return len(self.contexts)

    def _check(self, rc: int):
        if rc == 0:
            return
        msg = self._lib.icp_batch_last_error(self._h).decode() or STATUS_MESSAGES.get(rc, str(rc))
        if rc == _lib.ICP_ERR_INVALID_JACOBIAN:
            raise InvalidJacobianError("Invalid Jacobian in Gauss Newton minimization")
        if rc == _lib.ICP_ERR_INVALID_ARGUMENT:
            raise AssertionError(msg)
        raise RuntimeError(f"libicp_mi355x: {msg} ({rc})")

    def use_torch_stream(self):
        """Every member enqueues on torch's current stream of the batch's device."""
        c0 = self.contexts[0]
        stream = _RAW_STREAM(c0._device_index) if _RAW_STREAM is not None else \
            torch.cuda.current_stream(c0.device).cuda_stream
        if stream != getattr(self, "_bound_stream", None) or any(getattr(c, "_bound_stream", None) != stream
                                                                 for c in self.contexts):
            self._check(self._lib.icp_batch_set_stream(self._h, C.c_void_p(stream)))
            self._bound_stream = stream
            for c in self.contexts:
                c._bound_stream = stream

    def register_launch(self, scans, init_poses=None, skip_null: bool = False):
        """`scans[b]`: member b's scan ([n_b, 3] float32; all cuda tensors or all host arrays).  `init_poses`: a list of
        4x4 matrices (None entries: identity), None (identity for all) or "last" (every member starts from the device-resident
        pose of its previous registration: constant-velocity initialisation without a host round trip)."""
        if len(scans) != len(self.contexts):
            raise AssertionError(f"expected {len(self.contexts)} scans, got {len(scans)}")
        if _on_device(*scans):
            self.use_torch_stream()
        ptrs, ns, keep, mems = [], [], [], set()
        for a in scans:
            p, mem, k = _ptr_mem(a)
            ptrs.append(p)
            ns.append(int(k.shape[0]))
            keep.append(k)
            mems.add(mem)
        if len(mems) != 1:
            raise AssertionError("the scans of a batch must live in one memory space")
        mem = mems.pop()
        self._keep = (getattr(self, "_keep", None) or [])[-1:] + [keep]
        xyz = (C.c_void_p * len(ptrs))(*ptrs)
        n = (C.c_int64 * len(ns))(*ns)
        mode = TARGETS_SKIP_NULL if skip_null else TARGETS_ALL
        if isinstance(init_poses, str):
            if init_poses != "last":
                raise AssertionError(f"unknown initial pose {init_poses!r}")
            self._check(self._lib.icp_batch_register_launch(self._h, xyz, n, mem, mode, None, 1))
            return
        init = None
        if init_poses is not None:
            flat = np.stack([np.asarray(m if m is not None else np.eye(4), dtype=np.float32).reshape(16)
                             for m in init_poses])
            if flat.shape[0] != len(self.contexts):
                raise AssertionError("one initial pose per member")
            init = np.ascontiguousarray(flat)
        self._check(self._lib.icp_batch_register_launch(self._h, xyz, n, mem, mode,
                                                        init.ctypes.data if init is not None else None, 0))

    def project(self, scans, outs):
        """`IcpContext.project(scans[b], out=outs[b])` for every member in two launches (cuda tensors: [n_b, 3] float32 scans,
        [3, H, W] float32 vertex maps)."""
        if len(scans) != len(self.contexts) or len(outs) != len(self.contexts):
            raise AssertionError("one scan and one vertex map per member")
        self.use_torch_stream()
        keep = []
        for a, o in zip(scans, outs):
            p, mem, k = _ptr_mem(a)
            if mem != MEM_DEVICE or not (isinstance(o, torch.Tensor) and o.is_cuda and o.dtype == torch.float32
                                         and o.is_contiguous()):
                raise AssertionError("batched projection takes cuda tensors (float32, contiguous)")
            keep.append(k)
        xyz = (C.c_void_p * len(keep))(*[k.data_ptr() for k in keep])
        n = (C.c_int64 * len(keep))(*[int(k.shape[0]) for k in keep])
        vm = (C.c_void_p * len(outs))(*[o.data_ptr() for o in outs])
        self._check(self._lib.icp_batch_project(self._h, xyz, n, vm))
        return outs

    def map_update(self):
        """`map_update(None)` on every member: the pose-only update by the device-resident pose of its registration."""
        self._check(self._lib.icp_batch_map_update(self._h))

    def register_end(self):
        """One wait for all members; a list of `RegisterResult`s (raises what the first failing member would raise)."""
        b = len(self.contexts)
        cap = max(1, int(self.contexts[0].config.max_num_alignments))
        res = (IcpRegisterResult * b)()
        losses = (C.c_double * (cap * b))()
        dxs = (C.c_float * (6 * cap * b))()
        rc = self._lib.icp_batch_register_end(self._h, res, losses, dxs)
        self._check(rc)
        la = np.array(losses, np.float64).reshape(b, cap)
        da = np.array(dxs, np.float32).reshape(b, cap, 6)
        out = []
        for i in range(b):
            k = int(res[i].iterations)
            out.append(RegisterResult(np.array(res[i].pose, np.float32).reshape(4, 4), np.array(res[i].params, np.float32), k,
                                      bool(res[i].converged), int(res[i].num_targets), int(res[i].normals_computed),
                                      la[i, :k].copy(), da[i, :k].copy()))
        return out
