"""`OracleContext`: the handful of `IcpContext` calls the host-side plugin code makes, computed by oracle/icp_oracle.py
(numpy, CPU).  TEST INFRASTRUCTURE: it lets the CPU suite drive the plugin's host logic — and the reference's own SLAM
loop around it — where no MI355X is present; the same code runs on the real `IcpContext` in tests/test_gpu_parity.py.
Never imported by the package."""
import numpy as np
import torch

import icp_oracle as O


class OracleContext:
    """`IcpContext` protocol used by MI355XICPFrameToModel / HashGridLocalMap / the alignment seams / the filters."""
    device = torch.device("cpu")

    def __init__(self, height=64, width=1024, up_fov=3.0, down_fov=-24.0, max_num_alignments=100,
                 threshold_delta_pose=1.0e-4, scheme="default", sigma=0.5, local_map_size=20, num_neighbors_normals=10,
                 **kwargs):
        self.hw = (int(height), int(width), float(up_fov), float(down_fov))
        self.scheme, self.sigma = scheme, float(sigma)
        self.lm = O.KdTreeLocalMapOracle(local_map_size, num_neighbors_normals, workers=1)
        self.cfg = O.ICPOracleConfig(
            max_num_alignments=max_num_alignments, threshold_delta_pose=threshold_delta_pose, scheme=scheme, sigma=sigma,
            height=height, width=width, local_map_size=local_map_size)
        self.reg = O.ICPFrameToModelOracle(self.cfg)
        self.reg.local_map = self.lm
        self.calls = []

    def use_torch_stream(self):
        pass

    def set_cost(self, mode):
        self.cfg.alignment = "point_to_point" if "point_to_point" in mode else "point_to_plane"

    @staticmethod
    def _np(a):
        return a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)

    def project(self, pc, **kwargs):
        h, w, up, down = self.hw
        return torch.from_numpy(O.build_projection_map(self._np(pc).reshape(-1, 3).astype(np.float32), h, w, up, down))

    def grid_sample(self, pc, voxel):
        pts, idx = O.grid_sample(self._np(pc), voxel)
        return pts, idx

    def grid_sample_f64(self, pc, voxel):
        return O.grid_sample(self._np(pc).astype(np.float64), voxel)

    def distort(self, pc, timestamps, rpose):
        return O.distort(self._np(pc), self._np(timestamps), np.asarray(rpose))

    def map_init(self):
        self.lm.init()

    def map_set(self, points):
        self.lm.set_map_pointcloud(self._np(points).reshape(-1, 3).astype(np.float32))

    def map_points(self):
        return self.lm.local_map

    def map_update(self, rel_pose, new_points=None, skip_null=False):
        self.calls.append("insert" if new_points is not None else "move")
        pts = None if new_points is None else self._np(new_points).reshape(-1, 3)
        if pts is not None and skip_null:
            pts = pts[np.abs(pts).max(axis=1) > 0]
        self.lm.update(np.asarray(rel_pose, np.float32), pts)
        return 0 if pts is None else int(pts.shape[0])

    # the two-step update around a registration (icp_map_stage_cloud / icp_map_update_staged): the staged rows are kept —
    # a copy: the caller may reuse its buffer — until an update consumes them or the next staging replaces them
    def map_stage_cloud(self, new_points, skip_null=False):
        self.calls.append("stage")
        pts = self._np(new_points).reshape(-1, 3).copy()
        pts = pts[~np.isnan(pts).any(axis=1)]  # (the library's k_flag_not_nan always drops NaN rows)
        self._staged = pts[np.abs(pts).max(axis=1) > 0] if skip_null else pts

    def map_update_staged(self, rel_pose):
        assert getattr(self, "_staged", None) is not None, "no staged cloud (icp_map_stage_cloud)"
        pts, self._staged = self._staged, None
        self.calls.append("insert_staged")
        self.lm.update(np.asarray(rel_pose, np.float32), pts)
        return int(pts.shape[0])

    def map_update_vertex_map(self, rel_pose, vmap):
        self.calls.append("insert_vmap")
        self.lm.update(np.asarray(rel_pose, np.float32), None, self._np(vmap))
        return int(self.lm.num_elements[-1])

    def nearest_neighbor_search(self, points, with_normals=True, with_index=False):
        q, n, idx = self.lm.nearest_neighbor_search(self._np(points).reshape(-1, 3).astype(np.float32))
        return q, (n if with_normals else None), (idx.astype(np.int32) if with_index else None)

    def align_point_to_plane(self, ref, tgt, normals, with_residuals=False):
        r, t, n = (self._np(x).reshape(-1, 3).astype(np.float32) for x in (ref, tgt, normals))
        step = O.gauss_newton_step(t, r, n, self.scheme, self.sigma)
        out = (O.build_pose_matrix(step.dx), step.dx, step.loss, None)
        if not with_residuals:
            return out
        res, _ = O.point_to_plane_rows(t, r, n)
        w = O.ls_weights(self.scheme, self.sigma, res, t, r)
        return out + (((res * w) ** 2).astype(np.float32),)

    def register(self, points, init_pose=None, skip_null=False):
        from pylidar_slam_amd.engine import RegisterResult
        pts = self._np(points).reshape(-1, 3).astype(np.float32)
        pts = pts[~np.isnan(pts).any(axis=1)]
        if skip_null:
            pts = pts[np.abs(pts).max(axis=1) > 0]
        init = np.eye(4, dtype=np.float32) if init_pose is None else np.asarray(init_pose, np.float32)
        params, pose = self.reg.register_new_frame(pts, init)
        tr = self.reg.traces[-1]
        return RegisterResult(pose, params, len(tr.dx), False, pts.shape[0], 0, np.array(tr.loss), np.array(tr.dx))

    # the asynchronous hand-off of the plugin's frame loop: here the "launch" just remembers its arguments
    def register_launch(self, points, init_pose=None, skip_null=False):
        assert getattr(self, "_pending", None) is None, "collect the pending result first"
        self._pending = (points, init_pose, skip_null)

    def register_end(self):
        points, init_pose, skip_null = self._pending
        self._pending = None
        return self.register(points, init_pose, skip_null)
