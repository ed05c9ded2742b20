"""CPU, world size 2, gloo: the N > 1 driver of the hot path (pylidar_slam_amd.distributed) — sharding of the target
points, all-reduce of the packed normal equations, identical solve on every rank.  The per-rank compute engine is the
numpy oracle behind the engine protocol of `IcpContext` (on the GPU box the same driver runs on the HIP engine over
RCCL; `tests/test_gpu_parity.py::test_split_iteration_seam_equals_fused_register` pins the HIP side of the seam)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleEngine:
    """Engine protocol (register_begin / iteration_accumulate / normal_equations_tensor / iteration_solve /
    register_end) implemented with oracle/icp_oracle.py — test infrastructure."""

    def __init__(self, model, iters, scheme="geman_mcclure", sigma=0.3):
        import icp_oracle as O
        self.O = O
        self.lm = O.KdTreeLocalMapOracle(workers=1)
        self.lm.set_map_pointcloud(model)
        self.scheme, self.sigma, self.iters = scheme, sigma, iters
        self.neq = torch.zeros(32, dtype=torch.float64)

        class _Cfg:
            max_num_alignments = iters
        self.config = _Cfg()

    def normal_equations_tensor(self):
        return self.neq

    def register_begin(self, pts, init_pose=None, skip_null=False):
        self.pts = np.asarray(pts, np.float32)
        self.pose = np.eye(4, dtype=np.float32) if init_pose is None else np.asarray(init_pose, np.float32)
        self.params = np.zeros(6, np.float32)
        self.losses = []

    def iteration_accumulate(self):
        O = self.O
        v = np.zeros(32)
        if self.pts.shape[0]:
            p = O.apply_transformation(self.pts, self.pose)
            q, n, _ = self.lm.nearest_neighbor_search(p)
            res, jac = O.point_to_plane_rows(p, q, n)
            w = O.ls_weights(self.scheme, self.sigma, res, p, q)
            rw = (res * w).astype(np.float64)
            jw = (jac * w.reshape(-1, 1)).astype(np.float64)
            H = jw.T @ jw
            v[:21] = H[np.triu_indices(6)]
            v[21:27] = jw.T @ rw
            v[27] = (rw * rw).sum()
            v[28] = (res.astype(np.float64) ** 2).sum()
            v[29] = self.pts.shape[0]
        self.neq.copy_(torch.from_numpy(v))

    def iteration_solve(self):
        O = self.O
        v = self.neq.numpy()
        H = np.zeros((6, 6))
        H[np.triu_indices(6)] = v[:21]
        H = H + np.triu(H, 1).T
        dx = -np.linalg.solve(H, v[21:27]).astype(np.float32)
        self.losses.append(v[27])
        self.params = O.from_pose_matrix((O.build_pose_matrix(dx) @ self.pose).astype(np.float32))
        self.pose = O.build_pose_matrix(self.params)

    def register_end(self):
        return self.pose.copy(), np.array(self.losses)

    # map-sharded normals (engine protocol of `sharded_map_normals`): ownership by index parity stands in for the spatial
    # bucket hash of the HIP engine — the driver only relies on "every index has exactly one owner"
    def map_normals_owned(self, rank, world):
        m = self.lm.model.shape[0]
        out = torch.zeros((m, 4), dtype=torch.float32)
        mine = np.arange(m)[np.arange(m) % world == rank]
        out[mine, :3] = torch.from_numpy(self.O.knn_normals(self.lm.model, self.lm.tree, mine, self.lm.k)
                                         .astype(np.float32))
        out[mine, 3] = 1.0
        return out

    def map_normals_install(self, normals_by_index):
        self.installed = normals_by_index.clone()


def _workload():
    from pylidar_slam_amd.synthetic import SceneConfig, make_sequence
    import icp_oracle as O
    scans, _ = make_sequence(SceneConfig(height=16, width=256), 2)
    model, _ = O.grid_sample(scans[0], 0.2)
    return model, scans[1]


def _worker(rank, world, port, out):
    for p in (os.path.join(ROOT, "pylidar-slam_amd"), os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pylidar_slam_amd.distributed import shard_bounds, sharded_map_normals, sharded_register
    model, scan = _workload()
    b, e = shard_bounds(scan.shape[0], world, rank)
    eng = OracleEngine(model, iters=4)
    sharded_map_normals(eng)  # owner-computed normals, summed by original index over the ranks
    assert float(eng.installed[:, 3].min()) == 1.0 and float(eng.installed[:, 3].max()) == 1.0
    full = torch.from_numpy(eng.O.knn_normals(eng.lm.model, eng.lm.tree, np.arange(model.shape[0]), eng.lm.k)
                            .astype(np.float32))
    assert torch.equal(eng.installed[:, :3], full)  # exactly the single-process normals on every rank
    pose, losses = sharded_register(eng, scan[b:e], None, 4)
    gathered = [torch.zeros(16, dtype=torch.float32) for _ in range(world)]
    dist.all_gather(gathered, torch.from_numpy(pose.reshape(-1).copy()))
    if rank == 0:
        np.savez(out, pose=pose, losses=losses, all=np.stack([g.numpy() for g in gathered]))
    dist.destroy_process_group()


def test_shard_bounds_tile_the_scan():
    from pylidar_slam_amd.distributed import shard_bounds
    for n in (0, 1, 7, 131072, 131073):
        for w in (1, 2, 3, 8):
            cuts = [shard_bounds(n, w, r) for r in range(w)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(w - 1))
            assert max(e - b for b, e in cuts) - min(e - b for b, e in cuts) <= (n + w - 1) // w
    with pytest.raises(AssertionError):
        shard_bounds(10, 2, 2)


@pytest.mark.parametrize("world", [2, 8])  # (8: the node the driver's scaling runs use — uneven shards, 8-way all-reduce)
def test_sharded_registration_matches_single_process(tmp_path, world):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "rank0.npz")
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    got = np.load(out)
    # every rank ends with the identical pose (same all-reduced vector, same solve)
    assert all(np.array_equal(got["all"][0], got["all"][r]) for r in range(1, world))
    # and it is the single-process registration of the whole scan
    from pylidar_slam_amd.distributed import sharded_register
    model, scan = _workload()
    eng = OracleEngine(model, iters=4)
    pose, losses = sharded_register(eng, scan, None, 4)
    np.testing.assert_allclose(got["pose"], pose, atol=1e-6)
    np.testing.assert_allclose(got["losses"], losses, rtol=1e-9)
