"""CPU test of the HOST logic of the odometry plugin (`MI355XICPFrameToModel.do_process_next_frame`: input dispatch,
first-frame map, key-frame thresholds, pose chain, outputs written to the data_dict) with the numpy oracle standing in for
the HIP context — the same loop runs on the real `IcpContext` in tests/test_gpu_parity.py.  The stand-in implements the
handful of `IcpContext` calls the plugin makes; it is test infrastructure and lives here, not in the package."""
import numpy as np
import pytest
import torch

import icp_oracle as O


from oracle_context import OracleContext  # noqa: E402


@pytest.mark.parametrize("run", ["A_numpy_ls", "B_tensor_gm"])
def test_frame_loop_host_logic_matches_reference(monkeypatch, golden_c1, c1_scans, run):
    import pylidar_slam_amd.odometry as odo_mod
    monkeypatch.setattr(odo_mod, "IcpContext", OracleContext)
    g = golden_c1
    scans, _ = c1_scans
    mode, scheme, sigma, iters, thr = (str(v) for v in g[f"{run}_cfg"])
    h, w = (int(v) for v in g["hw"])
    cfg = odo_mod.MI355XICPConfig(
        max_num_alignments=int(iters), threshold_delta_pose=float(thr), data_key="input_data",
        alignment=dict(mode="point_to_plane_gauss_newton",
                       gauss_newton_config=dict(max_iters=1, scheme=scheme, sigma=float(sigma))))
    odo = odo_mod.MI355XICPFrameToModel(cfg, projector=odo_mod.SphericalProjector(h, w), device=torch.device("cpu"))
    odo.init()
    cv = O.ConstantVelocityOracle()
    for f, s in enumerate(scans):
        pts, _ = O.grid_sample(s, 0.3)
        d = {"input_data": pts if mode == "numpy" else torch.from_numpy(pts), "init_rpose": cv.next_initial_pose()}
        odo.process_next_frame(d)
        if f == 0:  # frame 0 writes nothing (icp_odometry.py:171-181)
            assert "odometry_pose" not in d and "odometry_pc" not in d
            continue
        pose = d["odometry_pose"]
        assert pose.shape == (4, 4) and pose.dtype == np.float32 and d["odometry_pc"].shape[1] == 3
        cv.save_real_motion(pose)
        dt, dr = O.pose_error(pose, g[f"{run}_rel"][f])
        assert dt < 2e-5 and dr < 2e-5, (f, dt, dr)
    rel = odo.get_relative_poses()
    assert rel.shape == (len(scans), 4, 4) and np.array_equal(rel[0], np.eye(4, dtype=np.float32))
    np.testing.assert_allclose(np.stack(odo.absolute_poses), g[f"{run}_abs"], atol=1e-4)
    # the first frame goes in as a vertex map, 0.4 m per frame is above the 0.1 m key-frame threshold: every later
    # frame is inserted as a cloud; the map ends with the reference's size
    # (frames of up to `stage_insert_max_rows` rows are staged for the map in front of their registration and inserted
    # from there: icp_map_stage_cloud / icp_map_update_staged)
    calls = odo.ctx.calls
    assert calls[0] == "insert_vmap" and calls[1:] == ["stage", "insert_staged"] * (len(scans) - 1)
    assert odo.ctx.lm.local_map.shape[0] == int(g[f"{run}_map_size"])
    assert len(odo.elapsed) == len(scans) and odo.get_elapsed() > 0


def test_small_motion_is_a_pose_only_update(monkeypatch):
    """Below both key-frame thresholds (0.1 m / 0.3 deg) the map is only re-expressed; the deltas accumulate until the
    threshold is crossed (icp_odometry.py:360-380)."""
    import pylidar_slam_amd.odometry as odo_mod
    monkeypatch.setattr(odo_mod, "IcpContext", OracleContext)
    from pylidar_slam_amd.synthetic import SceneConfig, ray_directions, render_scan, pose_matrix
    sc = SceneConfig(height=16, width=256)
    dirs = ray_directions(sc)
    poses = [pose_matrix(np.array([0.04 * k, 0, 0, 0, 0, 0.0])) for k in range(5)]  # 4 cm per frame
    cfg = odo_mod.MI355XICPConfig(max_num_alignments=10, threshold_delta_pose=1e-4, data_key="numpy_pc")
    odo = odo_mod.MI355XICPFrameToModel(cfg, projector=odo_mod.SphericalProjector(16, 256), device=torch.device("cpu"))
    odo.init()
    for k, p in enumerate(poses):
        odo.process_next_frame({"numpy_pc": render_scan(sc, p, k, dirs)})
    # frames 1, 2: 4 and 8 cm since the last insertion -> pose-only; frame 3: 12 cm -> insertion; frame 4: 4 cm again
    # (every frame is staged in front of its registration; a pose-only update leaves the staged rows unused, the next
    # staging replaces them)
    assert odo.ctx.calls == ["insert_vmap", "stage", "move", "stage", "move", "stage", "insert_staged", "stage", "move"]
    # frames above the row limit are inserted the classic way
    cfg2 = odo_mod.MI355XICPConfig(max_num_alignments=10, threshold_delta_pose=1e-4, data_key="numpy_pc",
                                   stage_insert_max_rows=100)
    odo2 = odo_mod.MI355XICPFrameToModel(cfg2, projector=odo_mod.SphericalProjector(16, 256), device=torch.device("cpu"))
    odo2.init()
    for k, p in enumerate(poses):
        odo2.process_next_frame({"numpy_pc": render_scan(sc, p, k, dirs)})
    assert odo2.ctx.calls == ["insert_vmap", "move", "move", "insert", "move"]
    np.testing.assert_array_equal(odo2.get_relative_poses(), odo.get_relative_poses())
    np.testing.assert_array_equal(odo2.ctx.lm.local_map, odo.ctx.lm.local_map)
    with pytest.raises(AssertionError):
        odo.process_next_frame({"wrong_key": None})


# ---- filters: the pass-through / error logic around the GPU calls --------------------------------------------------
class _FilterContext:
    """distort / grid_sample of the `IcpContext` protocol by the oracle."""

    def distort(self, pc, ts, rpose):
        return O.distort(np.asarray(pc, np.float32), np.asarray(ts, np.float64).reshape(-1), np.asarray(rpose, np.float64))

    def grid_sample(self, pc, voxel):
        pts, idx = O.grid_sample(np.asarray(pc), voxel)
        return pts, idx

    def grid_sample_f64(self, pc, voxel):
        return self.grid_sample(pc, voxel)


def test_distortion_filter_pass_through_rules():
    from pylidar_slam_amd.odometry import Distortion, DistortionConfig, GridSample, GridSampleConfig
    from pylidar_slam_amd.synthetic import pose_matrix
    rng = np.random.default_rng(0)
    pc = rng.normal(size=(50, 3)).astype(np.float32)
    ts = np.linspace(0.0, 0.1, 50)
    rpose = pose_matrix(np.array([0.5, 0, 0, 0, 0, 0.02]))
    ctx = _FilterContext()
    f = Distortion(DistortionConfig(), ctx=ctx)
    # no timestamps / no initial pose / deactivated: the SAME array object goes through (preprocessing.py:157-162)
    for d in ({"numpy_pc": pc, "init_rpose": rpose},
              {"numpy_pc": pc, "numpy_pc_timestamps": ts, "init_rpose": None}):
        f.filter(d)
        assert d["input_data"] is pc
    d = {"numpy_pc": pc, "numpy_pc_timestamps": ts, "init_rpose": rpose}
    Distortion(DistortionConfig(activate=False), ctx=ctx).filter(d)
    assert d["input_data"] is pc
    # active: float64 de-skewed cloud, first point (alpha = 0) untouched, last point moved by the whole motion
    d = {"numpy_pc": pc, "numpy_pc_timestamps": ts, "init_rpose": rpose}
    f.filter(d)
    out = d["input_data"]
    assert out.dtype == np.float64 and out.shape == pc.shape
    np.testing.assert_allclose(out[0], pc[0], atol=1e-7)
    np.testing.assert_allclose(out[-1], rpose[:3, :3] @ pc[-1] + rpose[:3, 3], atol=1e-6)
    with pytest.raises(AssertionError):
        f.filter({"numpy_pc": torch.from_numpy(pc), "numpy_pc_timestamps": ts, "init_rpose": rpose})
    # GridSample on the float64 output keeps float64 samples and int64 indices
    g = GridSample(GridSampleConfig(voxel_size=0.5, pointcloud_key="input_data"), ctx=ctx)
    g.filter(d)
    assert d["sample_points"].dtype == np.float64 and d["sample_indices"].dtype == np.int64
    np.testing.assert_array_equal(d["sample_points"], out[d["sample_indices"]])
    with pytest.raises(AssertionError):
        g.filter({"input_data": np.zeros((5, 2), np.float32)})


def test_bench_gpus_n_without_gpus_reports_instead_of_asking_for_a_launcher():
    """`python bench.py --gpus N` starts its own ranks (self_launch); with fewer than N visible GPUs under the RCCL backend
    it prints one JSON line that says so and exits 2 — it does not ask the caller for a launcher any more."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "BENCH_DIST_BACKEND"):
        env.pop(k, None)
    env["HIP_VISIBLE_DEVICES"] = ""  # no device on any box this test runs on
    env["CUDA_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 2, (r.returncode, r.stderr[-1000:])
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 2 and "needs 2 visible GPUs" in d["error"]


def test_bench_sequence_threads_get_their_inputs_from_the_constructing_thread():
    """bench.py's throughput arrangement (S sequences on S host threads): the scans and the map of every sequence are
    generated when its `SequenceThread` is CONSTRUCTED — one after the other on the caller's thread — not inside the
    threads: generated concurrently on the GPU box's 256-thread host they came out different in one run in four (numpy /
    BLAS rounding under concurrency; a ray that grazes a box edge then lands metres away), and the thread-driven sequences
    returned other poses than in the other runs (round 5).  Same generator, same seed -> the same bytes."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    argv = sys.argv
    try:
        sys.argv = ["bench.py"]
        import bench
        args = bench.parse()
    finally:
        sys.argv = argv
    t = bench.SequenceThread(args, 101, 0, frames=4)
    assert not t.is_alive()  # (constructed, not started: nothing here needs a GPU)
    scans, poses, model, order, start = t.workload
    again = bench.make_workload(101, args.trajectory, 4)
    assert sorted(scans) == sorted(again[0])
    for f in scans:
        np.testing.assert_array_equal(scans[f], again[0][f])
    np.testing.assert_array_equal(model, again[2])
    assert order == again[3] and start == again[4]
