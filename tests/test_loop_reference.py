"""The reference's PUBLISHED configuration as a full odometry loop (stand-in for BASELINE.json configs[2]; SURVEY §8 row
f3): `tests/golden/loop_reference.npz` holds the trajectory the reference's own `SLAM` loop produced on 36 seeded
64x2048 synthetic frames with `CV + kd-tree F2M, neighborhood sigma 0.2, <= 20 iterations (threshold 1e-4), map of 30
key frames, grid sample 0.4 m` (docs/results/KITTI/kitti_benchmark.md:19; generator: oracle/make_golden_loop.py).

CPU part (this file, no GPU): the fixture is self-consistent under `pylidar_slam_amd.eval` (the reference's ATE / ARE /
segment errors are reproduced from the stored poses), and the plugin's host logic — driven by the numpy oracle context —
follows the reference's trajectory over the first frames.  The `-m gpu` part (tests/test_gpu_loop.py) runs all 36 frames
through the HIP library."""
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN


@pytest.fixture(scope="module")
def golden_loop():
    return np.load(os.path.join(GOLDEN, "loop_reference.npz"))


@pytest.fixture(scope="module")
def loop_scans(golden_loop):
    from pylidar_slam_amd.synthetic import SceneConfig, make_sequence
    h, w = (int(v) for v in golden_loop["hw"])
    scans, gt = make_sequence(SceneConfig(height=h, width=w), len(golden_loop["scan_sha"]))
    for s, ref in zip(scans, golden_loop["scan_sha"]):
        if hashlib.sha1(np.ascontiguousarray(s).tobytes()).hexdigest() != str(ref):
            pytest.fail("the seeded synthetic generator no longer reproduces the frames the reference's loop was run on "
                        "(tests/golden/loop_reference.npz holds their sha1)")
    np.testing.assert_array_equal(gt, golden_loop["gt_abs"])
    return scans, gt


def published_config(**over):
    from pylidar_slam_amd.odometry import MI355XICPConfig
    kw = dict(max_num_alignments=20, threshold_delta_pose=1.0e-4, data_key="input_data",
              local_map=dict(type="kdtree_local_map", local_map_size=30, num_neighbors_normals=10),
              alignment=dict(mode="point_to_plane_gauss_newton",
                             gauss_newton_config=dict(max_iters=1, scheme="neighborhood", sigma=0.2)))
    kw.update(over)
    return MI355XICPConfig(**kw)


def trajectory_metrics(rel, gt_abs, segments):
    from pylidar_slam_amd import eval as ev
    gt_rel = ev.compute_relative_poses(gt_abs)
    gt_rel[0] = np.eye(4)
    est_abs, gt0 = ev.compute_absolute_poses(np.asarray(rel, np.float64)), ev.compute_absolute_poses(gt_rel)
    ate, _ = ev.compute_ate(np.asarray(rel, np.float64), gt_rel)
    are, _ = ev.compute_are(np.asarray(rel, np.float64), gt_rel)
    tr, rot, errors = ev.compute_kitti_metrics(est_abs, gt0, list(segments), step_size=10)
    return ate, are, tr, rot, len(errors)


def test_fixture_metrics_are_reproduced_by_eval(golden_loop):
    g = golden_loop
    for pre in ("", "forced_"):
        ate, are, tr, rot, n = trajectory_metrics(g[pre + "rel"], g["gt_abs"], g["segments"])
        assert n == int(g[pre + "num_segments"]) > 0
        np.testing.assert_allclose([ate, are], [g[pre + "ate"][0], g[pre + "are"][0]], rtol=1e-9)
        np.testing.assert_allclose([tr, rot], g[pre + "kitti"], rtol=1e-9)
    assert int(g["forced_iters"][1:].min()) == int(g["forced_iters"][1:].max()) == int(g["forced_iters_per_frame"])
    # the run the fixture holds: an insertion per frame, evictions from frame 30 on, live convergence well below the cap
    assert g["map_sizes"][29] > g["map_sizes"][30] and int(g["iters"][1:].max()) < 20 and g["ate"][0] < 5e-3


def test_plugin_host_logic_follows_the_reference_loop(golden_loop, loop_scans, monkeypatch):
    """First 6 frames through the plugin with the numpy oracle context behind it (host grid sample -> tensor -> ICP, the
    reference's own preprocessing order): poses within 1e-4 m / 1e-4 rad of the reference's run, same iteration counts."""
    import torch
    import icp_oracle as O
    from oracle_context import OracleContext
    from pylidar_slam_amd import odometry as our
    monkeypatch.setattr(our, "IcpContext", OracleContext)
    scans, _ = loop_scans
    g = golden_loop
    odo = our.MI355XICPFrameToModel(published_config(device="cpu"), projector=our.SphericalProjector(64, 2048),
                                    device=torch.device("cpu"))
    odo.init()
    init = our.ConstantVelocityInitialization()
    init.init()
    for f in range(6):
        pts, _ = O.grid_sample(scans[f], 0.4)
        assert pts.shape[0] == int(g["samples"][f])
        d = {"input_data": torch.from_numpy(pts)}
        init.next_frame(d)
        odo.process_next_frame(d)
        if f == 0:
            continue
        init.save_real_motion(d["odometry_pose"], d)
        dt, dr = O.pose_error(d["odometry_pose"], g["rel"][f])
        assert dt < 1e-4 and dr < 1e-4, (f, dt, dr)
        assert odo.last_result.iterations == int(g["iters"][f]), (f, odo.last_result.iterations, int(g["iters"][f]))
        # (the first cloud enters through the projection: a half-pixel rounding decided by the last bit of atan2f / asinf
        # — which differs between the reference's own CPU code paths, tests/test_gpu_parity.py::test_projection — may merge
        # or split one pair of its 7 000 points)
        assert abs(odo.ctx.lm.local_map.shape[0] - int(g["map_sizes"][f])) <= 2
