"""`-m gpu`: the reference's published configuration as a full odometry loop on the MI355X (SURVEY §8 row f3 / BASELINE
configs[2] stand-in), against `tests/golden/loop_reference.npz` — the trajectory of the reference's own `SLAM` loop on
the same 36 seeded 64x2048 frames (oracle/make_golden_loop.py).  Every frame within 1e-4 m / 1e-4 rad of the reference's,
the same ATE / ARE / segment errors through `pylidar_slam_amd.eval`, the sliding window (insertions, evictions) of the
same sizes — with the reference's host preprocessing order (numpy grid sample -> tensor) and with the device-resident
filters of config/slam/preprocessing/grid_sample_mi355x.yaml."""
import numpy as np
import pytest

from test_loop_reference import golden_loop, loop_scans, published_config, trajectory_metrics  # noqa: F401 (fixtures)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X (no CPU fallback exists for the product path)")
    return torch


def _filters(kind, dev):
    from pylidar_slam_amd import odometry as our
    if kind == "host":  # config/slam/preprocessing/grid_sample.yaml of the reference, with our GridSample
        return [our.Distortion(our.DistortionConfig(output_key="distorted")),
                our.GridSample(our.GridSampleConfig(voxel_size=0.4, pointcloud_key="distorted")),
                our.ToTensor(our.ToTensorConfig(device=str(dev), keys={"sample_points": "input_data"}), device=dev)]
    # "device_padded": the grid sample that reads nothing back (sample rows padded with NaN rows to the frame's size, the
    # count left on the device): one host synchronisation per frame
    return [our.ToDevice(our.ToDeviceConfig(device=str(dev)), device=dev),
            our.Distortion(our.DistortionConfig(pointcloud_key="pc_device", timestamps_key="timestamps_device",
                                                output_key="distorted")),
            our.GridSample(our.GridSampleConfig(voxel_size=0.4, pointcloud_key="distorted", padded=kind == "device_padded")),
            our.ToTensor(our.ToTensorConfig(device=str(dev), keys={"sample_points": "input_data"}, dtype="float32"),
                         device=dev)]


def _sample_count(d) -> int:
    """Rows the grid sample selected: the row count of `sample_points`, or — padded variant — the device-side count."""
    return int(d["sample_count"]) if "sample_count" in d else int(d["sample_points"].shape[0])


def _drive(torch, scans, config, preprocessing):
    from pylidar_slam_amd import odometry as our
    dev = torch.device("cuda:0")
    odo = our.MI355XICPFrameToModel(config, projector=our.SphericalProjector(64, 2048), device=dev)
    filters = _filters(preprocessing, dev)
    init = our.ConstantVelocityInitialization()
    odo.init()
    init.init()
    for f, scan in enumerate(scans):
        d = {"numpy_pc": scan}
        init.next_frame(d)  # slam/slam.py:126-127
        for flt in filters:
            flt.filter(d)   # :129-130
        odo.process_next_frame(d)
        if f > 0:
            init.save_real_motion(d["odometry_pose"], d)  # :139-140
        yield f, d, odo


@pytest.mark.parametrize("preprocessing", ["host", "device", "device_padded"])
def test_forced_iteration_loop_matches_the_reference_run_frame_by_frame(torch_cuda, golden_loop, loop_scans, preprocessing):
    """The loop with the stop test off (threshold 0, 6 iterations per frame; `forced_*` of the fixture): grid sample ->
    registration -> sliding-window map with an insertion per frame and evictions from frame 30 on.  EVERY frame within
    1e-4 m / 1e-4 rad of the reference's run, the same ATE / ARE / segment translation error through
    `pylidar_slam_amd.eval`, window sizes equal up to the coin tosses of the first cloud's projection."""
    import icp_oracle as O
    g = golden_loop
    scans, gt_abs = loop_scans
    cfg = published_config(max_num_alignments=int(g["forced_iters_per_frame"]), threshold_delta_pose=0.0)
    worst, odo = (0.0, 0.0), None
    for f, d, odo in _drive(torch_cuda, scans, cfg, preprocessing):
        assert _sample_count(d) == int(g["forced_samples"][f])  # grid sampling is index-exact
        if f == 0:
            assert "odometry_pose" not in d
            continue
        dt, dr = O.pose_error(d["odometry_pose"], g["forced_rel"][f])
        worst = (max(worst[0], dt), max(worst[1], dr))
        assert dt < 1e-4 and dr < 1e-4, (preprocessing, f, dt, dr)
        assert odo.last_result.iterations == int(g["forced_iters"][f])
        assert abs(odo.ctx.map_size() - int(g["forced_map_sizes"][f])) <= 2, (f, odo.ctx.map_size())
        assert d["odometry_pc"] is d["distorted"]  # icp_odometry.py:210-211: the de-skewed frame when there is one
    assert odo.ctx.map_num_clouds() == 30  # six evictions happened
    ate, are, tr, rot, n = trajectory_metrics(odo.get_relative_poses(), gt_abs, g["segments"])
    print(f"forced loop ({preprocessing} preprocessing): worst frame {worst[0]:.1e} m / {worst[1]:.1e} rad vs the reference; "
          f"ATE {ate:.4e} (reference {g['forced_ate'][0]:.4e}) m, tr_err {tr:.4e} ({g['forced_kitti'][0]:.4e}) m/m")
    assert n == int(g["forced_num_segments"])
    assert abs(ate - g["forced_ate"][0]) < 1e-5 and abs(are - g["forced_are"][0]) < 1e-5
    assert abs(tr - g["forced_kitti"][0]) < 1e-5


@pytest.mark.parametrize("preprocessing", ["host", "device", "device_padded"])
def test_published_configuration_loop_matches_the_reference_run(torch_cuda, golden_loop, loop_scans, preprocessing):
    """The published configuration itself (live stop at |dx| < 1e-4, at most 20 iterations).  Every frame whose loop ran the
    reference's number of iterations: within 1e-4 m / 1e-4 rad of the reference's run.  A frame may stop after another
    number of iterations only where the REFERENCE's own stop was a close call, and by what it then differs is bounded by
    what the reference measured there (tests/golden/loop_spread.npz, oracle/make_golden_loop_spread.py: |delta_pose| of
    every iteration of the reference's run, and the same under three perturbations of its float evaluation): the deciding
    |delta_pose| within 2 % of the threshold (frame 24: 1.0064e-4 — the reference's scalar and vector ATen paths move that
    very number by 0.18 %), and the frame off by at most that step — the one evaluation applies it, the other does not
    (icp_odometry.py:292-297) — plus the 1e-4 of a frame without a flip.  (Round 4 allowed a flat 2e-4 behind any flip:
    argued, not measured.)  The trajectory metrics (ATE / ARE / segment translation error) equal to 2e-5."""
    import os
    import icp_oracle as O
    from conftest import GOLDEN
    g = golden_loop
    spread = np.load(os.path.join(GOLDEN, "loop_spread.npz"))
    assert bool(spread["base_reproduces_loop_reference"])  # the fixture describes the run loop_reference.npz holds
    scans, gt_abs = loop_scans
    worst, flips, odo = (0.0, 0.0), [], None
    for f, d, odo in _drive(torch_cuda, scans, published_config(), preprocessing):
        assert _sample_count(d) == int(g["samples"][f])
        if f == 0:
            continue
        dt, dr = O.pose_error(d["odometry_pose"], g["rel"][f])
        worst = (max(worst[0], dt), max(worst[1], dr))
        ours, theirs = int(odo.last_result.iterations), int(g["iters"][f])
        bound = 1e-4
        if ours != theirs:
            margin = float(spread["stop_margin"][f])
            step = float(spread["base_dx_norm"][f, min(ours, theirs) - 1])  # the step one loop applied and the other did not
            flips.append((f, ours, theirs, margin, step, dt))
            assert abs(ours - theirs) == 1 and margin < 0.02, (preprocessing, f, ours, theirs, margin)
            bound = 1e-4 + step
        assert dt < bound and dr < 1e-4, (preprocessing, f, dt, dr, ours, theirs)
        assert abs(odo.ctx.map_size() - int(g["map_sizes"][f])) <= 2, (f, odo.ctx.map_size(), int(g["map_sizes"][f]))
    assert odo.ctx.map_num_clouds() == 30
    assert odo.ctx.handoff_fallbacks() == 0  # (no hand-off of the lead launches may have timed out: it would be repaired silently)
    rel = odo.get_relative_poses()
    ate, are, tr, rot, n = trajectory_metrics(rel, gt_abs, g["segments"])
    assert n == int(g["num_segments"])
    print(f"loop ({preprocessing} preprocessing): worst frame {worst[0]:.1e} m / {worst[1]:.1e} rad vs the reference; ATE "
          f"{ate:.4e} (reference {g['ate'][0]:.4e}) m, tr_err {tr:.4e} ({g['kitti'][0]:.4e}) m/m, r_err {rot:.4e} "
          f"({g['kitti'][1]:.4e}) rad/m; frames with another iteration count (frame, ours, reference's, the reference's "
          f"stop margin, the step, |dt|): {flips}")
    assert abs(ate - g["ate"][0]) < 2e-5 and abs(are - g["are"][0]) < 2e-5
    assert abs(tr - g["kitti"][0]) < 2e-5
    # the segment ROTATION error is arccos((trace - 1) / 2) of float32 pose products: at 1e-4 rad it sits below the
    # sqrt(float32 epsilon) = 3e-4 rad noise floor of that formula (the reference's own figure is noise too), so it is
    # bounded, not compared; ARE above (linear in the error) is the rotation figure that is compared
    assert rot < 1e-3 and g["kitti"][1] < 1e-3
    assert len(flips) <= 3
