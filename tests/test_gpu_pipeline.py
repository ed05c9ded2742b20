"""GPU tests of the pieces AROUND the registration kernels (round 2): the device-resident preprocessing hand-off
(SURVEY §8f-1), the inner plugin seams on CUDA tensors, the point-to-point alignment mode of the odometry, the dataset
items under the reference runner's DataLoader defaults, and the failed-registration order of the asynchronous path.
Everything goes through the C ABI of libicp_mi355x.so; the oracle is the checker."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X (no CPU fallback exists for the product path)")
    return torch


@pytest.fixture(scope="module")
def O():
    import icp_oracle
    return icp_oracle


def _scans(h=32, w=512, n=4):
    from pylidar_slam_amd.synthetic import SceneConfig, make_sequence
    return make_sequence(SceneConfig(height=h, width=w), n)


def test_device_resident_preprocessing_handoff(torch_cuda):
    """`to_device -> distortion -> grid_sample -> to_tensor -> ICP` with every intermediate in HBM: the frame is uploaded
    once, nothing comes back to the host before the pose, the odometry consumes the filter's tensor itself (pointer
    identity) — and the trajectory equals, bit for bit, the one of the host (numpy) filters of the same library."""
    torch = torch_cuda
    from pylidar_slam_amd.odometry import (Distortion, DistortionConfig, GridSample, GridSampleConfig, MI355XICPConfig,
                                           MI355XICPFrameToModel, SphericalProjector, ToDevice, ToDeviceConfig, ToTensor,
                                           ToTensorConfig)
    h, w = 32, 512
    scans, _ = _scans(h, w, 4)
    rng = np.random.default_rng(5)
    stamps = [np.sort(rng.uniform(0.0, 0.1, s.shape[0])) for s in scans]

    def odometry():
        cfg = MI355XICPConfig(max_num_alignments=8, threshold_delta_pose=0.0, data_key="input_data")
        o = MI355XICPFrameToModel(cfg, projector=SphericalProjector(h, w), device=torch.device("cuda:0"))
        o.init()
        return o

    dev = torch.device("cuda:0")
    host_filters = [Distortion(DistortionConfig(output_key="distorted")),
                    GridSample(GridSampleConfig(voxel_size=0.3, pointcloud_key="distorted"))]
    dev_filters = [ToDevice(ToDeviceConfig(), device=dev),
                   Distortion(DistortionConfig(pointcloud_key="pc_device", timestamps_key="timestamps_device",
                                               output_key="distorted")),
                   GridSample(GridSampleConfig(voxel_size=0.3, pointcloud_key="distorted")),
                   ToTensor(ToTensorConfig(keys={"sample_points": "input_data"}, dtype="float32"), device=dev)]
    odo_h, odo_d = odometry(), odometry()
    last = None
    for f, (s, ts) in enumerate(zip(scans, stamps)):
        dh = {"numpy_pc": s, "numpy_pc_timestamps": ts, "init_rpose": last}
        dd = {"numpy_pc": s, "numpy_pc_timestamps": ts, "init_rpose": last}
        for flt in host_filters:
            flt.filter(dh)
        dh["input_data"] = torch.from_numpy(dh["sample_points"]).to(dev)  # the reference's ToTensor (float64 tensor)
        for flt in dev_filters:
            flt.filter(dd)
        for key in ("pc_device", "distorted", "sample_points", "sample_indices", "input_data"):
            assert isinstance(dd[key], torch.Tensor) and dd[key].is_cuda, key
        # without an initial pose the de-skew is a pass-through (same tensor); with one it yields float64 like the reference
        assert dd["input_data"].dtype == torch.float32
        assert dd["distorted"].dtype == (torch.float64 if last is not None else torch.float32)
        np.testing.assert_array_equal(dd["sample_indices"].cpu().numpy(), dh["sample_indices"])
        np.testing.assert_array_equal(dd["sample_points"].cpu().numpy(), dh["sample_points"])
        odo_h.process_next_frame(dh)
        odo_d.process_next_frame(dd)
        assert odo_d._tgt_pc.data_ptr() == dd["input_data"].data_ptr()  # zero-copy into the registration
        if f:
            assert np.array_equal(dd["odometry_pose"], dh["odometry_pose"])
            last = dd["odometry_pose"].astype(np.float64)
    assert odo_d.last_result.iterations == 8


def test_inner_seams_on_cuda_tensors(torch_cuda, O):
    """LocalMap / RigidAlignment seams hand back tensors on the device of their inputs, in the reference's shapes
    (local_map.py:389-394, alignment.py:91-127): the reference's loop multiplies the pose with its own device tensor and
    sums the residuals (icp_odometry.py:284-297)."""
    torch = torch_cuda
    from conftest import GOLDEN
    from pylidar_slam_amd.engine import IcpContext
    from pylidar_slam_amd.odometry import (HashGridLocalMap, HashGridLocalMapConfig, PointToPlaneAlignment,
                                           PointToPlaneAlignmentConfig, PointToPointAlignment, ProjectiveLocalMap,
                                           SphericalProjector)
    g = np.load(os.path.join(GOLDEN, "components.npz"))
    dev = torch.device("cuda:0")
    lm = HashGridLocalMap(HashGridLocalMapConfig(local_map_size=3), projector=SphericalProjector(32, 256))  # registry form
    lm.init()
    lm.update(torch.eye(4).unsqueeze(0), new_pc_data=g["nn_map"])
    q = torch.from_numpy(g["nn_queries"]).to(dev)
    res = lm.nearest_neighbor_search(q)
    for t in (res.neighbor_points, res.neighbor_normals, res.new_target_points):
        assert isinstance(t, torch.Tensor) and t.device == q.device and tuple(t.shape) == (1, q.shape[0], 3)
    np.testing.assert_array_equal(res.neighbor_points[0].cpu().numpy(), g["nn_points"])
    assert (np.abs((res.neighbor_normals[0].cpu().numpy() * g["nn_normals"]).sum(axis=1)) > 1 - 1e-5).all()
    assert tuple(lm.get_last_frame().shape) == tuple(g["nn_map"].shape)
    # a cpu tensor query gives cpu tensors (the reference's CPU configuration)
    res_cpu = lm.nearest_neighbor_search(torch.from_numpy(g["nn_queries"]))
    assert res_cpu.neighbor_points.device.type == "cpu"
    al = PointToPlaneAlignment(PointToPlaneAlignmentConfig(gauss_newton_config=dict(max_iters=1, scheme="geman_mcclure",
                                                                                     sigma=0.3)))
    pose, params, residuals = al.align(res.neighbor_points, res.new_target_points, res.neighbor_normals)
    assert pose.device == q.device and tuple(pose.shape) == (1, 4, 4) and tuple(params.shape) == (1, 6)
    assert residuals.device == q.device and tuple(residuals.shape) == (1, q.shape[0])
    np.testing.assert_allclose(params[0].cpu().numpy(), g["gn_geman_mcclure_dx"], atol=2e-5)
    np.testing.assert_allclose(float(residuals.sum()), float(g["gn_geman_mcclure_loss"]), rtol=1e-4)
    composed = pose @ torch.eye(4, device=dev).unsqueeze(0)  # what the reference's loop does with it
    assert composed.device == q.device
    # numpy inputs keep giving numpy; point to point returns the residual vector too
    p2, prm2, res2 = PointToPointAlignment(IcpContext()).align(g["nn_points"], g["nn_queries"])
    assert isinstance(p2, np.ndarray) and p2.shape == (1, 4, 4) and res2.shape == (1, g["nn_queries"].shape[0])
    # projective map: device rows out for device points in; the last frame is the newest stored vertex map
    gp = np.load(os.path.join(GOLDEN, "projective.npz"))
    h, w = (int(v) for v in gp["hw"])
    pm = ProjectiveLocalMap(IcpContext(height=h, width=w, local_map_size=2))
    pm.init()
    vm = torch.from_numpy(gp["vmaps"][0]).to(dev)
    pm.update(torch.eye(4).unsqueeze(0), new_vertex_map=vm.unsqueeze(0))
    pts = vm.permute(1, 2, 0).reshape(-1, 3)
    pts = pts[pts.abs().amax(dim=1) > 0].contiguous()
    r = pm.nearest_neighbor_search(pts)
    assert r.neighbor_points.is_cuda and r.neighbor_normals.is_cuda and r.new_target_points.is_cuda
    assert r.neighbor_points.shape == r.new_target_points.shape and r.neighbor_points.shape[1] > 0.9 * pts.shape[0]
    assert tuple(pm.get_last_frame().shape) == (h * w, 3)


def test_point_to_point_mode_in_the_odometry(torch_cuda, O):
    """`alignment.mode = point_to_point_gauss_newton` selects `GaussNewtonPointToPointAlignment` in the loop
    (icp_odometry.py:98): search -> one point-to-point Gauss-Newton step from x0 = 0 -> pose composition, all on the
    device.  The reference's own loop cannot run this mode (it hands the map normals to `align` as `initial_estimate`,
    oracle/make_golden_p2p.py) and, with that argument dropped, iterates chaotically (its float32 and float64 runs part
    by decimetres within two frames), so the end-to-end pin is the FIRST alignment of a frame against the reference's
    value; the whole loop is checked against the oracle over few iterations, before the divergence amplifies."""
    torch = torch_cuda
    from conftest import GOLDEN
    from pylidar_slam_amd.odometry import MI355XICPConfig, MI355XICPFrameToModel, SphericalProjector, grid_sample
    g = np.load(os.path.join(GOLDEN, "p2p_sequence.npz"))
    h, w = (int(v) for v in g["hw"])
    scans, _ = _scans(h, w, 3)
    for name in ("ls", "gm"):
        scheme, sigma, _ = (str(v) for v in g[f"{name}_cfg"])
        cfg = MI355XICPConfig(max_num_alignments=3, threshold_delta_pose=0.0, data_key="sample_points",
                              alignment=dict(mode="point_to_point_gauss_newton",
                                             gauss_newton_config=dict(max_iters=1, scheme=scheme, sigma=float(sigma))))
        odo = MI355XICPFrameToModel(cfg, projector=SphericalProjector(h, w), device=torch.device("cuda:0"))
        odo.init()
        orc = O.ICPFrameToModelOracle(O.ICPOracleConfig(max_num_alignments=3, threshold_delta_pose=0.0, scheme=scheme,
                                                        sigma=float(sigma), height=h, width=w,
                                                        alignment="point_to_point", accumulate=np.float64))
        last = None
        for f, s in enumerate(scans):
            pts, _ = grid_sample(s, float(g["voxel"]), odo.ctx)
            d = {"sample_points": pts, "init_rpose": last}
            odo.process_next_frame(d)
            opose = orc.process_next_frame(O.grid_sample(s, float(g["voxel"]))[0], last)
            if f == 0:
                continue
            assert odo.last_result.normals_computed == 0  # point to point needs no map normals
            np.testing.assert_allclose(odo.last_result.losses, orc.traces[-1].loss, rtol=1e-3)
            dt, dr = O.pose_error(d["odometry_pose"], opose)
            assert dt < 1e-4 and dr < 1e-4, (name, f, dt, dr)
            if f == 1:  # first alignment of the first registered frame: the reference's own value
                np.testing.assert_allclose(odo.last_result.losses[0], g[f"{name}_loss"][1][0], rtol=1e-4)
            last = opose.astype(np.float64)  # both follow the same chain of initial guesses


def test_dataset_items_under_the_reference_dataloader_defaults(torch_cuda):
    """The reference runner builds `DataLoader(..., pin_memory=True)` whenever the device is not the cpu
    (slam/odometry/odometry_runner.py:51,88,149): the loaders return CPU tensors by default (pinning a CUDA tensor
    raises); `device_items=True` keeps the vertex map on the device for runners configured with pin_memory=false."""
    torch = torch_cuda
    from torch.utils.data import DataLoader
    from pylidar_slam_amd.dataset import SyntheticDatasetConfig, SyntheticDatasetLoader
    loader = SyntheticDatasetLoader(SyntheticDatasetConfig(lidar_height=16, lidar_width=128, num_frames=3))
    (train, names), _, _, _ = loader.sequences()
    keep_numpy = lambda batch: {k: (v if isinstance(v, np.ndarray) else v.unsqueeze(0)) for k, v in batch[0].items()}
    items = list(DataLoader(train[0], collate_fn=keep_numpy, pin_memory=True, batch_size=1, num_workers=0))
    assert len(items) == 3 and not items[0]["vertex_map"].is_cuda and items[0]["vertex_map"].is_pinned()
    assert tuple(items[0]["vertex_map"].shape) == (1, 3, 16, 128) and items[0]["numpy_pc"].shape == (16 * 128, 3)
    on_device = SyntheticDatasetLoader(SyntheticDatasetConfig(lidar_height=16, lidar_width=128, num_frames=2,
                                                              device_items=True))
    (train_d, _), _, _, _ = on_device.sequences()
    assert train_d[0][1]["vertex_map"].is_cuda
    np.testing.assert_array_equal(train_d[0][1]["vertex_map"].cpu().numpy(), train[0][1]["vertex_map"].numpy())


def test_failed_registration_leaves_the_map_where_it_was(torch_cuda):
    """The reference raises `RuntimeError("Invalid Jacobian ...")` inside the alignment (optimization.py:334-336), i.e.
    before `__update_map` touches the local map (icp_odometry.py:192-199).  On the asynchronous path the map
    re-expression is enqueued behind the registration with the device-resident pose: when the registration stopped on
    that error the kept points must not move, and the error surfaces in register_end."""
    from pylidar_slam_amd.engine import IcpContext, InvalidJacobianError
    ctx = IcpContext(height=16, width=128, max_num_alignments=4, threshold_delta_pose=0.0)
    # a map on the plane z = 0 and targets right above it: every row has the same normal, J^T J is singular
    xs, ys = np.meshgrid(np.arange(40, dtype=np.float32) * 0.1, np.arange(40, dtype=np.float32) * 0.1)
    plane = np.stack([xs.ravel(), ys.ravel(), np.zeros(xs.size, np.float32)], axis=1)
    ctx.map_set(plane)
    before = ctx.map_points()
    targets = plane[::3] + np.array([0.013, 0.007, 0.05], np.float32)
    ctx.register_launch(targets, None)
    ctx.map_update(None, None)
    with pytest.raises(InvalidJacobianError):
        ctx.register_end()
    np.testing.assert_array_equal(ctx.map_points(), before)
    # and the context keeps working afterwards
    good = IcpContext(height=16, width=128, max_num_alignments=4, threshold_delta_pose=0.0)
    scans, _ = _scans(16, 128, 2)
    good.map_set(scans[0])
    assert good.register(scans[1]).iterations == 4


def test_two_frames_in_flight_equal_the_synchronous_loop(torch_cuda):
    """`icp_register_launch_from_last`: the constant-velocity guess (= the previous result) is read on the device, so
    frame t + 1 and the map update by the pose of frame t are enqueued before the pose of frame t is collected.  Same
    poses, losses and final map, bit for bit, as the loop that takes every pose through the host; results come back
    oldest first; a third launch without collecting is refused."""
    torch = torch_cuda
    from pylidar_slam_amd.engine import IcpContext
    from pylidar_slam_amd.synthetic import SceneConfig, make_fixed_map, make_sequence
    cfg = SceneConfig(height=32, width=1024)
    scans, poses = make_sequence(cfg, 9)
    model = make_fixed_map(cfg, scans[:4], poses[:4], ref_frame=3, num_points=30_000)
    kw = dict(height=32, width=1024, max_num_alignments=8, threshold_delta_pose=0.0, scheme="geman_mcclure", sigma=0.3)
    dscans = [torch.from_numpy(s).cuda() for s in scans]
    sync, pipe = IcpContext(**kw), IcpContext(**kw)
    for c in (sync, pipe):
        c.map_set(model)
    # synchronous reference loop: every pose goes through the host
    want, last = [], None
    for f in range(4, 9):
        sync.register_launch(dscans[f], last)
        sync.map_update(None, None)
        r = sync.register_end()
        want.append(r)
        last = r.pose
    # pipelined: first frame as above, then two in flight
    got = []
    pipe.register_launch(dscans[4], None)
    pipe.map_update(None, None)
    got.append(pipe.register_end())
    pending = 0
    for f in range(5, 9):
        pipe.register_launch(dscans[f], "last")
        pipe.map_update(None, None)
        pending += 1
        if pending == 2:
            got.append(pipe.register_end())
            pending -= 1
    pipe.register_launch(dscans[4], "last")  # two pending now
    with pytest.raises(AssertionError):
        pipe.register_launch(dscans[5], "last")
    while len(got) < 5:
        got.append(pipe.register_end())
    extra = pipe.register_end()
    assert extra.iterations == 8
    with pytest.raises(AssertionError):
        pipe.register_end()  # nothing pending any more
    for a, b in zip(got, want):
        assert np.array_equal(a.pose, b.pose) and np.array_equal(a.losses, b.losses) and np.array_equal(a.dx, b.dx)
    fresh = IcpContext(**kw)
    fresh.map_set(model)
    with pytest.raises(AssertionError):
        fresh.register_launch(dscans[4], "last")  # no previous registration to start from


# ---- round-3 regressions (ADVICE.md of round 2) ----------------------------------------------------------------------
def test_projection_after_a_projective_registration_starts_from_a_clean_z_buffer(torch_cuda):
    """The projective iteration leaves its target keys in the context's z-buffer; the next `icp_project` on that context
    must clear them first (the "left clean by the resolve kernel" shortcut only holds behind another projection).
    project -> pmap_register (threshold 0: the loop ends on its last iteration) -> project of a SMALLER cloud, compared
    with a fresh context bit for bit — on a stale buffer old (range, ~index) keys win pixels and index past the cloud."""
    from pylidar_slam_amd.engine import IcpContext
    h, w = 32, 512
    scans, _ = _scans(h, w, 3)
    kw = dict(height=h, width=w, max_num_alignments=5, threshold_delta_pose=0.0, local_map_size=3)
    ctx, fresh = IcpContext(**kw), IcpContext(**kw)
    d0 = torch_cuda.from_numpy(scans[0]).cuda()
    vm0 = ctx.project(d0)
    ctx.pmap_init()
    ctx.pmap_update(np.eye(4, dtype=np.float32), vm0)
    ctx.pmap_register(torch_cuda.from_numpy(scans[1]).cuda(), None)
    small = scans[2][::3].copy()  # fewer points than the frame whose keys stayed behind, and farther ranges win nothing
    small *= 1.5
    got, gi = ctx.project(small, with_index=True)
    want, wi = fresh.project(small, with_index=True)
    assert np.array_equal(got, want) and np.array_equal(gi, wi)
    # the same through the plugin: projective local map fed with [N, 3] clouds projects on the registration's context
    from pylidar_slam_amd.odometry import MI355XICPConfig, MI355XICPFrameToModel, SphericalProjector
    cfg = MI355XICPConfig(max_num_alignments=5, threshold_delta_pose=0.0, data_key="numpy_pc",
                          local_map=dict(type="projective_local_map", local_map_size=3))
    odo = MI355XICPFrameToModel(cfg, projector=SphericalProjector(h, w), device=torch_cuda.device("cuda:0"))
    odo.init()
    for s in scans:
        odo.process_next_frame({"numpy_pc": s})
        assert np.array_equal(odo._tgt_vmap.cpu().numpy(), fresh.project(s))


def test_insertion_by_the_device_pose_waits_for_the_pending_result(torch_cuda):
    """`icp_map_update(rel_pose = NULL, new cloud)` behind a registration whose status the host has not seen would evict /
    insert even if that registration failed (the reference raises before it touches the map): refused while the result is
    pending, accepted once it has been collected; the pose-only form stays available."""
    from pylidar_slam_amd.engine import IcpContext
    scans, _ = _scans(16, 256, 3)
    ctx = IcpContext(height=16, width=256, max_num_alignments=4, threshold_delta_pose=0.0, local_map_size=2)
    ctx.map_init()
    ctx.map_update(np.eye(4, dtype=np.float32), scans[0])
    ctx.register_launch(scans[1], None)
    with pytest.raises(AssertionError):
        ctx.map_update(None, scans[1])
    assert ctx.map_num_clouds() == 1 and ctx.map_size() == scans[0].shape[0]
    res = ctx.register_end()
    ctx.map_update(None, scans[1])  # collected: the device pose is the pose the host holds
    assert ctx.map_num_clouds() == 2
    ref = IcpContext(height=16, width=256, max_num_alignments=4, threshold_delta_pose=0.0, local_map_size=2)
    ref.map_init()
    ref.map_update(np.eye(4, dtype=np.float32), scans[0])
    ref.map_update(res.pose, scans[1])
    np.testing.assert_array_equal(ctx.map_points(), ref.map_points())


def test_compact_targets_equals_boolean_indexing(torch_cuda):
    """`icp_compact_targets`: the passing rows in order at the head, zeros behind — and a registration of the compacted
    rows (skip_null) gives the bits of the registration of all rows."""
    torch = torch_cuda
    from pylidar_slam_amd.engine import IcpContext
    scans, _ = _scans(32, 512, 2)
    ctx = IcpContext(height=32, width=512, max_num_alignments=6, threshold_delta_pose=0.0)
    rows = scans[1].copy()
    rows[::3] = 0.0
    rows[5] = np.nan
    keep = ~np.isnan(rows).any(axis=1) & (np.abs(rows).max(axis=1) > 0)
    cap = int(keep.sum()) + 37
    got = ctx.compact_targets(torch.from_numpy(rows).cuda(), cap).cpu().numpy()
    assert got.shape == (cap, 3)
    np.testing.assert_array_equal(got[:keep.sum()], rows[keep])
    assert not got[keep.sum():].any()
    tight = ctx.compact_targets(torch.from_numpy(rows).cuda(), int(keep.sum()) - 5).cpu().numpy()  # too small: dropped
    np.testing.assert_array_equal(tight, rows[keep][:keep.sum() - 5])
    ctx.map_set(scans[0])
    a = ctx.register(torch.from_numpy(rows).cuda(), None, skip_null=True)
    b = ctx.register(torch.from_numpy(got).cuda(), None, skip_null=True)
    assert a.num_targets == b.num_targets == int(keep.sum())
    np.testing.assert_allclose(a.pose, b.pose, atol=1e-6)  # (another grouping of the rows into partial sums)


@pytest.mark.parametrize("threshold", [1.0e-4, 1.0e-3])
def test_chunked_launch_equals_the_full_launch(torch_cuda, threshold):
    """A launched registration with a live stop threshold is enqueued in chunks (first chunk = the iterations of the last
    frame + 1; `icp_register_end` adds chunks while the loop runs).  Same poses, losses and iteration counts, bit for bit,
    as with every iteration enqueued up front — over frames whose iteration counts go up and down — and the pose-only
    map update by the device pose still follows the END of the registration."""
    torch = torch_cuda
    from pylidar_slam_amd.engine import IcpContext
    from pylidar_slam_amd.synthetic import SceneConfig, make_fixed_map, make_sequence
    cfg = SceneConfig(height=32, width=1024)
    scans, poses = make_sequence(cfg, 10)
    model = make_fixed_map(cfg, scans[:4], poses[:4], ref_frame=3, num_points=30_000)
    kw = dict(height=32, width=1024, max_num_alignments=20, threshold_delta_pose=threshold, scheme="geman_mcclure",
              sigma=0.3)
    dscans = [torch.from_numpy(s).cuda() for s in scans]
    runs = {}
    for chunked in (1, 0):
        ctx = IcpContext(**kw)
        ctx.set_option("chunked_launch", chunked)
        ctx.map_set(model)
        out, last = [], None
        for f in (4, 5, 9, 6, 7):  # (9: a jump, more iterations than the frame before; 6: fewer again)
            ctx.register_launch(dscans[f], last if f != 9 else None)
            ctx.map_update(None, None)
            r = ctx.register_end()
            out.append(r)
            last = r.pose
        runs[chunked] = (out, ctx.map_points())
    iters = [r.iterations for r in runs[1][0]]
    assert min(iters) < max(iters) < 20, iters
    for a, b in zip(*[runs[k][0] for k in (1, 0)]):
        assert a.iterations == b.iterations and a.converged == b.converged
        assert np.array_equal(a.pose, b.pose) and np.array_equal(a.losses, b.losses) and np.array_equal(a.dx, b.dx)
    np.testing.assert_array_equal(runs[1][1], runs[0][1])


def test_calls_between_a_chunked_launch_and_its_end_follow_the_whole_registration(torch_cuda):
    """with a live stop threshold only the first chunk of iterations is on the stream when
    `icp_register_launch` returns.  A map update with an EXPLICIT pose, an option change or a new alignment called before
    `icp_register_end` must not reach the held-back iterations: the library enqueues them first, so the result equals the
    un-chunked launch bit for bit (stream order = call order)."""
    torch = torch_cuda
    from pylidar_slam_amd.engine import IcpContext
    from pylidar_slam_amd.synthetic import SceneConfig, make_fixed_map, make_sequence
    cfg = SceneConfig(height=32, width=1024)
    scans, poses = make_sequence(cfg, 8)
    model = make_fixed_map(cfg, scans[:4], poses[:4], ref_frame=3, num_points=30_000)
    kw = dict(height=32, width=1024, max_num_alignments=20, threshold_delta_pose=1e-4, scheme="geman_mcclure", sigma=0.3)
    dscans = [torch.from_numpy(s).cuda() for s in scans]
    shift = np.eye(4, dtype=np.float32)
    shift[:3, 3] = (0.3, -0.2, 0.05)
    runs = {}
    for chunked in (1, 0):
        ctx = IcpContext(**kw)
        ctx.set_option("chunked_launch", chunked)
        ctx.map_set(model)
        out = []
        # frame 4 converges in a few iterations: the first chunk of frame 7's launch (a jump of three frames, identity
        # guess) is far shorter than the iterations it needs
        out.append(ctx.register(dscans[4]))
        ctx.register_launch(dscans[7], None)
        ctx.map_update(shift, None)            # explicit pose: rebuilds the grid the held-back iterations search
        ctx.set_option("prune_guard", 0.004)   # an option of the search
        out.append(ctx.register_end())
        ctx.register_launch(dscans[5], None)
        ctx.set_alignment("huber", 0.5, 20, 1e-4)  # must apply to the NEXT registration only
        out.append(ctx.register_end())
        out.append(ctx.register(dscans[6]))
        runs[chunked] = (out, ctx.map_points())
        ctx.close()
    assert runs[1][0][1].iterations > runs[1][0][0].iterations + 1, [r.iterations for r in runs[1][0]]
    for a, b in zip(runs[1][0], runs[0][0]):
        assert a.iterations == b.iterations and a.converged == b.converged
        assert np.array_equal(a.pose, b.pose) and np.array_equal(a.losses, b.losses) and np.array_equal(a.dx, b.dx)
    np.testing.assert_array_equal(runs[1][1], runs[0][1])



@pytest.mark.gpu
def test_staged_map_update_equals_the_direct_update(torch_cuda):
    """`icp_map_stage_cloud` before a registration + `icp_map_update_staged` after it = `icp_map_update` with the cloud
    (update() of local_map.py:302-362: rows with a NaN dropped, null rows too under skip_null, order kept, the oldest
    cloud evicted beyond the window) — same counts, same map, same registration afterwards, on device and host input; a
    staged cloud is consumed once; staging replaces what was staged."""
    torch = torch_cuda
    from pylidar_slam_amd.engine import IcpContext
    from pylidar_slam_amd.synthetic import SceneConfig, make_sequence
    cfg = SceneConfig(height=32, width=1024)
    scans, _ = make_sequence(cfg, 6)
    rng = np.random.default_rng(5)
    clouds = []
    for s in scans:
        c = s.copy()
        c[rng.integers(0, len(c), 200)] = np.nan      # rows with a NaN
        c[rng.integers(0, len(c), 300)] = 0.0         # null rows
        clouds.append(c)
    shift = np.eye(4, dtype=np.float32)
    shift[:3, 3] = (0.2, -0.1, 0.02)
    kw = dict(height=32, width=1024, max_num_alignments=8, threshold_delta_pose=0.0, scheme="geman_mcclure", sigma=0.3,
              local_map_size=3)
    runs = {}
    for staged in (False, True):
        for on_device in (True, False):
            ctx = IcpContext(**kw)
            ctx.map_init()
            counts, results = [], []
            for i, c in enumerate(clouds):
                rows = torch.from_numpy(c).cuda() if on_device else c
                skip_null = i % 2 == 0
                if staged:
                    ctx.map_stage_cloud(rows, skip_null=skip_null)
                    if i == 2:  # staging again replaces the staged cloud
                        ctx.map_stage_cloud(rows, skip_null=skip_null)
                if i > 0:
                    results.append(ctx.register(torch.from_numpy(scans[i]).cuda()))
                counts.append(ctx.map_update_staged(shift) if staged else ctx.map_update(shift, rows, skip_null=skip_null))
            if staged:
                with pytest.raises(AssertionError):  # (ICP_ERR_INVALID_ARGUMENT)
                    ctx.map_update_staged(shift)  # consumed
            runs[(staged, on_device)] = (counts, ctx.map_points(), ctx.map_num_clouds(), results)
            ctx.close()
    ref = runs[(False, True)]
    assert ref[2] == 3 and len(ref[1]) == sum(ref[0][-3:])
    for key, (counts, pts, nclouds, results) in runs.items():
        assert counts == ref[0], key
        assert nclouds == ref[2], key
        np.testing.assert_array_equal(pts, ref[1], err_msg=str(key))
        for a, b in zip(results, ref[3]):
            np.testing.assert_array_equal(a.pose, b.pose, err_msg=str(key))
            np.testing.assert_array_equal(a.losses, b.losses, err_msg=str(key))


def test_masked_sparse_vertex_map_equals_the_compacted_one(torch_cuda):
    """`sample_points` of the reference (icp_odometry.py:301-308) keeps the non-null pixels of the new vertex map.  The
    plugin either compacts them on the device (`compact_sparse_vertex_map: true`, round 4) or hands all H x W pixels to
    the registration, which masks the null ones in its kernels (default since round 5).  Same targets in the same pixel
    order: the same iteration counts, and poses equal up to the summation order of the float32 normal equations (the
    rows sit in other workgroups)."""
    torch = torch_cuda
    from pylidar_slam_amd import odometry as our
    scans, _ = _scans(32, 1024, 5)
    runs = {}
    for compact in (True, False):
        cfg = our.MI355XICPConfig(compact_sparse_vertex_map=compact, max_num_alignments=10, data_key="input_data",
                                  threshold_delta_pose=1e-4)  # (tensor rows in: the targets are the vertex map's pixels)
        odo = our.MI355XICPFrameToModel(cfg, projector=our.SphericalProjector(32, 1024), device=torch.device("cuda:0"))
        odo.init()
        poses, iters = [], []
        for s in scans:
            # a grid-sampled frame: a few thousand of the 32768 pixels are set
            pts, _ = our.grid_sample(s, 0.4)
            d = {"input_data": torch.from_numpy(np.ascontiguousarray(pts, np.float32)).cuda()}
            odo.process_next_frame(d)
            if "odometry_pose" in d:
                poses.append(np.asarray(d["odometry_pose"]))
                iters.append(int(odo.last_result.iterations))
        runs[compact] = (np.stack(poses), iters)
        assert odo.ctx.handoff_fallbacks() == 0
    assert runs[True][1] == runs[False][1]
    np.testing.assert_allclose(runs[True][0], runs[False][0], atol=2e-6, rtol=0)


def test_project_rows_equals_the_transposed_vertex_map(torch_cuda):
    """`icp_project_rows`: the vertex map of `icp_project` and, from the same launch, its pixels as [H*W, 3] rows
    (`vmap.permute(1, 2, 0).reshape(-1, 3)`, what sample_points of icp_odometry.py:301-308 indexes) — bit for bit, NaN rows
    of a padded frame skipped, a second call on the same context clean."""
    torch = torch_cuda
    from pylidar_slam_amd.engine import IcpContext
    scans, _ = _scans(32, 512, 2)
    ctx = IcpContext(height=32, width=512)
    for s in scans:
        rows_in = s.copy()
        rows_in[::7] = np.nan
        pc = torch.from_numpy(rows_in).cuda()
        vmap, rows = ctx.project_rows(pc)
        ref = ctx.project(pc)
        assert torch.equal(vmap, ref)
        assert torch.equal(rows, ref.permute(1, 2, 0).reshape(-1, 3))
        assert rows.shape == (32 * 512, 3) and int((rows.abs().sum(dim=1) > 0).sum()) > 1000
    ctx.close()
