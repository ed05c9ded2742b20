"""GPU parity tests: the HIP path (through the C ABI of libicp_mi355x.so) against the CPU oracle on the same seeded
inputs and against the golden vectors produced by the reference's own code (tests/golden, oracle/make_golden.py).

Tolerances (BASELINE.json north_star): poses within 1e-4 m / 1e-4 rad per frame; index / integer work bit-exact.
"""
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


def _ctx(**kw):
    from pylidar_slam_amd.engine import IcpContext
    return IcpContext(**kw)


# ----------------------------------------------------------------------------------------------------------------------
def test_library_loaded_is_in_tree(torch_cuda):
    from pylidar_slam_amd import _lib
    lib = _lib.load_library()
    assert lib.icp_version().decode().startswith("icp_mi355x")
    with open("/proc/self/maps") as f:
        assert any("pylidar_slam_amd/_lib/libicp_mi355x.so" in line for line in f)


def test_projection(torch_cuda, O, golden_components):
    """Pixel coordinates and the vertex map against the reference's own (tests/golden/components.npz).

    The float coordinates agree to a few ulp; the INTEGER pixel of a point is the rounding of that float, so a point whose
    coordinate sits within those few ulp of x.5 is a coin toss — in the reference itself: its own CPU code paths
    (AVX2 / AVX512 Sleef vs scalar libm) disagree on 2 % of the coordinates by up to 2.3e-5 pixel
    (tests/golden/projection_spread.npz, oracle/make_golden_projection_spread.py).  The bound proven here: every point
    the HIP kernel assigns to another pixel than the reference did has its REFERENCE coordinate within the band (4 ulp, or
    1.5 x the reference's own spread on this cloud) of a half-integer, and the vertex map differs from the reference's only at pixels such a point leaves or enters."""
    import os
    from conftest import GOLDEN
    g = golden_components
    h, w = (int(v) for v in g["proj_hw"])
    up, down = (float(v) for v in g["proj_fov"])
    ctx = _ctx(height=h, width=w, up_fov=up, down_fov=down)
    rows, cols = ctx.project_pixels(g["proj_pc"])
    ref_r, ref_c = g["proj_pixels"][:, 0], g["proj_pixels"][:, 1]
    spread = np.load(os.path.join(GOLDEN, "projection_spread.npz"))
    own = max(float(spread[k]) for k in spread.files if k.startswith("small_") and k.endswith("max_pixel_difference"))
    # the band: 4 ulp of the largest coordinate, and no tighter than 1.5 x what the reference's own code paths differ by
    # on this very cloud (2.3e-5: the cancellation in 1 - (phi + fov_down) / fov carries asin's last bit into the row)
    band_r = band_c = max(4 * float(np.spacing(np.float32(max(h, w)))), 1.5 * own)
    assert own > 0.0
    np.testing.assert_allclose(rows, ref_r, atol=band_r)
    np.testing.assert_allclose(cols, ref_c, atol=band_c)
    print(f"projection: max |d row| {np.abs(rows - ref_r).max():.2e} (band {band_r:.1e}), max |d col| "
          f"{np.abs(cols - ref_c).max():.2e} (band {band_c:.1e}); the reference differs from itself by up to {own:.2e}")
    vmap, idx = ctx.project(g["proj_pc"], with_index=True)
    # points that land in another pixel than in the reference's run: each one a coin toss by the criterion above
    moved = (np.rint(rows) != np.rint(ref_r)) | (np.rint(cols) != np.rint(ref_c))
    toss = (np.abs(ref_r - np.floor(ref_r) - 0.5) < band_r) | (np.abs(ref_c - np.floor(ref_c) - 0.5) < band_c)
    assert not (moved & ~toss).any(), np.nonzero(moved & ~toss)[0]
    assert moved.sum() <= 4
    touched = np.zeros((h, w), bool)
    for i in np.nonzero(moved)[0]:
        for r_, c_ in ((np.rint(rows[i]), np.rint(cols[i])), (np.rint(ref_r[i]), np.rint(ref_c[i]))):
            if 0 <= r_ < h and 0 <= c_ < w:
                touched[int(r_), int(c_)] = True
    differs = np.abs(vmap - g["proj_vmap"]).max(axis=0) > 0
    assert not (differs & ~touched).any(), np.argwhere(differs & ~touched)
    ovmap, oidx = O.build_projection_map(g["proj_pc"], h, w, up, down, return_index=True)
    assert (idx != oidx).sum() <= 2 * max(1, int(moved.sum())) + 2  # (the oracle rounds numpy's libm values: its own tosses)
    # device-resident variant gives the same bits
    t = torch_cuda.from_numpy(g["proj_pc"]).cuda()
    dv = ctx.project(t)
    assert dv.is_cuda and np.array_equal(dv.cpu().numpy(), vmap)


def test_projection_edge_cases(torch_cuda, O):
    ctx = _ctx(height=8, width=16)
    # empty cloud -> all-zero map
    v, i = ctx.project(np.zeros((0, 3), np.float32), with_index=True)
    assert v.shape == (3, 8, 16) and not v.any() and (i == -1).all()
    # zeros, NaN rows and duplicates: never selected / highest index wins the tie
    pc = np.array([[0, 0, 0], [np.nan, 1, 1], [5, 0.1, -0.5], [5, 0.1, -0.5], [10, 0.2, -1.0]], np.float32)
    v, i = ctx.project(pc, with_index=True)
    ov, oi = O.build_projection_map(pc, 8, 16, 3.0, -24.0, return_index=True)
    np.testing.assert_array_equal(i, oi)
    assert 3 in i and 2 not in i and 0 not in i and 1 not in i
    np.testing.assert_array_equal(v, ov)


def test_voxel_hash_and_grid_sample(torch_cuda, O, golden_components):
    g = golden_components
    ctx = _ctx()
    vox, hashes = ctx.voxel_hash(g["gs_pc"], float(g["gs_voxel"]))
    np.testing.assert_array_equal(vox, g["gs_voxels"])  # bit exact vs the reference's numba body
    np.testing.assert_array_equal(hashes, g["gs_hashes"])
    pts, idx = ctx.grid_sample(g["gs_pc"], float(g["gs_voxel"]))
    np.testing.assert_array_equal(idx, g["gs_indices"])
    np.testing.assert_array_equal(pts, g["gs_pc"][g["gs_indices"]])
    # device-resident
    dp, di = ctx.grid_sample(torch_cuda.from_numpy(g["gs_pc"]).cuda(), float(g["gs_voxel"]))
    np.testing.assert_array_equal(di.cpu().numpy(), g["gs_indices"])
    # ragged / degenerate inputs
    p0, i0 = ctx.grid_sample(np.zeros((0, 3), np.float32), 0.3)
    assert p0.shape == (0, 3) and i0.shape == (0,)
    same = np.tile(np.array([[1.0, 2.0, 3.0]], np.float32), (100, 1))
    p1, i1 = ctx.grid_sample(same, 0.3)
    np.testing.assert_array_equal(i1, [0])
    # large coordinates: int64 wrap-around identical to numpy
    rng = np.random.default_rng(5)
    big = (rng.normal(size=(5000, 3)) * 1e6).astype(np.float32)
    _, hb = ctx.voxel_hash(big, 1e-3)
    np.testing.assert_array_equal(hb, O.voxel_hashing(O.voxelise(big, 1e-3)))
    _, ib = ctx.grid_sample(big, 1e-3)
    np.testing.assert_array_equal(ib, O.grid_sample(big, 1e-3)[1])


def test_padded_grid_sample_equals_the_exact_one(torch_cuda, O, golden_components):
    """`icp_grid_sample_padded[_f64]` (the device-resident pipeline's variant: no synchronisation, the count stays on the
    device): the first V rows / indices are the exact entry point's — which are the reference's (test above) — the rows
    behind them NaN / -1; float32 and float64 inputs, a full 64x2048 scan, a two-point cloud, an empty one, and a cloud
    whose voxels outnumber one workgroup's tiles several times over (the one-workgroup radix sort walks them chunk by
    chunk)."""
    from pylidar_slam_amd.synthetic import SceneConfig, make_sequence
    ctx = _ctx()
    scan = make_sequence(SceneConfig(height=64, width=2048), 1)[0][0]
    rng = np.random.default_rng(5)
    clouds = [(scan, 0.4), (scan, 0.05), (golden_components["gs_pc"], float(golden_components["gs_voxel"])),
              (np.array([[0.1, 0.2, 0.3], [5.0, -4.0, 1.0]], np.float32), 0.5),
              ((rng.random((70_000, 3)) * 200 - 100).astype(np.float32), 0.7), (np.zeros((0, 3), np.float32), 0.3)]
    for pts, voxel in clouds:
        for dtype in (torch_cuda.float32, torch_cuda.float64):
            dev = torch_cuda.from_numpy(np.ascontiguousarray(pts)).cuda().to(dtype)
            exact_p, exact_i = (ctx.grid_sample_f64 if dtype == torch_cuda.float64 else ctx.grid_sample)(dev, voxel)
            pad_p, pad_i, count = ctx.grid_sample_padded(dev, voxel)
            v = int(count)
            assert v == exact_p.shape[0] and pad_p.shape[0] == pts.shape[0] and pad_p.dtype == dtype
            assert torch_cuda.equal(pad_p[:v], exact_p) and torch_cuda.equal(pad_i[:v], exact_i)
            assert bool(torch_cuda.isnan(pad_p[v:]).all()) and bool((pad_i[v:] == -1).all())
            if pts.shape[0]:  # ... and the exact one is the reference's order: ascending int64 hash, first point of every voxel
                oi = O.grid_sample(pts.astype(np.float64 if dtype == torch_cuda.float64 else np.float32), voxel)[1]
                assert np.array_equal(exact_i.cpu().numpy(), oi)
    ctx.close()


def test_distortion_filter_and_f64_grid_sample(torch_cuda, O):
    """SURVEY §8f rank 1 (`Distortion` -> `GridSample` on the float64 de-skewed cloud): HIP vs the reference's outputs."""
    import os
    from pylidar_slam_amd.odometry import Distortion, DistortionConfig, GridSample, GridSampleConfig
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "distortion.npz"))
    ctx = _ctx()
    dist = Distortion(DistortionConfig(output_key="distorted"), ctx=ctx)
    gs = GridSample(GridSampleConfig(voxel_size=0.3, pointcloud_key="distorted"), ctx=ctx)
    for name in ("small", "large", "identity", "pure_translation"):
        d = {"numpy_pc": g["pc"], "numpy_pc_timestamps": g["timestamps"], "init_rpose": g[f"{name}_rpose"]}
        dist.filter(d)
        assert d["distorted"].dtype == np.float64
        np.testing.assert_allclose(d["distorted"], g[f"{name}_distorted"], atol=1e-11)
        gs.filter(d)
        np.testing.assert_array_equal(d["sample_indices"], g[f"{name}_sample_indices"])
        np.testing.assert_array_equal(d["sample_points"], d["distorted"][d["sample_indices"]])
    # pass-through cases return the very same array (slam/preprocessing.py:158-163)
    d = {"numpy_pc": g["pc"], "init_rpose": g["small_rpose"]}
    dist.filter(d)
    assert d["distorted"] is g["pc"] or d["distorted"] is d["numpy_pc"]
    d = {"numpy_pc": g["pc"], "numpy_pc_timestamps": g["timestamps"], "init_rpose": None}
    dist.filter(d)
    assert d["distorted"] is d["numpy_pc"]
    # constant timestamps -> alpha = 0
    out = ctx.distort(g["pc"], np.full(g["pc"].shape[0], 3.0), g["small_rpose"])
    np.testing.assert_allclose(out, g["constant_ts_distorted"], atol=1e-12)


def test_nearest_neighbor_and_normals(torch_cuda, O, golden_components):
    g = golden_components
    ctx = _ctx(cell_size=0.5)
    ctx.map_set(g["nn_map"])
    nb, nm, ix = ctx.nearest_neighbor_search(g["nn_queries"], with_index=True)
    bi, _ = O.brute_force_nn(g["nn_queries"], g["nn_map"])
    np.testing.assert_array_equal(ix, bi)  # index-exact
    np.testing.assert_array_equal(nb, g["nn_points"])  # = the reference's neighbours
    dots = np.abs((nm * g["nn_normals"]).sum(axis=1))
    assert dots.min() > 1 - 1e-5, dots.min()
    np.testing.assert_allclose(np.linalg.norm(nm, axis=1), 1.0, atol=1e-5)


@pytest.mark.parametrize("cell,rings", [(0.25, 1), (0.5, 4), (2.0, 2), (8.0, 1)])
def test_nearest_neighbor_exact_for_far_and_sparse_queries(torch_cuda, O, cell, rings):
    """No distance cap (local_map.py:385): queries far from the map go through ring expansion and the exhaustive
    fallback and must still return the exact nearest neighbour."""
    rng = np.random.default_rng(11)
    model = (rng.normal(size=(3000, 3)) * np.array([8.0, 8.0, 1.0])).astype(np.float32)
    q = np.concatenate([rng.normal(size=(500, 3)) * 10, rng.normal(size=(100, 3)) * 200 + 300,
                        model[:50] + 1e-3], axis=0).astype(np.float32)
    ctx = _ctx(cell_size=cell, max_rings=rings)
    ctx.map_set(model)
    _, _, ix = ctx.nearest_neighbor_search(q, with_normals=False, with_index=True)
    bi, bd2 = O.brute_force_nn(q, model)
    d2 = ((q.astype(np.float64) - model[ix].astype(np.float64)) ** 2).sum(axis=1)
    # same point, or an exact-distance tie within f32 rounding of the squared distance
    same = ix == bi
    assert same.mean() > 0.995
    np.testing.assert_allclose(d2[~same], bd2[~same], rtol=2e-6)


def test_tiny_maps_and_duplicates(torch_cuda, O):
    ctx = _ctx()
    model = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 0, 0]], np.float32)  # < k+1 points, duplicate
    ctx.map_set(model)
    q = np.array([[0.9, 0.1, 0.0], [0.1, 0.1, 0.8]], np.float32)
    nb, nm, ix = ctx.nearest_neighbor_search(q, with_index=True)
    np.testing.assert_array_equal(ix, [1, 3])  # ties -> lowest index
    assert np.isfinite(nm).all()
    ctx2 = _ctx()
    with pytest.raises(RuntimeError):
        ctx2.nearest_neighbor_search(q)  # empty map


@pytest.mark.parametrize("scheme", ["default", "least_square", "huber", "exp", "neighborhood", "geman_mcclure",
                                    "square_geman_mcclure", "cauchy"])
def test_gauss_newton_step(torch_cuda, O, golden_components, scheme):
    g = golden_components
    sigma = float(g[f"gn_{scheme}_sigma"])
    ctx = _ctx(scheme=scheme, sigma=sigma)
    pose, dx, loss, neq = ctx.align_point_to_plane(g["nn_points"], g["nn_queries"], g["nn_normals"])
    # vs the reference's own align(): its f32 normal equations carry ~1e-5 relative noise
    np.testing.assert_allclose(dx, g[f"gn_{scheme}_dx"], atol=2e-5, rtol=1e-4)
    assert abs(loss - float(g[f"gn_{scheme}_loss"])) <= 1e-4 * abs(float(g[f"gn_{scheme}_loss"]))
    np.testing.assert_allclose(pose, g[f"gn_{scheme}_mat"], atol=2e-5)
    # vs the oracle with f64 accumulation: the same f32 rows, exact sums
    st = O.gauss_newton_step(g["nn_queries"], g["nn_points"], g["nn_normals"], scheme, sigma, accumulate=np.float64)
    np.testing.assert_allclose(dx, st.dx, atol=2e-7, rtol=2e-5)
    H = np.zeros((6, 6))
    H[np.triu_indices(6)] = neq[:21]
    H = H + np.triu(H, 1).T
    np.testing.assert_allclose(H, st.H, rtol=1e-5, atol=1e-6 * np.abs(st.H).max())
    np.testing.assert_allclose(neq[21:27], st.g, rtol=1e-4, atol=1e-6 * np.abs(st.g).max())
    assert neq[29] == g["nn_points"].shape[0]
    # run-to-run bit reproducibility of the reduction
    _, dx2, loss2, neq2 = ctx.align_point_to_plane(g["nn_points"], g["nn_queries"], g["nn_normals"])
    assert np.array_equal(neq, neq2) and np.array_equal(dx, dx2)


def test_invalid_jacobian_raises_runtime_error(torch_cuda):
    """optimization.py:334-336: |det H| < 1e-7 -> RuntimeError("Invalid Jacobian in Gauss Newton minimization")."""
    ctx = _ctx()
    tgt = np.zeros((10, 3), np.float32)
    tgt[:, 0] = np.arange(10)
    n = np.tile(np.array([[0, 0, 1.0]], np.float32), (10, 1))
    with pytest.raises(RuntimeError, match="Invalid Jacobian"):
        ctx.align_point_to_plane(tgt + np.float32(0.1), tgt, n)


def test_map_update(torch_cuda, O, golden_components):
    g = golden_components
    ctx = _ctx(local_map_size=2)
    c, rel = g["mu_clouds"], g["mu_rel"]
    ctx.map_init()
    assert ctx.map_update(np.eye(4), c[0]) == c[0].shape[0]
    ctx.map_update(rel, c[1])
    ctx.map_update(rel, None)
    ctx.map_update(rel, c[2])
    withnan = c[3].copy()
    ctx.map_update(rel, withnan)
    assert ctx.map_num_clouds() == len(g["mu_counts"])
    assert ctx.map_size() == int(g["mu_counts"].sum())
    np.testing.assert_allclose(ctx.map_points(), g["mu_final"], atol=1e-5)
    # NaN rows are dropped on insert (remove_nan, local_map.py:327)
    withnan[::7, 1] = np.nan
    assert ctx.map_update(rel, withnan) == int((~np.isnan(withnan).any(axis=1)).sum())


@pytest.mark.parametrize("run", ["A_numpy_ls", "B_tensor_gm", "C_numpy_nbh_forced", "D_numpy_huber_forced"])
def test_c1_sequence_matches_reference_and_oracle(torch_cuda, O, golden_c1, c1_scans, run):
    """BASELINE.json configs[0]: 10 synthetic 64x1024 scans, grid_sample preprocessing, CV init — the MI355X odometry
    behind the reference's plugin surface reproduces the reference CPU poses frame by frame."""
    from pylidar_slam_amd.odometry import (ConstantVelocityInitialization, GridSample, GridSampleConfig,
                                           MI355XICPConfig, MI355XICPFrameToModel, SphericalProjector)
    g = golden_c1
    scans, _ = c1_scans
    mode, scheme, sigma, iters, thr = (str(v) for v in g[f"{run}_cfg"])
    h, w = (int(v) for v in g["hw"])
    cfg = MI355XICPConfig(max_num_alignments=int(iters), threshold_delta_pose=float(thr),
                          data_key="sample_points" if mode == "numpy" else "input_data",
                          alignment=dict(mode="point_to_plane_gauss_newton",
                                         gauss_newton_config=dict(max_iters=1, scheme=scheme, sigma=float(sigma))))
    odo = MI355XICPFrameToModel(cfg, projector=SphericalProjector(h, w), device=torch_cuda.device("cuda:0"))
    odo.init()
    init = ConstantVelocityInitialization()
    init.init()
    gs = GridSample(GridSampleConfig(voxel_size=0.3, pointcloud_key="numpy_pc"), ctx=odo.ctx)
    worst = (0.0, 0.0)
    for f, s in enumerate(scans):
        d = {"numpy_pc": s}
        init.next_frame(d)
        gs.filter(d)
        assert d["sample_points"].shape[0] == int(g[f"{run}_counts"][f])
        if mode == "tensor":
            d["input_data"] = torch_cuda.from_numpy(d["sample_points"])
        odo.process_next_frame(d)
        if f == 0:
            assert "odometry_pose" not in d  # frame 0 writes nothing (icp_odometry.py:171-181)
            continue
        init.save_real_motion(d["odometry_pose"], d)
        dt, dr = O.pose_error(d["odometry_pose"], g[f"{run}_rel"][f])
        worst = (max(worst[0], dt), max(worst[1], dr))
        assert dt < 1e-4 and dr < 1e-4, (run, f, dt, dr)
        assert d["odometry_pose"].dtype == np.float32 and d["odometry_pose"].shape == (4, 4)
        if float(thr) == 0.0:
            assert odo.last_result.iterations == int(g[f"{run}_iters"][f])
    assert odo.ctx.map_size() == int(g[f"{run}_map_size"])
    rel = odo.get_relative_poses()
    assert rel.shape == (len(scans), 4, 4) and np.array_equal(rel[0], np.eye(4, dtype=np.float32))
    np.testing.assert_allclose(np.stack(odo.absolute_poses), g[f"{run}_abs"], atol=5e-4)
    print(f"{run}: worst per-frame error vs reference {worst[0]:.2e} m / {worst[1]:.2e} rad")


def test_register_masks_nan_and_null_rows(torch_cuda, O, golden_components):
    g = golden_components
    ctx = _ctx(max_num_alignments=6, threshold_delta_pose=0.0)
    ctx.map_set(g["nn_map"])
    q = g["nn_queries"]
    base = ctx.register(q)
    dirty = np.concatenate([q[:100], np.full((7, 3), np.nan, np.float32), q[100:], np.zeros((5, 3), np.float32)])
    r1 = ctx.register(dirty, skip_null=True)
    assert r1.num_targets == q.shape[0] == base.num_targets
    np.testing.assert_allclose(r1.pose, base.pose, atol=1e-6)
    np.testing.assert_allclose(r1.losses, base.losses, rtol=1e-9)
    empty = ctx.register(np.full((4, 3), np.nan, np.float32))
    assert empty.num_targets == 0 and empty.converged and np.array_equal(empty.pose, np.eye(4, dtype=np.float32))


def test_fresh_context_after_a_large_one_reads_no_stale_memory(torch_cuda, O, golden_components):
    """Regression: device buffers recycled from a freed (larger) context hold garbage; nothing may depend on
    zero-initialised memory (the padding entry of the neighbour rows once did)."""
    rng = np.random.default_rng(0)
    big = _ctx()
    big.map_set((rng.normal(size=(300_000, 3)) * 20).astype(np.float32))
    big.register((rng.normal(size=(50_000, 3)) * 20).astype(np.float32))
    big.close()
    g = golden_components
    ctx = _ctx(max_num_alignments=3, threshold_delta_pose=0.0)
    ctx.map_set(g["nn_map"])
    _, _, ix = ctx.nearest_neighbor_search(g["nn_queries"], with_index=True)
    np.testing.assert_array_equal(ix, O.brute_force_nn(g["nn_queries"], g["nn_map"])[0])
    assert ctx.register(g["nn_queries"]).iterations == 3


def test_schedule_options_are_bit_identical(torch_cuda):
    """The tuning options of `icp_set_option` are pure schedule changes: the per-iteration NN cache (skip the search
    when the cached neighbour is provably still the nearest), the in-block compaction of its misses, the 64-register
    build, the cross-frame seeds, the cell size and the fallback of the grid build's one-launch scan (every tile summing
    the table itself instead of waiting for its predecessors) give the same poses, losses and maps bit for bit;
    the unfused path shares everything but the reduction order (1e-6 relative)."""
    from pylidar_slam_amd.synthetic import SceneConfig, make_fixed_map, make_sequence
    cfg = SceneConfig(height=32, width=1024)
    scans, poses = make_sequence(cfg, 7)
    model = make_fixed_map(cfg, scans[:4], poses[:4], ref_frame=3, num_points=30_000)
    variants = {"default": {}, "nocache": {"nn_cache": 0}, "cache_noseed": {"nn_cache": 1},
                "sparse_build": {"iterate_dense": 0},
                "no_wave_search": {"wave_misses": 0, "wave_misses_dense": 0},
                "never_narrow": {"narrow_from": -1}, "narrow_early": {"narrow_from": 1},
                "wave_search_always": {"wave_misses": 128, "wave_misses_dense": 128},
                "no_prune_guard": {"prune_guard": 0.0}, "no_refresh": {"refresh_margin": 0.0},
                "big_cells": {"target_occupancy": 40}, "small_cells": {"target_occupancy": 2},
                "no_frame_seed": {"frame_seed": 0},
                "lanes2": {"knn_lanes": 2}, "scan_gives_up": {"scan_poll_limit": 0},
                "cell_lists": {"cell_lists": 1},  # the slots a build claims listed: the next build empties those, the starts scanned over the lists
                "no_lead_solve": {"lead_solve": 0},  # every solve in a launch of its own (round 2's schedule)
                "no_flat_rows": {"flat_rows": 0},    # neighbour cells walked lane by lane (round 2's schedule)
                "flat_list": {"flat_rows": 1},       # surviving neighbour cells laid end to end (the first half of round 3)
                "no_xcd_sectors": {"xcd_sectors": 0},  # consecutive queries in consecutive workgroups (no XCD sectors)
                "no_lead_after_dense": {"lead_after_dense": 0},  # the dense launches keep their own solving launch
                "no_lead_never_narrow": {"lead_solve": 0, "narrow_from": -1},
                "no_ball_search": {"ball_search": 0},  # every miss by the 4-lane / whole-wave searches (round 3's schedule)
                "no_ball_never_narrow": {"ball_search": 0, "narrow_from": -1},
                "narrow_always": {"narrow_from": 0},   # the 512-query shape from the first iteration on (ball search)
                "narrow_always_nocache": {"narrow_from": 0, "nn_cache": 0},
                "ball_no_guard": {"prune_guard": 0.0, "narrow_from": 0},
                "ball_one_lane": {"ball_lanes": 1}, "ball_two_lanes": {"ball_lanes": 2},
                "never_wide": {"wide_until": 0}, "always_wide": {"wide_until": 99},
                # what the ball search hands back: 4 lanes / a wave each (far_lanes 0), 16 lanes however many, ... and a small ball cap
                "no_far_lanes": {"far_lanes": 0}, "far_lanes_two_passes": {"far_max": 512},
                "far_lanes_small_balls": {"far_max": 512, "ball_max": 16},
                # the resident tail (off by default) from iteration 3, 1, 5 and 11 (one iteration left: no tail)
                "tail_from_3": {"resident_tail": 3}, "tail_from_1": {"resident_tail": 1, "wide_until": 0},
                "tail_from_5": {"resident_tail": 5}, "tail_from_11": {"resident_tail": 11},
                "tail_no_cache_seed": {"resident_tail": 3, "nn_cache": 1},
                "tail_no_guard": {"resident_tail": 3, "prune_guard": 0.0, "refresh_margin": 0.0},
                "tail_small_scans_only": {"resident_tail": 3, "resident_tail_max_blocks": 8},  # (this scan has 64 workgroups: no tail)
                # normals on demand inside the fused kernel (by default only where the map dwarfs the scan) / never
                # (compared with `no_carry`: a normal estimated on first touch in a LATER frame is estimated from the re-expressed
                # points, where the default schedule has carried the first frame's estimate over by rotation: rounding apart)
                # round 6 (both off by default): hit records / the late kernel from the fourth launch on, from the second, in
                # its 80-register build, behind the 512-thread shape
                "cell_lists": {"cell_lists": 1}, "no_ball_empty": {"ball_empty": 0},
                "hit_records": {"hit_records": 1}, "late_kernel": {"hit_records": 1, "late_from": 3},
                "late_from_1": {"hit_records": 1, "late_from": 1},
                "late_80_registers": {"hit_records": 1, "late_from": 3, "late_waves": 6},
                "late_never_wide": {"hit_records": 1, "late_from": 1, "wide_until": 0},
                "no_carry": {"carry_normals": 0}, "lazy_fused": {"lazy_fused": 2, "carry_normals": 0},
                # (the stragglers of the eager normals on the map stream instead of inside the estimating launch)
                "no_carry_tail_stream": {"carry_normals": 0, "normals_tail_stream": 1},
                "no_carry_small_cells": {"carry_normals": 0, "target_occupancy": 2},  # (tiny cells: many stragglers)
                "no_carry_small_cells_tail_stream": {"carry_normals": 0, "target_occupancy": 2, "normals_tail_stream": 1},
                # round 6 (off by default): the stragglers in a launch of their own, sixteen lanes each, against a wave of their workgroup each
                "no_carry_listed_stragglers": {"carry_normals": 0, "normals_list": 1},
                "no_carry_small_cells_listed_stragglers": {"carry_normals": 0, "target_occupancy": 2, "normals_list": 1},
                "lazy_fused_no_lead": {"lazy_fused": 2, "lead_solve": 0, "carry_normals": 0},
                "lazy_fused_nocache": {"lazy_fused": 2, "nn_cache": 0, "carry_normals": 0}, "never_lazy_fused": {"lazy_fused": 0},
                "lazy_fused_carried": {"lazy_fused": 2},  # (held to 1e-6 below, not to the bit)
                "unfused": {"fuse_iteration": 0}}
    results = {}
    for name, opts in variants.items():
        ctx = _ctx(height=32, width=1024, max_num_alignments=12, threshold_delta_pose=0.0, scheme="geman_mcclure",
                   sigma=0.3)
        for k, v in opts.items():
            ctx.set_option(k, v)
        ctx.map_set(model)
        frames, init = [], None
        for f in (4, 5, 6):  # three consecutive frames: the second and third start from the first's neighbours
            ctx.register_launch(scans[f], init)
            ctx.map_update(None, None)
            r = ctx.register_end()
            frames.append(r)
            init = r.pose
        _, nrm, _ = ctx.nearest_neighbor_search(scans[6][::7])
        results[name] = (frames, ctx.map_points(), nrm)
        assert ctx.handoff_fallbacks() == 0, name  # (a hand-off that timed out would be repaired silently: same bits, 50 ms late)
        ctx.close()
    with pytest.raises(AssertionError):
        c = _ctx()
        c.set_option("no_such_option", 1)
    problems = []
    for name, (frames, mp, nrm) in results.items():
        ref_frames, ref_map, ref_nrm = results["no_carry" if variants[name].get("carry_normals", 1) == 0 else "default"]
        if name == "no_carry":  # (the reference's schedule against the carried one: rounding apart)
            ref_frames, ref_map, ref_nrm = results["default"]
        if name in ("no_carry", "lazy_fused_carried"):
            for r, ref in zip(frames, ref_frames):
                np.testing.assert_allclose(r.pose, ref.pose, atol=1e-6)
                np.testing.assert_allclose(r.losses, ref.losses, rtol=1e-5)
            continue
        for f, (r, ref) in enumerate(zip(frames, ref_frames)):
            assert r.iterations == ref.iterations == 12
            if name == "unfused":
                np.testing.assert_allclose(r.pose, ref.pose, atol=2e-7)
                np.testing.assert_allclose(r.losses, ref.losses, rtol=1e-6)
            elif not (np.array_equal(r.pose, ref.pose) and np.array_equal(r.losses, ref.losses)
                      and np.array_equal(r.dx, ref.dx)):
                first = int(np.argmax(r.losses != ref.losses)) if (r.losses != ref.losses).any() else -1
                problems.append(f"{name}: frame {f} differs (first loss mismatch at iteration {first}, "
                                f"max |dpose| {np.abs(r.pose - ref.pose).max():.1e})")
        if name != "unfused":
            if not np.array_equal(mp, ref_map):
                problems.append(f"{name}: map differs")
            if not np.array_equal(nrm, ref_nrm):
                problems.append(f"{name}: normals differ ({np.abs(nrm - ref_nrm).max():.1e})")
    assert not problems, "\n".join(problems)


def test_lazy_and_eager_normals_give_the_same_registration(torch_cuda):
    """Normals of the whole map at once (maps up to `eager_normals_limit` points, or at most twice the scan) or lazily for
    the map points the scan touches (`KdTreeLocalMap.__get_normals`' cache semantics, local_map.py:397-422): the same
    normals, hence the same registration up to the order of the float64 sums — on a scan a quarter the size of the map,
    with the forced iteration count and with a live threshold (polled, launched in chunks)."""
    from pylidar_slam_amd.synthetic import SceneConfig, make_fixed_map, make_sequence
    cfg = SceneConfig(height=32, width=1024)
    scans, poses = make_sequence(cfg, 6)
    model = make_fixed_map(cfg, scans[:4], poses[:4], ref_frame=3, num_points=30_000)
    targets = np.ascontiguousarray(scans[5][::4])
    for threshold in (0.0, 1.0e-4):
        got = {}
        for name, limit in (("eager", 1 << 20), ("lazy", 0)):
            ctx = _ctx(height=32, width=1024, max_num_alignments=15, threshold_delta_pose=threshold,
                       scheme="geman_mcclure", sigma=0.3)
            ctx.set_option("eager_normals_limit", limit)
            ctx.map_set(model)
            sync = ctx.register(targets)
            ctx.register_launch(targets, None)
            launched = ctx.register_end()
            assert np.array_equal(sync.pose, launched.pose) and sync.iterations == launched.iterations
            got[name] = sync
            ctx.close()
        assert got["eager"].normals_computed == model.shape[0]
        assert 0 < got["lazy"].normals_computed < model.shape[0] // 2
        assert got["eager"].iterations == got["lazy"].iterations
        np.testing.assert_allclose(got["lazy"].pose, got["eager"].pose, atol=2e-7)
        np.testing.assert_allclose(got["lazy"].losses, got["eager"].losses, rtol=1e-6)


def test_timed_out_handoff_finishes_on_per_iteration_launches(torch_cuda):
    """The resident tail and the lead launches wait for one another inside a launch; every wait has a wall-clock bound
    ("lead_timeout_ms").  With the bound at one tick of the clock the first wait that is not served at once runs out: the
    registration must be FINISHED on per-iteration launches — same poses, losses and steps bit for bit as the schedule
    without hand-offs — the pose-only map update enqueued behind it must move the map by the FINAL pose, the context must
    keep to per-iteration launches afterwards (`icp_handoff_fallbacks` counts once), with the forced iteration count and
    with a live stop threshold."""
    from pylidar_slam_amd.synthetic import SceneConfig, make_fixed_map, make_sequence
    cfg = SceneConfig(height=32, width=1024)
    scans, poses = make_sequence(cfg, 7)
    model = make_fixed_map(cfg, scans[:4], poses[:4], ref_frame=3, num_points=30_000)
    for threshold in (0.0, 1.0e-4):
        got = {}
        for name, opts in (("plain", {"lead_solve": 0}), ("timed_out", {"lead_timeout_ms": 1.0e-5, "resident_tail": 3}),
                           ("timed_out_no_tail", {"lead_timeout_ms": 1.0e-5})):
            ctx = _ctx(height=32, width=1024, max_num_alignments=12, threshold_delta_pose=threshold, scheme="geman_mcclure",
                       sigma=0.3)
            for k, v in opts.items():
                ctx.set_option(k, v)
            ctx.map_set(model)
            frames, init = [], None
            for f in (4, 5, 6):
                ctx.register_launch(scans[f], init)
                ctx.map_update(None, None)
                r = ctx.register_end()
                frames.append(r)
                init = r.pose
            got[name] = (frames, ctx.map_points(), ctx.handoff_fallbacks())
            ctx.close()
        assert got["plain"][2] == 0
        for name in ("timed_out", "timed_out_no_tail"):
            frames, mp, fallbacks = got[name]
            assert fallbacks == 1, (name, fallbacks)  # the first frame fell back, the context kept to plain launches
            for r, ref in zip(frames, got["plain"][0]):
                assert r.iterations == ref.iterations and r.converged == ref.converged
                assert np.array_equal(r.pose, ref.pose) and np.array_equal(r.losses, ref.losses) and np.array_equal(r.dx, ref.dx)
            assert np.array_equal(mp, got["plain"][1])


def test_device_side_initial_pose_tensor_into_the_plugin(torch_cuda):
    """BASELINE configs[4] hand-off (VERDICT r5 Missing #7): the initial pose comes from a network on the device — a cuda
    tensor, [1,4,4], bfloat16 — through `data_dict["init_rpose"]` (slam/initialization.py:222-283 -> icp_odometry.py:147-154).
    The plugin must take it (rounded to what bfloat16 holds) exactly as it takes the same matrix as a host float32 array."""
    from pylidar_slam_amd.odometry import MI355XICPConfig, MI355XICPFrameToModel, SphericalProjector
    from pylidar_slam_amd.synthetic import SceneConfig, make_sequence
    scans, gt = make_sequence(SceneConfig(height=32, width=1024), 3)
    guess = (np.linalg.inv(gt[0]) @ gt[1]).astype(np.float32)
    as_bf16 = torch_cuda.from_numpy(guess).cuda().to(torch_cuda.bfloat16).reshape(1, 4, 4)
    rounded = as_bf16.to(torch_cuda.float32).cpu().numpy().reshape(4, 4)
    poses = []
    for init in (as_bf16, rounded):
        cfg = MI355XICPConfig(max_num_alignments=6, threshold_delta_pose=0.0, data_key="numpy_pc")
        odo = MI355XICPFrameToModel(cfg, projector=SphericalProjector(32, 1024), device=torch_cuda.device("cuda:0"))
        odo.init()
        odo.process_next_frame({"numpy_pc": scans[0]})
        d = {"numpy_pc": scans[1], "init_rpose": init}
        odo.process_next_frame(d)
        poses.append(d["odometry_pose"].copy())
        odo.ctx.close()
    assert np.array_equal(poses[0], poses[1])
    assert np.linalg.norm(poses[0][:3, 3] - guess[:3, 3]) < 0.15  # (a 0.4 m motion: the guess was taken; six iterations against one sparse scan)


def test_plugin_warns_about_handoff_fallbacks(torch_cuda):
    """VERDICT r5 Weak #14: a hand-off that times out is repaired silently by the library (`icp_handoff_fallbacks`); the plugin
    says so once per sequence (at the next `init()` / `get_relative_poses()`)."""
    import warnings
    from pylidar_slam_amd.odometry import MI355XICPConfig, MI355XICPFrameToModel, SphericalProjector
    from pylidar_slam_amd.synthetic import SceneConfig, make_sequence
    scans, _ = make_sequence(SceneConfig(height=32, width=1024), 3)
    cfg = MI355XICPConfig(max_num_alignments=8, threshold_delta_pose=0.0, data_key="numpy_pc")
    odo = MI355XICPFrameToModel(cfg, projector=SphericalProjector(32, 1024), device=torch_cuda.device("cuda:0"))
    odo.init()
    odo.ctx.set_option("lead_timeout_ms", 1.0e-5)  # (one tick of the wall clock: the first lead launch gives up)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        for s in scans:
            odo.process_next_frame({"numpy_pc": s})
        assert odo.ctx.handoff_fallbacks() >= 1
        rel = odo.get_relative_poses()
        assert rel is not None and rel.shape[0] == 3
        said = [w for w in caught if "timed-out pose hand-off" in str(w.message)]
        assert len(said) == 1, [str(w.message) for w in caught]
        odo.get_relative_poses()
        odo.init()  # nothing new since: said once
        assert len([w for w in caught if "timed-out pose hand-off" in str(w.message)]) == 1
    odo.ctx.close()


def test_carried_normals_equal_reestimated_ones(torch_cuda):
    """Option "carry_normals" (default 1): a pose-only map update rotates the normals the grid holds with the points instead
    of clearing them; 0 is the reference's schedule (local_map.py:365-369: every build_model zeroes the cache, the next
    touch re-estimates from the re-expressed points).  Over a chain of frames with pose-only updates the two agree to
    float32 rounding: normals (|dot| > 1 - 1e-6), poses (1e-6 m / rad), losses; the carried schedule estimates nothing
    after the first frame, an update that inserts a cloud clears the cache under both."""
    from pylidar_slam_amd.synthetic import SceneConfig, make_fixed_map, make_sequence
    cfg = SceneConfig(height=32, width=1024)
    scans, poses = make_sequence(cfg, 9)
    model = make_fixed_map(cfg, scans[:4], poses[:4], ref_frame=3, num_points=30_000)
    got = {}
    for carry in (0, 1):  # (the reference-exact schedule first)
        ctx = _ctx(height=32, width=1024, max_num_alignments=10, threshold_delta_pose=0.0, scheme="geman_mcclure", sigma=0.3)
        ctx.set_option("carry_normals", carry)
        ctx.map_set(model)
        frames, init = [], None
        for f in (4, 5, 6, 7):
            ctx.register_launch(scans[f], init)
            ctx.map_update(None, None)
            r = ctx.register_end()
            frames.append(r)
            init = r.pose
        q, nrm, ix = ctx.nearest_neighbor_search(scans[7][::5], with_index=True)
        # ... then an insertion: the cache is cleared whatever the option says, every normal estimated again
        ctx.map_update(np.eye(4, dtype=np.float32), scans[8][::16])
        r8 = ctx.register(scans[8], init)
        got[carry] = (frames, nrm, ix, ctx.map_points(), r8)
        ctx.close()
    (f0, n0, i0, m0, a0), (f1, n1, i1, m1, a1) = got[0], got[1]
    assert f0[0].normals_computed == model.shape[0] and f1[0].normals_computed == model.shape[0]
    assert all(r.normals_computed == model.shape[0] for r in f0[1:])  # the reference's schedule re-estimates every frame
    assert all(r.normals_computed == 0 for r in f1[1:])               # carried: nothing to estimate
    # behind an insertion: all of them, both ways (the reference's schedule also reports the estimation behind the last
    # pose-only update, which no registration had collected yet)
    assert a1.normals_computed == m0.shape[0] and a0.normals_computed == m0.shape[0] + model.shape[0]
    for r0, r1 in zip(f0 + [a0], f1 + [a1]):
        assert np.abs(r0.pose - r1.pose).max() < 1e-6, np.abs(r0.pose - r1.pose).max()
        np.testing.assert_allclose(r1.losses, r0.losses, rtol=1e-5)
    assert np.abs(m0 - m1).max() < 1e-5  # (the maps moved by poses that differ by 1e-7)
    same = i0 == i1
    assert same.mean() > 0.999
    dots = np.abs((n0[same] * n1[same]).sum(axis=1))
    assert dots.min() > 1 - 1e-6, dots.min()
    np.testing.assert_allclose(np.linalg.norm(n1, axis=1), 1.0, atol=1e-6)


def test_carried_normals_over_a_long_chain_at_benchmark_size(torch_cuda, O):
    """VERDICT r5 Weak #1(i): the headline workload runs `carry_normals` (the library default) over 85+ chained pose-only
    updates — the normals rotated, conditionally re-normalised and re-filed every frame — while the test above chains four
    frames.  Here: the headline's own arrangement (64x2048 scans against a 100 000-point map of OTHER scans, tracked back
    and forth from a constant-velocity guess, 20 forced iterations) for 64 chained frames with carry_normals 1 and 0 (the
    reference's schedule, local_map.py:365-369): every frame's pose within 1e-5 m / 1e-5 rad of the re-estimating chain,
    the carried normals still unit vectors and parallel to freshly estimated ones (|dot| > 1 - 1e-5 for 99.9 % of them,
    > 0.99 for all) at the end."""
    from pylidar_slam_amd.synthetic import SceneConfig, make_fixed_map, make_sequence, rotate_rows
    cfg = SceneConfig(height=64, width=2048, step=0.2, yaw_rate=0.005)
    scans, poses = make_sequence(cfg, 16)
    even = list(range(0, 16, 2))
    model = make_fixed_map(cfg, [scans[f] for f in even], poses[even], ref_frame=0, num_points=100_000)
    rel = np.linalg.inv(poses[1]) @ poses[0]
    model = (rotate_rows(model.astype(np.float64), rel[:3, :3]) + rel[:3, 3]).astype(np.float32)
    order = list(range(3, 16, 2)) + list(range(13, 0, -2))
    dev = {f: torch_cuda.from_numpy(scans[f]).cuda() for f in range(1, 16, 2)}
    frames = 64
    got = {}
    for carry in (0, 1):
        ctx = _ctx(height=64, width=2048, max_num_alignments=20, threshold_delta_pose=0.0, scheme="geman_mcclure", sigma=0.3)
        ctx.set_option("carry_normals", carry)
        ctx.map_set(torch_cuda.from_numpy(model).cuda())
        out, init = [], None
        for k in range(frames):
            ctx.register_launch(dev[order[k % len(order)]], init)
            ctx.map_update(None, None)
            r = ctx.register_end()
            out.append(r)
            init = r.pose
        q, nrm, ix = ctx.nearest_neighbor_search(scans[order[(frames - 1) % len(order)]][::7], with_index=True)
        got[carry] = (out, nrm, ix, ctx.handoff_fallbacks())
        ctx.close()
    (f0, n0, i0, fb0), (f1, n1, i1, fb1) = got[0], got[1]
    assert fb0 == 0 and fb1 == 0
    assert all(r.normals_computed == 0 for r in f1[1:]) and all(r.normals_computed == model.shape[0] for r in f0[1:])
    worst_t = worst_r = 0.0
    for r0, r1 in zip(f0, f1):
        assert r0.iterations == r1.iterations == 20
        dt, dr = O.pose_error(r0.pose, r1.pose)
        worst_t, worst_r = max(worst_t, dt), max(worst_r, dr)
    assert worst_t < 1e-5 and worst_r < 1e-5, (worst_t, worst_r)
    same = i0 == i1
    assert same.mean() > 0.999
    dots = np.abs((n0[same] * n1[same]).sum(axis=1))
    # (a neighbourhood whose two smallest eigenvalues nearly coincide — an edge, a corner — amplifies the 1e-7 rounding of
    # the re-expressed points into a visibly different eigenvector when it is estimated AGAIN; measured: 0.99946 at worst)
    assert (dots > 1 - 1e-5).mean() > 0.999 and dots.min() > 0.99, ((dots > 1 - 1e-5).mean(), dots.min())
    np.testing.assert_allclose(np.linalg.norm(n1, axis=1), 1.0, atol=2e-6)


def _knn_clouds():
    """Point sets that stress the k-nearest-neighbour search behind the normals: a LiDAR map, a volume, exact duplicates
    (ties on the k-th distance by the dozen), a lattice (every distance tied), isolated points, fewer than k + 1 points."""
    from pylidar_slam_amd.synthetic import SceneConfig, make_fixed_map, make_sequence
    rng = np.random.default_rng(21)
    cfg = SceneConfig(height=32, width=1024)
    scans, poses = make_sequence(cfg, 4)
    clouds = {"lidar_map": make_fixed_map(cfg, scans, poses, ref_frame=3, num_points=30_000)}
    clouds["volume"] = (rng.random((20_000, 3)) * np.array([30.0, 30.0, 6.0])).astype(np.float32)
    base = (rng.normal(size=(1500, 3)) * np.array([10.0, 10.0, 1.0])).astype(np.float32)
    clouds["duplicates"] = np.repeat(base, rng.integers(1, 24, size=base.shape[0]), axis=0)
    ax = np.arange(24, dtype=np.float32) * 0.25
    clouds["lattice"] = np.stack(np.meshgrid(ax, ax, ax[:12], indexing="ij"), axis=-1).reshape(-1, 3)
    clouds["isolated"] = np.concatenate([base[:800], (rng.normal(size=(40, 3)) * 300 + 500).astype(np.float32)])
    clouds["seven"] = base[:7].copy()
    return clouds


@pytest.mark.parametrize("neighbors", [10, 5, 20])
def test_knn_normals_on_stress_clouds(torch_cuda, O, neighbors):
    """The three kNN-normal kernels — sharded (by original index, one and two ranks), lazy worklist (a bare search) and
    eager (a registration with at least half as many targets as map points) — give the same normals bit for bit on
    point sets that stress the selection (ties by the dozen, isolated points through the coarse level, fewer than k + 1
    points), and those normals are the reference's wherever the neighbourhood is unambiguous: `knn_normals` of the
    oracle restates local_map.py:397-422, a covariance whose two smallest eigenvalues (nearly) coincide has no defined
    normal, and ties on the k-th distance make the neighbour set itself a matter of the tree's tie order."""
    from scipy.spatial import cKDTree
    for name, cloud in _knn_clouds().items():
        ctx = _ctx(num_neighbors_normals=neighbors)
        ctx.map_set(cloud)
        whole = ctx.map_normals_owned(0, 1).cpu().numpy()
        halves = (ctx.map_normals_owned(0, 2) + ctx.map_normals_owned(1, 2)).cpu().numpy()
        ctx.close()
        assert np.array_equal(whole[:, 3], np.ones(cloud.shape[0], np.float32)), name
        assert np.isfinite(whole).all(), name
        assert np.array_equal(whole, halves), name
        if cloud.shape[0] > neighbors + 1 and name in ("lidar_map", "volume", "isolated"):
            tree = cKDTree(cloud.astype(np.float64))
            idx = np.arange(0, cloud.shape[0], 7)
            ref = O.knn_normals(cloud, tree, idx, k=neighbors)
            d, _ = tree.query(cloud[idx].astype(np.float64), k=neighbors + 2)
            pts = cloud[idx]
            _, nb = tree.query(pts.astype(np.float64), k=neighbors + 1)
            c = (cloud[nb[:, 1:].reshape(-1)].reshape(-1, neighbors, 3) - pts[:, None, :]).astype(np.float64)
            ev = np.linalg.eigvalsh((c[:, :, :, None] * c[:, :, None, :]).mean(axis=1))
            clear = (ev[:, 1] - ev[:, 0] > 1e-3 * ev[:, 2]) & (d[:, -1] - d[:, -2] > 1e-6 * d[:, -1])
            assert clear.mean() > 0.5, (name, clear.mean())
            dots = np.abs((whole[idx, :3] * ref).sum(axis=1))
            assert dots[clear].min() > 1 - 1e-4, (name, dots[clear].min())
        if cloud.shape[0] < 100:
            continue
        ctx = _ctx(num_neighbors_normals=neighbors, max_num_alignments=1, threshold_delta_pose=0.0)
        ctx.map_set(cloud)
        _, lazy, ix = ctx.nearest_neighbor_search(cloud[::3], with_index=True)
        np.testing.assert_array_equal(lazy, whole[ix, :3], err_msg=f"{name} lazy")
        try:
            ctx.register(cloud)  # eager normals of every map point, whatever becomes of the degenerate alignment
        except RuntimeError:
            pass
        _, eager, ix = ctx.nearest_neighbor_search(cloud, with_index=True)
        np.testing.assert_array_equal(eager, whole[ix, :3], err_msg=f"{name} eager")
        ctx.close()
        # the eager kernels of the earlier rounds behind their options: four lanes per point over the neighbourhood lists
        # (hoods 1), the row walk (hoods 0) — the default (hoods 2) is two lanes per point + a wave of the workgroup per
        # straggler; round 6's stragglers in a launch of their own, sixteen lanes each ("normals_list" 1): hoods = -2 below
        for hoods in (1, 0, -2):
            ctx = _ctx(num_neighbors_normals=neighbors, max_num_alignments=1, threshold_delta_pose=0.0)
            if hoods < 0:
                ctx.set_option("normals_list", 1)
            ctx.set_option("hoods", abs(hoods))
            ctx.map_set(cloud)
            owned = ctx.map_normals_owned(0, 1).cpu().numpy()
            np.testing.assert_array_equal(owned, whole, err_msg=f"{name} owned, hoods {hoods}")
            try:
                ctx.register(cloud)
            except RuntimeError:
                pass
            _, eager, ix = ctx.nearest_neighbor_search(cloud, with_index=True)
            np.testing.assert_array_equal(eager, whole[ix, :3], err_msg=f"{name} eager, hoods {hoods}")
            ctx.close()


@pytest.mark.parametrize("cell_lists", [0, 1])
def test_grid_build_with_more_scan_tiles_than_resident_workgroups(torch_cuda, O, cell_lists):
    """The table scan of the grid build ("cell_lists" 0) is one launch with decoupled look-back: a tile waits for the
    descriptors of its predecessors.  3 M points give 8192 tiles — four times what the chip keeps resident — so most tiles are
    dispatched while others spin; the grid must come out right (exact neighbours against a kd-tree, every point its own
    nearest neighbour) and a second build on the same context (stale descriptors of the first) as well.  With cell lists (the
    default) the same cloud is 3 M cells of one point: 366 chunks of the list scan, and the second build empties the table
    through the first one's lists (off by default)."""
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(3)
    model = (rng.random((3_000_000, 3)) * np.array([400.0, 400.0, 20.0])).astype(np.float32)
    ctx = _ctx()
    ctx.set_option("cell_lists", cell_lists)
    for build in range(2):
        ctx.map_set(model)
        q = (rng.random((2000, 3)) * np.array([400.0, 400.0, 20.0])).astype(np.float32)
        _, _, ix = ctx.nearest_neighbor_search(q, with_normals=False, with_index=True)
        _, best = cKDTree(model.astype(np.float64)).query(q.astype(np.float64), workers=-1)
        same = ix == best
        dq = ((q - model[ix]).astype(np.float64) ** 2).sum(-1)
        db = ((q - model[best]).astype(np.float64) ** 2).sum(-1)
        assert same.mean() > 0.99 and np.allclose(dq[~same], db[~same], rtol=2e-6), (build, same.mean())
        sub = model[::997]
        _, _, own = ctx.nearest_neighbor_search(sub, with_normals=False, with_index=True)
        assert np.array_equal(model[own], sub), build  # itself, or an exact duplicate with a smaller index
        model = model[::-1].copy()  # another order -> other table contents for the second build
    ctx.close()


def test_split_iteration_seam_equals_fused_register(torch_cuda, golden_components):
    """The multi-GPU seam (accumulate -> [all-reduce] -> solve) with world size 1 reproduces icp_register bit for bit,
    and two half-slices summed by hand give the same normal equations as the whole scan."""
    g = golden_components
    ctx = _ctx(max_num_alignments=5, threshold_delta_pose=0.0)
    ctx.map_set(g["nn_map"])
    q = g["nn_queries"]
    fused = ctx.register(q)
    neq = ctx.normal_equations_tensor()
    ctx.register_begin(q)
    for _ in range(5):
        ctx.iteration_accumulate()
        ctx.iteration_solve()
    split = ctx.register_end()
    assert np.array_equal(split.pose, fused.pose) and np.array_equal(split.losses, fused.losses)
    # slices
    ctx.register_begin(q)
    ctx.iteration_accumulate()
    whole = neq.clone()
    ctx.register_end()
    parts = []
    for sl in (q[:1000], q[1000:]):
        ctx.register_begin(sl)
        ctx.iteration_accumulate()
        parts.append(neq.clone())
        ctx.register_end()
    np.testing.assert_allclose((parts[0] + parts[1]).cpu().numpy(), whole.cpu().numpy(), rtol=1e-12)


@pytest.mark.parametrize("threshold", [0.0, 1e-3])
def test_async_register_and_device_pose_map_update_are_bit_identical(torch_cuda, golden_components, threshold):
    """icp_register_launch + icp_map_update(rel_pose = NULL) + icp_register_end (no host round trip between the
    registration and the map re-expression) against the synchronous sequence: same poses, losses and map, frame after
    frame — including the early-stop case where the launches behind the stop are device-side no-ops."""
    g = golden_components
    q = g["nn_queries"]
    rng = np.random.default_rng(5)
    frames = [q, (q + rng.normal(0, 0.01, q.shape)).astype(np.float32), q[::2].copy()]
    a = _ctx(max_num_alignments=8, threshold_delta_pose=threshold)
    b = _ctx(max_num_alignments=8, threshold_delta_pose=threshold)
    a.map_set(g["nn_map"])
    b.map_set(g["nn_map"])
    with pytest.raises(AssertionError):
        b.map_update(None, None)  # no registration yet: nothing to take the pose from
    init = None
    for f in frames:
        ra = a.register(f, init)
        a.map_update(ra.pose, None)
        b.register_launch(f, init)
        b.map_update(None, None)
        rb = b.register_end()
        assert np.array_equal(ra.pose, rb.pose) and ra.iterations == rb.iterations
        assert np.array_equal(ra.losses, rb.losses) and np.array_equal(ra.dx, rb.dx)
        assert np.array_equal(a.map_points(), b.map_points())
        init = ra.pose
    assert ra.iterations == 8 if threshold == 0.0 else ra.iterations <= 8


def _c2_inputs():
    """Scan / map pair of the C2 parity tests (the inputs oracle/make_golden_c2.py ran the reference on)."""
    from pylidar_slam_amd.synthetic import SceneConfig, make_fixed_map, make_sequence
    cfg = SceneConfig(height=64, width=2048)
    scans, poses = make_sequence(cfg, 9)
    model = make_fixed_map(cfg, scans[:8], poses[:8], ref_frame=7, num_points=100_000)
    return scans[8], model


def test_c2_full_size_registration_vs_reference_and_oracle(torch_cuda, O):
    """BASELINE.json configs[1] (the headline configuration): 64x2048 scan (131072 points) vs a 100k-point map, 20
    forced iterations, the three schemes of BASELINE.md.  Pinned on the REFERENCE's own run at this size
    (tests/golden/c2_reference.npz, oracle/make_golden_c2.py: `ICPFrameToModel.register_new_frame` on the same inputs):
    pose within 1e-4 m / 1e-4 rad, per-iteration loss and delta pose; also against the oracle, plus exactness of the
    search on a sample."""
    import hashlib
    import os
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "c2_reference.npz"))
    scan, model = _c2_inputs()
    assert hashlib.sha1(np.ascontiguousarray(scan).tobytes()).hexdigest() == str(g["scan_sha"]), \
        "the seeded generator no longer reproduces the scan the reference was run on"
    assert hashlib.sha1(np.ascontiguousarray(model).tobytes()).hexdigest() == str(g["model_sha"])
    ctx = _ctx(height=64, width=2048, max_num_alignments=20, threshold_delta_pose=0.0)
    ctx.map_set(model)
    dscan = torch_cuda.from_numpy(scan).cuda()
    # search exactness on a sample of the full-size problem
    sample = scan[::61]
    _, _, ix = ctx.nearest_neighbor_search(sample, with_normals=False, with_index=True)
    bi, bd2 = O.brute_force_nn(sample, model)
    d2 = ((sample.astype(np.float64) - model[ix].astype(np.float64)) ** 2).sum(axis=1)
    assert (ix == bi).mean() > 0.999
    np.testing.assert_allclose(d2, bd2, rtol=2e-6, atol=1e-12)
    for scheme, sigma in zip((str(v) for v in g["schemes"]), (float(v) for v in g["sigmas"])):
        ctx.set_alignment(scheme, sigma, int(g["iters"]), 0.0)
        ctx.map_set(model)  # clears the normal cache, like every map update
        res = ctx.register(dscan)
        assert res.iterations == 20 and res.num_targets == scan.shape[0]
        # ---- the reference itself
        dt, dr = O.pose_error(res.pose, g[f"{scheme}_pose"])
        print(f"C2 {scheme} vs REFERENCE: |dt| = {dt:.2e} m |dr| = {dr:.2e} rad, loss {res.losses[-1]:.4f} vs "
              f"{g[scheme + '_loss'][-1]:.4f}")
        assert dt < 1e-4 and dr < 1e-4, ("reference", scheme, dt, dr)
        np.testing.assert_allclose(res.params, g[f"{scheme}_params"], atol=1e-4)
        np.testing.assert_allclose(res.losses, g[f"{scheme}_loss"], rtol=2e-3)
        np.testing.assert_allclose(res.dx, g[f"{scheme}_dx"], atol=2e-5)
        # ---- the oracle (f64 accumulation like the device)
        lm = O.KdTreeLocalMapOracle()
        lm.set_map_pointcloud(model)
        oc = O.ICPOracleConfig(max_num_alignments=20, threshold_delta_pose=0.0, scheme=scheme, sigma=sigma,
                               height=64, width=2048, accumulate=np.float64)
        orc = O.ICPFrameToModelOracle(oc)
        orc.local_map = lm
        _, opose = orc.register_new_frame(scan, np.eye(4, dtype=np.float32))
        dt, dr = O.pose_error(res.pose, opose)
        print(f"C2 {scheme} vs oracle: |dt| = {dt:.2e} m |dr| = {dr:.2e} rad, normals computed {res.normals_computed}")
        assert dt < 1e-4 and dr < 1e-4, (scheme, dt, dr)
        np.testing.assert_allclose(res.losses[-1], orc.traces[-1].loss[-1], rtol=1e-3)


def _bench_workload():
    """bench.py's headline workload (`make_workload(0, "pingpong")`): tracked scans the map has never seen, the map the
    voxel-subsampled union of eight OTHER scans — restated here so that the test does not import the benchmark."""
    from pylidar_slam_amd.synthetic import SceneConfig, make_fixed_map, make_sequence
    cfg = SceneConfig(height=64, width=2048, seed=1234, step=0.2, yaw_rate=0.005)
    scans, poses = make_sequence(cfg, 16)
    even = list(range(0, 16, 2))
    model = make_fixed_map(cfg, [scans[f] for f in even], poses[even], ref_frame=0, num_points=100_000)
    rel = np.linalg.inv(poses[1]) @ poses[0]
    model = (model.astype(np.float64) @ rel[:3, :3].T + rel[:3, 3]).astype(np.float32)
    return {f: scans[f] for f in (3, 5, 7)}, poses, model


def test_schedule_options_at_benchmark_size(torch_cuda):
    """the schedule options proven bit-identical AT THE BENCHMARK'S SIZE (131 072 x 100 000 x 20: 1024
    workgroups in the 128-query shape — every slot of the chip — 257 in the lead launches, cells auto-tuned to 16 points,
    the mailbox / pose history / look-back under real dispatch pressure), over three chained frames of the bench's own
    workload (the second and third start from frame seeds and the constant-velocity guess); and the neighbour every
    target had in the LAST iteration, read back through the NN cache (icp_last_neighbors), is the true nearest map point
    (kd-tree on the host) for the whole scan — a cache that kept a second-nearest neighbour a few times per launch would
    pass a 1e-4 pose test."""
    from scipy.spatial import cKDTree
    scans, poses, model = _bench_workload()
    variants = {"default": {}, "nocache": {"nn_cache": 0}, "no_lead_solve": {"lead_solve": 0},
                "never_narrow": {"narrow_from": -1}, "narrow_always": {"narrow_from": 0}, "no_hoods": {"hoods": 0},
                "hoods_4_lanes": {"hoods": 1}, "narrow_from_3": {"narrow_from": 3}, "ball_max_64": {"ball_max": 64}, "ball_one_lane": {"ball_lanes": 1}, "ball_two_lanes": {"ball_lanes": 2},
                "never_wide": {"wide_until": 0}, "always_wide": {"wide_until": 99},
                "no_far_lanes": {"far_lanes": 0}, "far_lanes_many": {"far_max": 512},
                "far_lanes_small_balls": {"far_max": 512, "ball_max": 32},
                "no_flat_rows": {"flat_rows": 0}, "no_ball_search": {"ball_search": 0},
                "round3": {"ball_search": 0, "narrow_from": 3},
                # round 6: hit records — since the end of that round in builds of their own (REC_BUILD) — and the late kernel
                "hit_records": {"hit_records": 1}, "late_kernel": {"hit_records": 1, "late_from": 3},
                "hit_records_dense_then_narrow": {"hit_records": 1, "ball_search": 0, "narrow_from": 3},
                # round 4's schedule (a launch per iteration) / the resident tail from iteration 7
                "tail_from_3": {"resident_tail": 3}, "tail_from_7": {"resident_tail": 7},  # the resident tail (off by default)
                "no_carry": {"carry_normals": 0},
                # the stragglers of the eager normals on the map stream instead of inside the estimating launch: equal to `no_carry`
                "no_carry_tail_stream": {"carry_normals": 0, "normals_tail_stream": 1},
                "no_carry_listed_stragglers": {"carry_normals": 0, "normals_list": 1},  # (round 6: a launch of their own, 16 lanes each)
                "lazy_fused": {"lazy_fused": 2, "carry_normals": 0}}  # normals on demand inside the fused kernel: equal to `no_carry`
    results = {}
    for name, opts in variants.items():
        ctx = _ctx(height=64, width=2048, max_num_alignments=20, threshold_delta_pose=0.0, scheme="geman_mcclure",
                   sigma=0.3)
        for k, v in opts.items():
            ctx.set_option(k, v)
        ctx.map_set(model)
        frames, init = [], None
        for f in (3, 5):
            ctx.register_launch(scans[f], init)
            ctx.map_update(None, None)
            r = ctx.register_end()
            frames.append(r)
            init = r.pose
        r = ctx.register(scans[7], init)  # (no map update behind it: the cache still describes this registration)
        frames.append(r)
        ix, pose12 = ctx.last_neighbors(scans[7].shape[0])
        _, nrm, _ = ctx.nearest_neighbor_search(scans[7][::13])
        results[name] = (frames, ix, pose12, nrm, ctx.map_points())
        assert ctx.handoff_fallbacks() == 0, name
        ctx.close()
    problems = []
    for name, (frames, ix, pose12, nrm, mp) in results.items():
        ref = results["no_carry" if name in ("lazy_fused", "no_carry_tail_stream", "no_carry_listed_stragglers") else "default"]
        if name == "no_carry":  # (the reference's schedule against the carried one: rounding apart)
            for r, rr in zip(frames, results["default"][0]):
                np.testing.assert_allclose(r.pose, rr.pose, atol=1e-6)
            continue
        for f, (r, rr) in enumerate(zip(frames, ref[0])):
            assert r.iterations == 20
            if not (np.array_equal(r.pose, rr.pose) and np.array_equal(r.losses, rr.losses) and np.array_equal(r.dx, rr.dx)):
                first = int(np.argmax(r.losses != rr.losses)) if (r.losses != rr.losses).any() else -1
                problems.append(f"{name}: frame {f} differs (first loss mismatch at iteration {first}, "
                                f"max |dpose| {np.abs(r.pose - rr.pose).max():.1e})")
        if not np.array_equal(pose12, ref[2]):
            problems.append(f"{name}: pose of the last iteration differs")
        if not np.array_equal(ix, ref[1]):  # (with the cache off every launch searches and writes its entries all the same)
            problems.append(f"{name}: {int((ix != ref[1]).sum())} neighbours of the last iteration differ")
        if not np.array_equal(nrm, ref[3]):
            problems.append(f"{name}: normals differ")
        if not np.array_equal(mp, ref[4]):
            problems.append(f"{name}: map differs")
    assert not problems, "\n".join(problems)
    ref = results["default"]
    # ---- the neighbours of the last iteration against the kd-tree, every target.  The device transforms in float32:
    # p = fma(z, T2, fma(y, T1, x * T0)) + T3 (search_device.h::transform_point), restated here (products of two floats
    # are exact in float64)
    frames, ix, pose12, _, mp = ref
    t = pose12.astype(np.float32)
    s7 = scans[7].astype(np.float32)
    def row(k):
        acc = (s7[:, 0] * t[k, 0]).astype(np.float32).astype(np.float64)
        acc = (s7[:, 1].astype(np.float64) * np.float64(t[k, 1]) + acc).astype(np.float32).astype(np.float64)
        acc = (s7[:, 2].astype(np.float64) * np.float64(t[k, 2]) + acc).astype(np.float32)
        return (acc + t[k, 3]).astype(np.float32)
    p = np.stack([row(0), row(1), row(2)], axis=1).astype(np.float64)
    cur = mp.astype(np.float64)  # the map the last registration ran against (insertion order = original indices)
    assert (ix >= 0).all() and ix.max() < cur.shape[0]
    bd, bi = cKDTree(cur).query(p)
    d_used = np.linalg.norm(p - cur[ix], axis=1)
    same = (ix == bi).mean()
    print(f"last-iteration neighbours: {same * 100:.4f} % equal to the kd-tree's, max excess distance "
          f"{(d_used - bd).max():.2e} m")
    assert same > 0.9995
    # where the index differs the two candidates are equidistant to float32 resolution of the squared distance
    np.testing.assert_allclose(d_used ** 2, bd ** 2, rtol=4e-6, atol=1e-12)


def test_targets_far_from_the_map_in_the_fused_kernel(torch_cuda):
    """No distance cap (local_map.py:385-386) INSIDE the fused iteration kernel: a sparse frame (a few thousand targets over
    131 072 pixels, a handful of cache misses per workgroup — the published configuration's shape) in which some targets sit
    1-6 m away from every map point (own cell empty, fine rings 1-2 empty: the coarse level, searched by a whole wave since
    round 5) and a few hundreds of metres away (the exhaustive scan).  The neighbour of every target after the last
    iteration (icp_last_neighbors) is the kd-tree's; and the schedules that send those queries down other paths — 4-lane
    groups only, 16 lanes, no ball search — give the same registration bit for bit."""
    from scipy.spatial import cKDTree
    scans, poses, model = _bench_workload()
    rng = np.random.default_rng(3)
    frame = np.full((131072, 3), np.nan, np.float32)
    keep = rng.choice(131072, 6000, replace=False)
    frame[keep] = scans[3][keep]
    far = rng.choice(keep, 60, replace=False)
    frame[far[:40], 2] += rng.uniform(1.5, 6.0, 40).astype(np.float32)       # above the scene: beyond the fine rings
    frame[far[40:52]] += np.float32(40.0)                                    # beyond a few coarse rings
    frame[far[52:]] = (rng.normal(size=(8, 3)) * 50 + 400).astype(np.float32)  # beyond every ring: exhaustive
    variants = {"default": {}, "four_lanes_only": {"wave_misses": 0, "far_lanes": 0}, "no_ball_search": {"ball_search": 0},
                "never_wide": {"wide_until": 0}, "far_lanes_from_1": {"far_min": 0}}
    results = {}
    for name, opts in variants.items():
        ctx = _ctx(height=64, width=2048, max_num_alignments=4, threshold_delta_pose=0.0, scheme="geman_mcclure", sigma=0.3)
        for k, v in opts.items():
            ctx.set_option(k, v)
        ctx.map_set(model)
        r = ctx.register(frame)
        ix, pose12 = ctx.last_neighbors(frame.shape[0])
        results[name] = (r, ix, pose12, ctx.map_points())
        assert ctx.handoff_fallbacks() == 0, name
        ctx.close()
    ref = results["default"]
    for name, (r, ix, pose12, _) in results.items():
        assert r.iterations == 4
        np.testing.assert_array_equal(r.pose, ref[0].pose, err_msg=name)
        np.testing.assert_array_equal(r.losses, ref[0].losses, err_msg=name)
        np.testing.assert_array_equal(ix, ref[1], err_msg=name)
    r, ix, pose12, mp = ref
    valid = ~np.isnan(frame).any(axis=1)
    assert (ix[valid] >= 0).all() and (ix[~valid] < 0).all()
    t = pose12.astype(np.float64)
    p = frame[valid].astype(np.float64) @ t[:, :3].T + t[:, 3]
    cur = mp.astype(np.float64)
    bd, bi = cKDTree(cur).query(p)
    d_used = np.linalg.norm(p - cur[ix[valid]], axis=1)
    assert (ix[valid] == bi).mean() > 0.999
    np.testing.assert_allclose(d_used, bd, rtol=1e-5, atol=1e-5)  # (the device transforms in float32)
    assert (bd > 1.0).sum() >= 50  # the far targets really are far


def test_c2_full_size_properties(torch_cuda, O):
    """Size-independent properties at the headline size (131072-point scan, 100k-point map, 20 iterations), where the
    oracle is too slow to be run case by case:
      * order invariance — shuffling the scan rows or the map rows changes which workgroup sums what, not the problem:
        same pose to float32 reduction noise;
      * rigid equivariance — map and initial guess moved by the same rigid G: the result moves by G;
      * fixed point — restarting from the converged pose moves it by less than the noise floor of the last iterations;
      * every reported neighbour is a true nearest neighbour (brute force over the whole map on a sample)."""
    scan, model = _c2_inputs()
    rng = np.random.default_rng(11)

    def register(points, map_points, init=None, iters=20):
        ctx = _ctx(height=64, width=2048, max_num_alignments=iters, threshold_delta_pose=0.0, scheme="geman_mcclure",
                   sigma=0.3)
        ctx.map_set(map_points)
        r = ctx.register(points, init)
        ctx.close()
        return r

    base = register(scan, model)
    assert base.iterations == 20
    # ---- order invariance
    r_scan = register(scan[rng.permutation(scan.shape[0])], model)
    r_map = register(scan, model[rng.permutation(model.shape[0])])
    for name, r in (("scan order", r_scan), ("map order", r_map)):
        dt, dr = O.pose_error(r.pose, base.pose)
        assert dt < 2e-6 and dr < 2e-6, (name, dt, dr)
        np.testing.assert_allclose(r.losses[-1], base.losses[-1], rtol=1e-5)
    # ---- rigid equivariance
    G = O.build_pose_matrix(np.array([1.5, -0.7, 0.2, 0.02, -0.03, 0.4], np.float32)).astype(np.float64)
    moved = (model.astype(np.float64) @ G[:3, :3].T + G[:3, 3]).astype(np.float32)
    r_g = register(scan, moved, init=G.astype(np.float32))
    expect = (G @ base.pose.astype(np.float64)).astype(np.float32)
    dt, dr = O.pose_error(r_g.pose, expect)
    assert dt < 1e-4 and dr < 1e-4, ("equivariance", dt, dr)  # float32 re-expression of the map: ~1e-6 per coordinate
    # ---- fixed point
    again = register(scan, model, init=base.pose, iters=3)
    dt, dr = O.pose_error(again.pose, base.pose)
    assert dt < 2e-5 and dr < 2e-5, ("fixed point", dt, dr)
    # ---- exactness of the neighbours at the converged pose
    ctx = _ctx(height=64, width=2048)
    ctx.map_set(model)
    sample = (scan[::97].astype(np.float64) @ base.pose[:3, :3].astype(np.float64).T + base.pose[:3, 3]).astype(np.float32)
    _, _, ix = ctx.nearest_neighbor_search(sample, with_normals=False, with_index=True)
    _, bd2 = O.brute_force_nn(sample, model)
    d2 = ((sample.astype(np.float64) - model[ix].astype(np.float64)) ** 2).sum(axis=1)
    np.testing.assert_allclose(d2, bd2, rtol=2e-6, atol=1e-12)
    ctx.close()


# ---- projective local map ---------------------------------------------------------------------
@pytest.fixture(scope="module")
def golden_projective():
    import os
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "projective.npz"))


def test_projective_components(torch_cuda, O, golden_projective):
    g = golden_projective
    h, w = (int(v) for v in g["hw"])
    vm = g["vmaps"]
    ctx = _ctx(height=h, width=w)
    # compute_neighbors: exact vs the reference's own output (and the reference's property test)
    nb, nf = ctx.compute_neighbors(vm[1], np.stack([vm[0], vm[2], vm[3]]), g["cn_fields"])
    np.testing.assert_array_equal(nb, g["cn_neighbors"])
    np.testing.assert_array_equal(nf, g["cn_neighbor_fields"])
    rng = np.random.default_rng(1)
    tgt = rng.normal(size=(3, h, w)).astype(np.float32)
    ref = rng.normal(size=(10, 3, h, w)).astype(np.float32)
    tgt[:, 0, 0] = 0.0
    nb2, _ = ctx.compute_neighbors(tgt, ref)
    assert np.linalg.norm(nb2[:, 0, 0]) == 0.0
    np.testing.assert_array_equal(nb2, O.compute_neighbors(tgt, ref)[0])
    # normal map: float64 window sums on the device = the exact value (tight), which the reference's float32 result
    # approximates with a median ~3e-4 rad / p99 ~5e-3 rad error (loose)
    nm = ctx.compute_normal_map(vm[0], 5)
    exact = O.compute_normal_map(vm[0], 5, dtype=np.float64)
    both = (np.abs(nm).max(axis=0) > 0) & (np.abs(exact).max(axis=0) > 0)
    assert ((np.abs(nm).max(axis=0) > 0) == (np.abs(exact).max(axis=0) > 0)).mean() > 0.999
    ang = np.linalg.norm(np.cross(nm, exact, axis=0), axis=0)[both]  # sin(angle): well conditioned near 0
    assert ang.max() < 1e-4 and np.median(ang) < 1e-6, (ang.max(), np.median(ang))
    ref_n = g["nmap0"]
    v2 = both & (np.abs(ref_n).max(axis=0) > 0)
    ang_ref = np.linalg.norm(np.cross(nm, ref_n, axis=0), axis=0)[v2]
    assert np.median(ang_ref) < 1e-3 and np.percentile(ang_ref, 99) < 2e-2
    dn = ctx.compute_normal_map(torch_cuda.from_numpy(vm[0]).cuda(), 5)
    assert dn.is_cuda and np.array_equal(dn.cpu().numpy(), nm)


def test_projective_map_model_and_search(torch_cuda, O, golden_projective):
    g = golden_projective
    h, w = (int(v) for v in g["hw"])
    vm = g["vmaps"]
    ctx = _ctx(height=h, width=w, local_map_size=2)
    orc = O.ProjectiveLocalMapOracle(h, w, 3.0, -24.0, local_map_size=2, normals_dtype=np.float64)
    rel = O.build_pose_matrix(np.array([0.4, 0.01, -0.01, 0.002, -0.001, 0.01], np.float32))
    ctx.pmap_init()
    for k, (pose, v) in enumerate([(np.eye(4, dtype=np.float32), vm[0]), (rel, vm[1]), (rel, None), (rel, vm[2])]):
        ctx.pmap_update(pose, v)
        orc.update(pose, v)
        assert ctx.pmap_num_maps() == len(orc.vmaps)
    mv, mn = ctx.pmap_model()
    # the model maps: same pixels occupied, same points (f32 transform rounding), normals within the f64-vs-f32 rounding
    occ = np.abs(mv).max(axis=1) > 0
    assert (occ == (np.abs(orc.model_vmap).max(axis=1) > 0)).mean() > 0.9995
    same = occ & (np.abs(orc.model_vmap).max(axis=1) > 0)
    dv = np.abs(mv - orc.model_vmap).max(axis=1)[same]
    assert np.percentile(dv, 99.9) < 1e-4, np.percentile(dv, 99.9)  # a few pixels pick another z-buffer winner
    # association of a transformed scan
    pts = O.apply_transformation(O.vertex_map_to_points(vm[3]), rel)
    pts = pts[np.abs(pts).max(axis=1) > 0]
    nb, nm, tg = ctx.pmap_nearest_neighbor_search(pts)
    onb, onm, otg = orc.nearest_neighbor_search(pts)
    assert abs(nb.shape[0] - onb.shape[0]) <= max(3, onb.shape[0] // 2000)
    if nb.shape[0] == onb.shape[0]:
        close = np.abs(nb - onb).max(axis=1) < 1e-4
        assert close.mean() > 0.998
        np.testing.assert_array_equal(tg, otg)


@pytest.mark.parametrize("run", ["ls", "nbh"])
def test_projective_icp_sequence(torch_cuda, O, golden_projective, run):
    """Row a19 end to end behind the plugin surface: `local_map.type = projective_local_map`, vertex-map input, the
    configuration (scheme, iteration cap, stop threshold) the reference was run with.  Per frame within 1e-4 m /
    1e-4 rad of the REFERENCE's own run (tests/golden/projective.npz) and of the reference with its two box-filter
    convolutions carried out in float64 (tests/golden/projective_spread.npz, oracle/make_golden_projective_spread.py),
    and of the oracle with float64 window sums.  The reference's float32 box filter is bistable at this precision: its
    own runs on a mirrored image land 1.9e-3 m away on the `ls` sequence (same fixture) — the HIP path sits with the
    baseline."""
    import os
    from conftest import GOLDEN
    from pylidar_slam_amd.odometry import MI355XICPConfig, MI355XICPFrameToModel, SphericalProjector
    g = golden_projective
    sp = np.load(os.path.join(GOLDEN, "projective_spread.npz"))
    h, w = (int(v) for v in g["hw"])
    scheme, sigma, iters, thr = (str(v) for v in g[f"{run}_cfg"])
    cfg = MI355XICPConfig(max_num_alignments=int(iters), threshold_delta_pose=float(thr), data_key="vertex_map",
                          local_map=dict(type="projective_local_map", local_map_size=4),
                          alignment=dict(mode="point_to_plane_gauss_newton",
                                         gauss_newton_config=dict(max_iters=1, scheme=scheme, sigma=float(sigma))))
    odo = MI355XICPFrameToModel(cfg, projector=SphericalProjector(h, w), device=torch_cuda.device("cuda:0"))
    odo.init()
    oc = O.ICPOracleConfig(max_num_alignments=int(iters), threshold_delta_pose=float(thr), scheme=scheme,
                           sigma=float(sigma), height=h, width=w, local_map_size=4, accumulate=np.float64)
    orc = O.ICPProjectiveOracle(oc, normals_dtype=np.float64)
    last = None
    worst = {"reference": (0.0, 0.0), "reference_f64conv": (0.0, 0.0), "oracle": (0.0, 0.0)}
    for f, vm in enumerate(g["vmaps"]):
        d = {"vertex_map": torch_cuda.from_numpy(vm), "init_rpose": last}
        odo.process_next_frame(d)
        opose = orc.process_next_frame(vm, last)
        if f == 0:
            continue
        pose = d["odometry_pose"]
        last = pose.astype(np.float64)
        assert odo.last_result.iterations == int(g[f"{run}_iters"][f]), (run, f, odo.last_result.iterations)
        for name, ref in (("reference", g[f"{run}_rel"][f]), ("reference_f64conv", sp[f"{run}_float64_rel"][f]),
                          ("oracle", opose)):
            dt, dr = O.pose_error(pose, ref)
            worst[name] = (max(worst[name][0], dt), max(worst[name][1], dr))
            assert dt < 1e-4 and dr < 1e-4, (name, run, f, dt, dr)
    print(f"projective {run}: " + ", ".join(f"{k} {v[0]:.1e} m / {v[1]:.1e} rad" for k, v in worst.items()))


# ---- point-to-point alignment + weighted Procrustes --------------------------------------------
@pytest.mark.parametrize("name", ["ls", "huber", "nbh", "gm_svd", "ls_svd"])
def test_point_to_point_alignment(torch_cuda, O, name):
    import os
    from conftest import GOLDEN
    from pylidar_slam_amd.odometry import PointToPointAlignment
    g = np.load(os.path.join(GOLDEN, "alignment.npz"))
    scheme, sigma, svd = g[f"{name}_cfg"]
    ctx = _ctx(scheme=str(scheme), sigma=float(sigma))
    algo = PointToPointAlignment(ctx, initialize_with_svd=bool(int(svd)))
    pose, params, residuals = algo.align(g["ref"], g["tgt"])
    assert residuals.shape == (1, g["ref"].shape[0])  # the residual vector (w r)^2 per row, like the reference
    loss = float(residuals.astype(np.float64).sum())
    # vs the reference's own float32 result (its normal equations are float32, ours float64)
    np.testing.assert_allclose(params[0], g[f"{name}_params"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(pose[0], g[f"{name}_pose"], atol=5e-5)
    np.testing.assert_allclose(loss, float(g[f"{name}_loss"]), rtol=1e-4)
    # vs the oracle accumulating in float64 like the device: tight
    x0 = O.from_pose_matrix(O.weighted_procrustes(g["ref"], g["tgt"]).astype(np.float32)) if int(svd) else None
    _, p64, l64 = O.point_to_point_step(g["tgt"], g["ref"], x0, str(scheme), float(sigma), accumulate=np.float64)
    np.testing.assert_allclose(params[0], p64, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(loss, l64, rtol=1e-5)
    # device-resident inputs give the same bits
    r_t, t_t = torch_cuda.from_numpy(g["ref"]).cuda(), torch_cuda.from_numpy(g["tgt"]).cuda()
    pose_d, params_d, res_d = algo.align(r_t[None], t_t[None])
    assert params_d.is_cuda and res_d.is_cuda  # results live where the inputs live
    assert np.array_equal(params_d[0].cpu().numpy(), params[0])
    assert np.array_equal(res_d.cpu().numpy(), residuals)


def test_weighted_procrustes(torch_cuda, O):
    import os
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "alignment.npz"))
    ctx = _ctx()
    np.testing.assert_allclose(ctx.weighted_procrustes(g["tgt"], g["ref"]), g["procrustes_np"], atol=2e-6)
    np.testing.assert_allclose(ctx.weighted_procrustes(g["tgt"], g["ref"], g["weights"]), g["procrustes_np_weighted"],
                               atol=2e-6)
    np.testing.assert_allclose(ctx.weighted_procrustes(g["flat_tgt"], g["flat_ref"]), g["procrustes_flat"], atol=2e-6)
    T = ctx.weighted_procrustes(g["tgt"], g["ref"])
    assert abs(np.linalg.det(T[:3, :3]) - 1.0) < 1e-12
    # degenerate inputs: all points equal (rank 0) -> a proper rotation and the centroid shift; mirrored cloud -> still
    # a rotation (the reflection fix), never a reflection
    same = np.tile(np.array([[1.0, 2.0, 3.0]], np.float32), (50, 1))
    T0 = ctx.weighted_procrustes(same, same + np.float32(0.5))
    assert abs(np.linalg.det(T0[:3, :3]) - 1.0) < 1e-9
    np.testing.assert_allclose(T0[:3, :3] @ same[0] + T0[:3, 3], same[0] + 0.5, atol=1e-5)
    mirrored = g["ref"] * np.array([1, 1, -1], np.float32)
    Tm = ctx.weighted_procrustes(mirrored, g["ref"])
    assert abs(np.linalg.det(Tm[:3, :3]) - 1.0) < 1e-9
    np.testing.assert_allclose(Tm, O.weighted_procrustes(mirrored, g["ref"]), atol=1e-5)
    with pytest.raises(AssertionError):
        ctx.weighted_procrustes(g["tgt"], g["ref"], np.zeros(g["ref"].shape[0], np.float32))


# ---- Voxelization statistics --------------------------------------------------------------------
@pytest.mark.parametrize("name", ["v02", "v10"])
def test_voxelization_filter(torch_cuda, O, name):
    import os
    from conftest import GOLDEN
    from pylidar_slam_amd.odometry import Voxelization, VoxelizationConfig
    g = np.load(os.path.join(GOLDEN, "voxelization.npz"))
    d = {"numpy_pc": g["pc"]}
    Voxelization(VoxelizationConfig(voxel_size=float(g[f"{name}_size"]))).filter(d)
    # integer work bit-exact; the float32 sums follow the sorted order (the reference's own order inside a voxel is
    # unspecified: its argsort is not stable), so they agree to rounding
    for k in ("voxel_hashes", "voxel_coordinates", "voxel_sizes", "voxel_indices"):
        assert d[k].dtype == np.int64
        np.testing.assert_array_equal(d[k], g[f"{name}_{k}"])
    assert d["voxel_means"].dtype == np.float32 and d["voxel_covariances"].shape == g[f"{name}_voxel_covariances"].shape
    np.testing.assert_allclose(d["voxel_means"], g[f"{name}_voxel_means"], rtol=1e-6, atol=1e-5)
    np.testing.assert_allclose(d["voxel_covariances"], g[f"{name}_voxel_covariances"], rtol=1e-4, atol=1e-5)
    # and bit-exact against the oracle, which sums in the same (stable) order
    sizes, means, covs, ids = O.voxel_normal_distribution(g["pc"], d["voxel_hashes"])
    np.testing.assert_allclose(d["voxel_means"], means, rtol=0, atol=2e-6)
    # without statistics: only coordinates and hashes are written (preprocessing.py:89-92)
    d2 = {"numpy_pc": g["pc"]}
    Voxelization(VoxelizationConfig(voxel_size=0.2, with_normal_distribution=False)).filter(d2)
    assert set(d2) == {"numpy_pc", "voxel_hashes", "voxel_coordinates"}
    # edge cases: empty cloud, a single point, all points in one voxel
    ctx = _ctx()
    e = ctx.voxel_statistics(np.zeros((0, 3), np.float32), 0.2)
    assert e["num_voxels"] == 0 and e["voxel_sizes"].shape == (0,)
    one = ctx.voxel_statistics(np.array([[1.0, 2.0, 3.0]], np.float32), 0.2)
    assert one["num_voxels"] == 1 and one["voxel_sizes"][0] == 1 and np.all(one["voxel_covariances"] == 0)
    blob = np.random.default_rng(0).uniform(-0.05, 0.05, (500, 3)).astype(np.float32)
    b = ctx.voxel_statistics(blob, 1.0)
    assert b["num_voxels"] == 1 and b["voxel_sizes"][0] == 500 and np.all(b["voxel_indices"] == 0)
    np.testing.assert_allclose(b["voxel_means"][0], blob.mean(axis=0), atol=1e-6)
