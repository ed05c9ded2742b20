"""GPU tests of the multi-GPU paths on the ONE GPU of the test box: the map-sharded normal estimation (ownership by
spatial bucket, exchange by original map index), the C4-sized problem (200k-point scan vs 1M-point map), and the HIP
engine under world size 2 — two processes sharing the GPU over gloo, the same `sharded_register` / `sharded_map_normals`
drivers that run over RCCL on a multi-GPU node."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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


def _small_problem():
    from pylidar_slam_amd.synthetic import SceneConfig, make_fixed_map, make_sequence
    cfg = SceneConfig(height=32, width=1024)
    scans, poses = make_sequence(cfg, 5)
    model = make_fixed_map(cfg, scans[:4], poses[:4], ref_frame=3, num_points=30_000)
    return model, scans[4]


def test_map_sharded_normals_equal_the_single_gpu_estimation(torch_cuda):
    """Two "ranks" (two contexts holding the same map) estimate the normals of their own spatial buckets; the arrays,
    summed by original index, installed on both, give bit for bit the normals — hence the registration — of the
    single-GPU eager estimation."""
    torch = torch_cuda
    model, scan = _small_problem()
    kw = dict(height=32, width=1024, max_num_alignments=10, threshold_delta_pose=0.0, scheme="geman_mcclure", sigma=0.3)
    a, b, single = _ctx(**kw), _ctx(**kw), _ctx(**kw)
    for c in (a, b, single):
        c.map_set(model)
    ta, tb = a.map_normals_owned(0, 2), b.map_normals_owned(1, 2)
    owned_a, owned_b = ta[:, 3].cpu().numpy(), tb[:, 3].cpu().numpy()
    assert set(np.unique(owned_a)) <= {0.0, 1.0} and np.array_equal(owned_a + owned_b, np.ones(model.shape[0], np.float32))
    assert 0.25 < owned_a.mean() < 0.75  # the spatial hash splits the map roughly evenly
    assert (ta[owned_a == 0].abs().sum() == 0) and (tb[owned_b == 0].abs().sum() == 0)
    total = ta + tb  # what the all-reduce computes
    np.testing.assert_allclose(total[:, :3].norm(dim=1).cpu().numpy(), 1.0, atol=1e-5)
    a.map_normals_install(total)
    b.map_normals_install(total)
    ra, rb, rs = a.register(scan), b.register(scan), single.register(scan)
    for r in (ra, rb):
        assert np.array_equal(r.pose, rs.pose) and np.array_equal(r.losses, rs.losses) and np.array_equal(r.dx, rs.dx)
    assert ra.normals_computed == rs.normals_computed == model.shape[0]
    # world size 1 through the driver = the eager estimation
    from pylidar_slam_amd.distributed import sharded_map_normals
    c = _ctx(**kw)
    c.map_set(model)
    shard = sharded_map_normals(c)
    assert float(shard[:, 3].min()) == 1.0
    assert np.array_equal(c.register(scan).pose, rs.pose)
    with pytest.raises(AssertionError):
        c.map_normals_install(total[:-1])


def test_c4_size_map_sharded_registration_vs_oracle(torch_cuda, O):
    """BASELINE.json configs[3] on one GPU: a 128-beam 200k-point scan against a 1M-point map.  The map is > 2x the scan,
    so below `eager_normals_limit` normals are estimated lazily for the map points the scan touches (unfused iterations); with the
    map-sharded estimation (here two simulated ranks) the normals are all there and the fused kernel runs.  Both within
    1e-4 m / 1e-4 rad of the oracle over the benchmark's 20 iterations, every iteration's loss within 1e-3."""
    torch = torch_cuda
    from pylidar_slam_amd.synthetic import SceneConfig, make_fixed_map, make_sequence
    cfg = SceneConfig(height=128, width=1563, up_fov=22.5, down_fov=-22.5, step=0.2, yaw_rate=0.005)
    scans, poses = make_sequence(cfg, 21)
    model = make_fixed_map(cfg, scans[:20], poses[:20], ref_frame=19, num_points=1_000_000, voxel=0.1)
    scan = scans[20]
    assert scan.shape[0] == 128 * 1563 and model.shape[0] == 1_000_000
    iters = 20  # (the benchmark's iteration count: VERDICT r3 item 3; five in round 3)
    kw = dict(height=128, width=1563, up_fov=22.5, down_fov=-22.5, max_num_alignments=iters, threshold_delta_pose=0.0,
              scheme="geman_mcclure", sigma=0.3)
    dmodel, dscan = torch.from_numpy(model).cuda(), torch.from_numpy(scan).cuda()
    lazy = _ctx(**kw)
    lazy.set_option("eager_normals_limit", 0)  # (a map of up to 2^20 points is otherwise estimated at once: faster)
    lazy.map_set(dmodel)
    r_lazy = lazy.register(dscan)
    assert r_lazy.iterations == iters and 0 < r_lazy.normals_computed < model.shape[0]  # on demand, like the reference
    ranks = [_ctx(**kw), _ctx(**kw)]
    for c in ranks:
        c.map_set(dmodel)
    total = ranks[0].map_normals_owned(0, 2) + ranks[1].map_normals_owned(1, 2)
    assert float(total[:, 3].min()) == 1.0 and float(total[:, 3].max()) == 1.0
    for c in ranks:
        c.map_normals_install(total)
    r0, r1 = ranks[0].register(dscan), ranks[1].register(dscan)
    assert np.array_equal(r0.pose, r1.pose) and r0.normals_computed == model.shape[0]
    lm = O.KdTreeLocalMapOracle()
    lm.set_map_pointcloud(model)
    orc = O.ICPFrameToModelOracle(O.ICPOracleConfig(max_num_alignments=iters, threshold_delta_pose=0.0,
                                                    scheme="geman_mcclure", sigma=0.3, height=128, width=1563,
                                                    up_fov=22.5, down_fov=-22.5, accumulate=np.float64))
    orc.local_map = lm
    _, opose = orc.register_new_frame(scan, np.eye(4, dtype=np.float32))
    for name, r in (("lazy", r_lazy), ("map-sharded", r0)):
        dt, dr = O.pose_error(r.pose, opose)
        print(f"C4 {name}: |dt| = {dt:.2e} m |dr| = {dr:.2e} rad vs oracle, loss {r.losses[-1]:.3f} vs "
              f"{orc.traces[-1].loss[-1]:.3f}")
        assert dt < 1e-4 and dr < 1e-4, (name, dt, dr)
        np.testing.assert_allclose(r.losses, orc.traces[-1].loss, rtol=1e-3)


# ---- world size 2 on one GPU (gloo) ----------------------------------------------------------------------------------
def _rank_main(rank, world, port, out, backend="gloo"):
    for p in (os.path.join(ROOT, "pylidar-slam_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    device = rank if backend == "nccl" else 0  # nccl (= RCCL): one GPU per rank; gloo: the ranks share GPU 0
    torch.cuda.set_device(device)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from pylidar_slam_amd.distributed import shard_bounds, sharded_map_normals, sharded_register
    from pylidar_slam_amd.engine import IcpContext
    model, scan = _small_problem()
    ctx = IcpContext(height=32, width=1024, max_num_alignments=10, threshold_delta_pose=0.0, scheme="geman_mcclure",
                     sigma=0.3, device=device)
    ctx.map_set(torch.from_numpy(model).cuda())
    shard = sharded_map_normals(ctx)  # owner-computed normals, all-reduced by original index
    b, e = shard_bounds(scan.shape[0], world, rank)
    res = sharded_register(ctx, torch.from_numpy(scan[b:e]).cuda(), None, 10)
    where = "cuda" if backend == "nccl" else "cpu"  # (RCCL moves device tensors only)
    poses = [torch.zeros(16, device=where) for _ in range(world)]
    dist.all_gather(poses, torch.from_numpy(res.pose.reshape(-1).copy()).to(where))
    if rank == 0:
        np.savez(out, pose=res.pose, losses=res.losses, dx=res.dx, all=np.stack([p.cpu().numpy() for p in poses]),
                 owned_everywhere=float(shard[:, 3].min()), targets=res.num_targets)
    dist.barrier()
    dist.destroy_process_group()


def test_hip_engine_world_size_2_on_one_gpu(torch_cuda, tmp_path):
    """Two processes, one GPU, gloo: each rank registers its slice of the scan against the replicated map with the HIP
    engine — map normals estimated by bucket owner and all-reduced, per iteration the packed normal equations
    all-reduced, the identical solve on both ranks.  Both ranks end with the same bits; against the single-process
    registration of the whole scan the only difference is the association of the float64 partial sums (per-rank sums
    added by the all-reduce instead of one fixed-order sum over all blocks), i.e. at most an ulp of the float32 step."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "rank0.npz")
    mp.start_processes(_rank_main, args=(2, port, out), nprocs=2, join=True, start_method="spawn")
    got = np.load(out)
    assert np.array_equal(got["all"][0], got["all"][1])  # bit-identical across ranks
    assert float(got["owned_everywhere"]) == 1.0
    model, scan = _small_problem()
    single = _ctx(height=32, width=1024, max_num_alignments=10, threshold_delta_pose=0.0, scheme="geman_mcclure",
                  sigma=0.3)
    single.map_set(model)
    ref = single.register(scan)
    assert len(got["losses"]) == ref.iterations == 10
    np.testing.assert_allclose(got["pose"], ref.pose, atol=1e-6)
    np.testing.assert_allclose(got["losses"], ref.losses, rtol=1e-9)
    np.testing.assert_allclose(got["dx"], ref.dx, atol=1e-7)


def _check_against_single_process(got, bitwise):
    model, scan = _small_problem()
    single = _ctx(height=32, width=1024, max_num_alignments=10, threshold_delta_pose=0.0, scheme="geman_mcclure",
                  sigma=0.3)
    single.map_set(model)
    ref = single.register(scan)
    assert float(got["owned_everywhere"]) == 1.0 and int(got["targets"]) > 0
    assert len(got["losses"]) == ref.iterations == 10
    if bitwise:
        assert np.array_equal(got["pose"], ref.pose) and np.array_equal(got["losses"], ref.losses)
    else:
        np.testing.assert_allclose(got["pose"], ref.pose, atol=1e-6)
        np.testing.assert_allclose(got["losses"], ref.losses, rtol=1e-9)
        np.testing.assert_allclose(got["dx"], ref.dx, atol=1e-7)


def test_rccl_is_loaded_and_called_world_size_1(torch_cuda, tmp_path):
    """The `nccl` backend (RCCL on ROCm) on the one GPU of the test box: a process group of one rank, through which
    `sharded_map_normals` (one all-reduce of [M,4] floats) and `sharded_register` (an all-reduce of the 32 doubles per
    ICP iteration, between icp_iteration_accumulate and icp_iteration_solve on torch's stream) really issue their
    collectives.  A one-rank sum changes no bit: the result equals the plain single-process registration exactly."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "rank0.npz")
    mp.start_processes(_rank_main, args=(1, port, out, "nccl"), nprocs=1, join=True, start_method="spawn")
    _check_against_single_process(np.load(out), bitwise=True)


def test_rccl_scan_sharded_registration_on_two_gpus(torch_cuda, tmp_path):
    """Two ranks on two GPUs over RCCL / xGMI — runs wherever the box has at least two devices (the single-GPU test box
    skips it): same bits on both ranks, the single-process result up to the association of the float64 sums."""
    if torch_cuda.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "rank0.npz")
    mp.start_processes(_rank_main, args=(2, port, out, "nccl"), nprocs=2, join=True, start_method="spawn")
    got = np.load(out)
    assert np.array_equal(got["all"][0], got["all"][1])
    _check_against_single_process(got, bitwise=False)


# ---- in-library exchange (peer-written inboxes instead of a collective) ---------------------------------------------
def _rank_main_exchange(rank, world, port, out):
    for p in (os.path.join(ROOT, "pylidar-slam_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pylidar_slam_amd.distributed import connect_exchange, shard_bounds, sharded_map_normals
    from pylidar_slam_amd.engine import ExchangeTimeoutError, IcpContext
    model, scan = _small_problem()
    ctx = IcpContext(height=32, width=1024, max_num_alignments=10, threshold_delta_pose=0.0, scheme="geman_mcclure",
                     sigma=0.3)
    ctx.set_option("exchange_timeout_ms", 4000)
    ctx.map_set(torch.from_numpy(model).cuda())
    sharded_map_normals(ctx)
    assert connect_exchange(ctx)
    b, e = shard_bounds(scan.shape[0], world, rank)
    mine = torch.from_numpy(scan[b:e]).cuda()
    res = ctx.register(mine)            # ONE call: every iteration and its exchange enqueued by the library
    res2 = ctx.register(mine, res.pose)  # a second registration: the exchange numbering carries on
    poses = [torch.zeros(16) for _ in range(world)]
    dist.all_gather(poses, torch.from_numpy(res2.pose.reshape(-1).copy()))
    timed_out = False
    dist.barrier()
    if rank == 0:  # the peer does not take part in a third registration: bounded wait, clean error, no hang
        ctx.set_option("exchange_timeout_ms", 300)
        try:
            ctx.register(mine)
        except ExchangeTimeoutError:
            timed_out = True
        # the ranks' exchange counters may have diverged: the context left exchange mode and solves rank-locally until
        # the exchange is created and connected again (no stale inbox payload is ever summed)
        alone = ctx.register(mine)
        solo = IcpContext(height=32, width=1024, max_num_alignments=10, threshold_delta_pose=0.0,
                          scheme="geman_mcclure", sigma=0.3)
        solo.map_set(torch.from_numpy(model).cuda())
        timed_out = timed_out and np.array_equal(alone.pose, solo.register(mine).pose)
        np.savez(out, pose=res.pose, losses=res.losses, dx=res.dx, pose2=res2.pose,
                 all=np.stack([p.numpy() for p in poses]), targets=res.num_targets, timed_out=timed_out)
    dist.barrier()
    dist.destroy_process_group()


def test_in_library_exchange_world_size_2_on_one_gpu(torch_cuda, tmp_path):
    """The per-iteration exchange behind the C ABI (icp_exchange_*): two processes on the one GPU map each other's
    inbox through IPC, `icp_register` on each rank enqueues iteration kernel + sum/exchange/solve kernel per iteration,
    nothing else.  Same bits on both ranks, the single-process result up to the association of the float64 sums, and a
    peer that stays away produces ICP_ERR_EXCHANGE after the configured wait instead of a hung GPU."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "rank0.npz")
    mp.start_processes(_rank_main_exchange, args=(2, port, out), nprocs=2, join=True, start_method="spawn")
    got = np.load(out)
    assert np.array_equal(got["all"][0], got["all"][1])
    assert bool(got["timed_out"])
    model, scan = _small_problem()
    single = _ctx(height=32, width=1024, max_num_alignments=10, threshold_delta_pose=0.0, scheme="geman_mcclure",
                  sigma=0.3)
    single.map_set(model)
    ref = single.register(scan)
    assert int(got["targets"]) == scan.shape[0] and len(got["losses"]) == 10
    np.testing.assert_allclose(got["pose"], ref.pose, atol=1e-6)
    np.testing.assert_allclose(got["losses"], ref.losses, rtol=1e-9)
    np.testing.assert_allclose(got["pose2"], single.register(scan, ref.pose).pose, atol=1e-6)
    # world size 1 through the same kernel = the plain registration, bit for bit
    solo = _ctx(height=32, width=1024, max_num_alignments=10, threshold_delta_pose=0.0, scheme="geman_mcclure",
                sigma=0.3)
    solo.map_set(model)
    solo.exchange_connect([solo.exchange_create(0, 1)])
    r = solo.register(scan)
    assert np.array_equal(r.pose, ref.pose) and np.array_equal(r.losses, ref.losses)
    solo.exchange_destroy()
    assert np.array_equal(solo.register(scan).pose, ref.pose)


def test_bench_gpus_2_launches_its_own_ranks(torch_cuda):
    """`python bench.py --gpus 2` with WORLD_SIZE unset must start its ranks itself (it re-executes
    under torch.distributed.run) and print ONE JSON line with n_gpus = 2 carrying the replicas headline and the
    `sharded` / `c4` objects.  Here: two ranks on the one GPU over gloo (BENCH_DIST_BACKEND), the launcher path that the
    driver's multi-GPU node takes with RCCL."""
    import json
    import subprocess
    env = dict(os.environ, BENCH_DIST_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "3",
                        "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["dist_backend"] == "gloo" and d["rccl_ranks"] == 0
    assert d["value"] > 0 and d["scaling"] == "weak" and d["max_pose_error_vs_ground_truth_m"] < 0.05
    sh = d["sharded"]
    for variant in ("library", "collective"):  # a failed exchange is REPORTED (and the collective still runs), never a hang
        assert "value" in sh[variant] or "error" in sh[variant], sh
    assert "value" in sh["collective"], sh
    assert sh["collective"]["max_pose_error_vs_ground_truth_m"] < 0.05
    assert "map_sharded_normals" in d["c4"], d["c4"]
