"""GPU: B sequences per launch (`icp_batch_*`).  The batched registration must give every member the poses, losses and
steps of the same registration run on its context alone — bit for bit: per sequence the launch runs the single launch's
body on the single launch's arguments (read from a descriptor table instead of the kernel-argument segment).

Reference semantics per sequence: ICPFrameToModel.register_new_frame (slam/odometry/icp_odometry.py:248-299) followed by
the pose-only branch of __update_map (:379)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X (no CPU fallback exists for the product path)")
    return torch


def _ctx(**kw):
    from pylidar_slam_amd.engine import IcpContext
    return IcpContext(**kw)


def _sequences(count, height, width, map_points, frames, seed0=1234, map_scans=4):
    """`count` independent sequences: (tracked scans, fixed map in the frame of the first tracked scan's predecessor)."""
    from pylidar_slam_amd.synthetic import SceneConfig, make_fixed_map, make_sequence
    out = []
    for s in range(count):
        cfg = SceneConfig(height=height, width=width, seed=seed0 + 1000 * s, step=0.3 + 0.05 * s)
        scans, poses = make_sequence(cfg, map_scans + frames)
        model = make_fixed_map(cfg, scans[:map_scans], poses[:map_scans], ref_frame=map_scans - 1, num_points=map_points)
        out.append((scans[map_scans:], model))
    return out


def _run_single(kw, options, seq, frames, init_mode, device=None, sync_order=False):
    scans, model = seq
    ctx = _ctx(**kw)
    for k, v in options.items():
        ctx.set_option(k, v)
    ctx.map_set(device.from_numpy(model).cuda() if device is not None else model)
    out, init = [], None
    for f in range(frames):
        scan = device.from_numpy(scans[f]).cuda() if device is not None else scans[f]
        ctx.register_launch(scan, "last" if (init_mode == "last" and f > 0) else init)
        if sync_order:  # (the plugin's order: the pose first — it decides about a key frame —, then the map)
            r = ctx.register_end()
            ctx.map_update(r.pose, None)
        else:
            ctx.map_update(None, None)
            r = ctx.register_end()
        out.append(r)
        init = r.pose
    mp = ctx.map_points()
    fb = ctx.handoff_fallbacks()
    ctx.close()
    return out, mp, fb


def _run_batch(kw, options, seqs, frames, init_mode, device=None, lengths=None, sync_order=False):
    from pylidar_slam_amd.engine import IcpBatch
    ctxs = []
    for scans, model in seqs:
        ctx = _ctx(**kw)
        for k, v in options.items():
            ctx.set_option(k, v)
        ctx.map_set(device.from_numpy(model).cuda() if device is not None else model)
        ctxs.append(ctx)
    batch = IcpBatch(ctxs)
    out = [[] for _ in seqs]
    inits = None
    for f in range(frames):
        scans = []
        for b, (sc, _) in enumerate(seqs):
            a = sc[f] if lengths is None else sc[f][:lengths[b]]
            scans.append(device.from_numpy(a).cuda() if device is not None else a)
        batch.register_launch(scans, "last" if (init_mode == "last" and f > 0) else inits)
        if sync_order:
            res = batch.register_end()  # (a live threshold: further chunks are enqueued here while a member still runs)
            for c, r in zip(ctxs, res):
                c.map_update(r.pose, None)
        else:
            batch.map_update()
            res = batch.register_end()
        for b, r in enumerate(res):
            out[b].append(r)
        inits = [r.pose for r in res]
    maps = [c.map_points() for c in ctxs]
    fbs = [c.handoff_fallbacks() for c in ctxs]
    batch.close()
    for c in ctxs:
        c.close()
    return out, maps, fbs


def _assert_same(single, batched, tag):
    (rs, mp_s, fb_s), (rb, mp_b, fb_b) = single, batched
    assert fb_s == 0 and fb_b == 0, tag
    for f, (a, b) in enumerate(zip(rs, rb)):
        assert a.iterations == b.iterations and a.converged == b.converged and a.num_targets == b.num_targets, (tag, f)
        assert np.array_equal(a.pose, b.pose), (tag, f, np.abs(a.pose - b.pose).max())
        assert np.array_equal(a.params, b.params), (tag, f)
        assert np.array_equal(a.losses, b.losses) and np.array_equal(a.dx, b.dx), (tag, f)
    assert np.array_equal(mp_s, mp_b), tag


@pytest.mark.parametrize("variant", ["default", "no_lead", "narrow_only", "live_threshold", "from_last", "host_arrays",
                                     "ragged", "hit_records", "late_kernel", "live_threshold_pose_first",
                                     "many_iterations_pose_first"])
def test_batched_registration_equals_single_sequences(torch_cuda, variant):
    """Three sequences with different scenes, four chained frames each (registration from the previous pose, pose-only map
    update by the device-resident pose): batched vs every sequence alone on a context of its own — poses, parameters,
    per-iteration losses and steps, iteration counts and the re-expressed maps, all bit-equal.  Variants: lead launches off
    (a summing / solving launch per iteration for all members), the 512-thread shape from the first iteration, a live stop
    threshold (members stop after different numbers of iterations: a finished member idles on the device), the initial
    guess read on the device, host arrays in, scans of different lengths in one batch, hit records and the late kernel (both
    off by default), and the plugin's order — poses first, then the maps — with a live threshold: the batch enqueues a first
    chunk of iterations and further chunks from `register_end` while a member is still running."""
    kw = dict(height=32, width=1024, max_num_alignments=12, threshold_delta_pose=0.0, scheme="geman_mcclure", sigma=0.3)
    options, init_mode, device, lengths, sync_order = {}, "pose", torch_cuda, None, False
    if variant == "no_lead":
        options = {"lead_solve": 0}
    elif variant == "narrow_only":
        options = {"wide_until": 0}
    elif variant == "live_threshold":
        kw["threshold_delta_pose"] = 1.0e-4
        kw["max_num_alignments"] = 15
    elif variant == "from_last":
        init_mode = "last"
    elif variant == "host_arrays":
        device = None
    elif variant == "ragged":
        lengths = [32 * 1024, 20 * 1024 + 77, 9 * 1024 + 5]
    elif variant == "live_threshold_pose_first":
        kw["threshold_delta_pose"] = 1.0e-4
        kw["max_num_alignments"] = 15
        sync_order = True
    elif variant == "many_iterations_pose_first":  # (the reference's dataclass default: 100 — most of them never enqueued)
        kw["threshold_delta_pose"] = 1.0e-5
        kw["max_num_alignments"] = 100
        sync_order = True
    elif variant == "hit_records":
        options = {"hit_records": 1}
    elif variant == "late_kernel":  # (the late kernel from the third launch on, batched: k_iterate_late_batch)
        options = {"hit_records": 1, "late_from": 2, "wide_until": 0}
    seqs = _sequences(3, 32, 1024, 30_000, 4)
    if lengths is not None:
        seqs = [([s[:lengths[b]] for s in sc], m) for b, (sc, m) in enumerate(seqs)]
    batched = _run_batch(kw, options, seqs, 4, init_mode, device, sync_order=sync_order)
    for b, seq in enumerate(seqs):
        single = _run_single(kw, options, seq, 4, init_mode, device, sync_order=sync_order)
        _assert_same(single, ([r for r in batched[0][b]], batched[1][b], batched[2][b]), (variant, b))
    if variant in ("live_threshold", "live_threshold_pose_first", "many_iterations_pose_first"):
        its = [[r.iterations for r in batched[0][b]] for b in range(3)]
        assert any(i < kw["max_num_alignments"] for row in its for i in row), its  # (the threshold did fire: members went idle inside the batch)


def test_batched_registration_at_benchmark_size(torch_cuda):
    """BASELINE configs[1] sizes (64x2048 scans against 100 000-point maps, 20 forced iterations), four sequences per
    launch, three chained frames: bit-equal to the four sequences run alone, and every pose within 1e-4 m / 1e-4 rad of
    the generator's ground truth motion is NOT asserted here (noise-limited: the C2 parity test pins the pose on the
    reference) — this test pins batched == single."""
    kw = dict(height=64, width=2048, max_num_alignments=20, threshold_delta_pose=0.0, scheme="geman_mcclure", sigma=0.3)
    seqs = _sequences(4, 64, 2048, 100_000, 3, seed0=4321, map_scans=8)
    batched = _run_batch(kw, {}, seqs, 3, "pose", torch_cuda)
    for b, seq in enumerate(seqs):
        single = _run_single(kw, {}, seq, 3, "pose", torch_cuda)
        _assert_same(single, ([r for r in batched[0][b]], batched[1][b], batched[2][b]), ("c2", b))
        assert all(r.iterations == 20 for r in batched[0][b])


def test_batch_refuses_what_it_cannot_run(torch_cuda):
    from pylidar_slam_amd.engine import IcpBatch
    kw = dict(height=32, width=1024, max_num_alignments=8, threshold_delta_pose=0.0)
    (scans, model), = _sequences(1, 32, 1024, 20_000, 1)
    a, b = _ctx(**kw), _ctx(**dict(kw, max_num_alignments=9))
    with pytest.raises(AssertionError):
        IcpBatch([a, a])  # the same context twice
    batch = IcpBatch([a, b])
    a.map_set(model)
    b.map_set(model)
    with pytest.raises(AssertionError, match="max_num_alignments"):
        batch.register_launch([scans[0], scans[0]])
    batch.close()
    b.close()
    c = _ctx(**kw)
    c.map_set(model)
    c.set_cost("point_to_point_gauss_newton")
    batch = IcpBatch([a, c])
    with pytest.raises(AssertionError, match="point-to-plane"):
        batch.register_launch([scans[0], scans[0]])
    # ... and the members are still usable on their own afterwards
    r = a.register(scans[0])
    assert r.iterations == 8
    batch.close()
    a.close()
    c.close()


def test_two_contexts_with_chunked_launches_on_one_device(torch_cuda):
    """whether a registration runs on lead launches is decided ONCE, when it begins.  Two contexts of one
    process register on the same device with a live stop threshold (chunked launches: a first chunk, further chunks from
    icp_register_end); the second one begins and ends while the first one's chunks are still being enqueued — which used to
    flip the first one between lead and plain launches from chunk to chunk (a stale pose from the mailbox).  Results must be
    those of the same registration alone."""
    kw = dict(height=32, width=1024, max_num_alignments=15, threshold_delta_pose=1.0e-6, scheme="geman_mcclure", sigma=0.3)
    seqs = _sequences(2, 32, 1024, 30_000, 2, seed0=77)
    alone = [_run_single(kw, {}, s, 2, "pose", torch_cuda) for s in seqs]
    a, b = _ctx(**kw), _ctx(**kw)
    dev = [[torch_cuda.from_numpy(x).cuda() for x in s[0]] for s in seqs]
    a.map_set(torch_cuda.from_numpy(seqs[0][1]).cuda())
    b.map_set(torch_cuda.from_numpy(seqs[1][1]).cuda())
    ia = ib = None
    for f in range(2):
        a.register_launch(dev[0][f], ia)  # first chunk of a (lead launches: a is alone at this point)
        b.register_launch(dev[1][f], ib)  # b begins while a is pending -> b runs plain launches, a must not change its mind
        rb = b.register_end()             # b ends: a is alone again when its next chunk is enqueued
        ra = a.register_end()
        a.map_update(ra.pose, None)
        b.map_update(rb.pose, None)
        for r, ref in ((ra, alone[0][0][f]), (rb, alone[1][0][f])):
            assert r.iterations == ref.iterations and r.converged == ref.converged
            assert np.array_equal(r.pose, ref.pose) and np.array_equal(r.losses, ref.losses) and np.array_equal(r.dx, ref.dx)
        ia, ib = ra.pose, rb.pose
    assert a.handoff_fallbacks() == 0 and b.handoff_fallbacks() == 0
    a.close()
    b.close()


def test_batched_projection_equals_single(torch_cuda):
    """`icp_batch_project`: the vertex maps of three scans of different lengths in two launches = the three single
    projections (Projector.build_projection_map, slam/common/projection.py:331-418), bit for bit — twice in a row (the
    resolve pass leaves every member's z-buffer clean for the next frame)."""
    from pylidar_slam_amd.engine import IcpBatch
    kw = dict(height=32, width=1024)
    seqs = _sequences(3, 32, 1024, 20_000, 2)
    ctxs = [_ctx(**kw) for _ in seqs]
    solo = [_ctx(**kw) for _ in seqs]
    batch = IcpBatch(ctxs)
    lengths = [32 * 1024, 17 * 1024 + 3, 1]
    for f in range(2):
        scans = [torch_cuda.from_numpy(sc[f][:lengths[b]].copy()).cuda() for b, (sc, _) in enumerate(seqs)]
        outs = [torch_cuda.empty((3, 32, 1024), dtype=torch_cuda.float32, device="cuda") for _ in seqs]
        batch.project(scans, outs)
        for b in range(3):
            ref = solo[b].project(scans[b])
            assert torch_cuda.equal(outs[b], ref), (f, b)
    batch.close()
    for c in ctxs + solo:
        c.close()
