"""Seeded synthetic LiDAR scans: a spinning multi-beam sensor ray-cast inside a closed box room with interior boxes.

This is the workload generator of SURVEY.md §8(d) / BASELINE.md §2 (the reference ships no data and the GPU box has no
datasets): every ray hits a surface, so a scan has exactly H*W points, ring-major, float32, sensor frame.

The beam elevations / azimuths are placed a quarter pixel inside the cells of the reference's spherical projector
(`SphericalProjector(H, W, 3, up_fov, down_fov)`, reference slam/common/projection.py:11-73: row 0 <-> +up_fov,
col = 0.5 * (-atan2(y, x) / pi + 1) * W) so the noiseless scan projects one point per pixel.
"""
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np

__all__ = ["SceneConfig", "ray_directions", "pose_matrix", "trajectory", "render_scan", "make_sequence",
           "make_fixed_map", "rotate_rows", "make_c2_workload", "make_c2_workloads", "LOOP_PERIOD"]


# ----------------------------------------------------------------------------------------------------------------------
# rows @ R.T without BLAS.  The generator's bits must not depend on how a BLAS library partitions a 131072 x 3 product
# among its threads (round 5: `dirs @ R.T` called from several Python threads at once rounded differently from run to
# run on a 256-thread host — a ray that grazes a box edge then lands metres away, and the golden fixtures, which hold the
# sha1 of the generated inputs, no longer match).  What OpenBLAS computes for such a row is the fused chain
# fma(x2, r2, fma(x1, r1, x0 * r0)); that chain is evaluated here with error-free transformations (Dekker's product,
# Knuth's sum) and ONE rounding to odd in front of the final addition (Boldo & Melquiond, "Emulation of FMA and correctly
# rounded sums: proved algorithms using rounding to odd", IEEE TC 2008): the correctly rounded a * b + c in float64, element
# by element, in a fixed order — the same bits as the BLAS product the fixtures were generated with, on every host.
# ----------------------------------------------------------------------------------------------------------------------
_SPLIT = 134217729.0  # 2^27 + 1 (Veltkamp)


def _two_sum(a, b):
    s = a + b
    bb = s - a
    return s, (a - (s - bb)) + (b - bb)


def _two_prod(a, b):
    p = a * b
    ca = _SPLIT * a
    ah = ca - (ca - a)
    al = a - ah
    cb = _SPLIT * b
    bh = cb - (cb - b)
    bl = b - bh
    return p, ((ah * bh - p) + ah * bl + al * bh) + al * bl


def _fma(a, b, c):
    """Correctly rounded a * b + c (float64 arrays / scalars; no overflow or underflow in the products)."""
    uh, ul = _two_prod(a, b)
    th, tl = _two_sum(c, uh)
    s, e = _two_sum(tl, ul)  # v = round-to-odd(tl + ul): RN, then one step towards the exact sum where RN came out even
    s = np.asarray(s, dtype=np.float64)
    even = (s.view(np.int64) & 1) == 0
    v = np.where((e != 0.0) & even, np.nextafter(s, np.where(e > 0.0, np.inf, -np.inf)), s)
    return th + v


def rotate_rows(rows: np.ndarray, rot: np.ndarray) -> np.ndarray:
    """`rows @ rot.T` for [N,3] float64 rows and a 3x3 matrix: column j = fma(x2, r_j2, fma(x1, r_j1, x0 * r_j0))."""
    rows = np.asarray(rows, dtype=np.float64)
    rot = np.asarray(rot, dtype=np.float64)
    x0, x1, x2 = (np.ascontiguousarray(rows[:, k]) for k in range(3))
    return np.stack([_fma(x2, rot[j, 2], _fma(x1, rot[j, 1], x0 * rot[j, 0])) for j in range(3)], axis=1)


@dataclass
class SceneConfig:
    height: int = 64
    width: int = 2048
    up_fov: float = 3.0
    down_fov: float = -24.0
    noise_sigma: float = 0.01
    seed: int = 1234
    # closed room: [xmin, xmax, ymin, ymax, zmin, zmax] (world frame, sensor starts at the origin 1.7 m above the floor)
    room: Tuple[float, ...] = (-20.0, 25.0, -15.0, 15.0, -1.7, 6.0)
    # interior axis-aligned boxes so that all 6 DoF are constrained
    boxes: List[Tuple[float, ...]] = field(default_factory=lambda: [
        (6.0, 9.0, 4.0, 7.0, -1.7, 2.5),
        (12.0, 14.0, -9.0, -5.0, -1.7, 4.0),
        (-8.0, -5.0, 6.0, 10.0, -1.7, 1.5),
        (-12.0, -9.5, -8.0, -4.0, -1.7, 3.0),
        (2.0, 3.0, -6.0, -5.0, -1.7, 6.0),
        (16.0, 19.0, 8.0, 11.0, -1.7, 2.0),
        (-3.0, -2.0, 11.0, 12.0, -1.7, 6.0),
        (19.0, 22.0, -3.0, 1.0, -1.7, 1.0),
    ])
    # trajectory: forward step per frame (m), yaw per frame (rad)
    step: float = 0.4
    yaw_rate: float = 0.01


def ray_directions(cfg: SceneConfig) -> np.ndarray:
    """Unit ray directions [H*W, 3] (float64), ring-major (row = beam, col = azimuth step)."""
    h, w = cfg.height, cfg.width
    fov = abs(cfg.up_fov) + abs(cfg.down_fov)
    phi = np.deg2rad(cfg.up_fov - (np.arange(h) + 0.25) * fov / h)  # row i + 0.25
    theta = ((np.arange(w) + 0.25) / w * 2.0 - 1.0) * np.pi  # col j + 0.25 ; theta = -atan2(y, x)
    az = -theta
    cp, sp = np.cos(phi)[:, None], np.sin(phi)[:, None]
    d = np.stack([cp * np.cos(az)[None, :], cp * np.sin(az)[None, :], np.broadcast_to(sp, (h, w))], axis=-1)
    return d.reshape(-1, 3)


def _rot(ex: float, ey: float, ez: float) -> np.ndarray:
    """R = Rz(ez) Ry(ey) Rx(ex) (the reference's euler 'xyz' convention, slam/common/rotation.py:144-150)."""
    cx, sx, cy, sy, cz, sz = np.cos(ex), np.sin(ex), np.cos(ey), np.sin(ey), np.cos(ez), np.sin(ez)
    rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return rz @ ry @ rx


def pose_matrix(params) -> np.ndarray:
    """[tx, ty, tz, ex, ey, ez] -> 4x4 float64."""
    t = np.eye(4)
    t[:3, :3] = _rot(params[3], params[4], params[5])
    t[:3, 3] = params[:3]
    return t


def trajectory(cfg: SceneConfig, num_frames: int) -> np.ndarray:
    """Absolute sensor poses [F, 4, 4] float64 (world <- sensor): forward drive with slow yaw and small wobble."""
    poses = np.zeros((num_frames, 4, 4))
    x = y = 0.0
    yaw = 0.0
    for f in range(num_frames):
        roll = 0.002 * np.sin(0.7 * f)
        pitch = 0.002 * np.sin(0.4 * f + 1.0)
        z = 0.02 * np.sin(0.5 * f)
        poses[f] = pose_matrix(np.array([x, y, z, roll, pitch, yaw]))
        x += cfg.step * np.cos(yaw)
        y += cfg.step * np.sin(yaw)
        yaw += cfg.yaw_rate
    return poses


def loop_trajectory(cfg: SceneConfig, period: int = 96, centre=(1.5, -5.0)) -> np.ndarray:
    """Closed circuit [period, 4, 4] float64 (world <- sensor): a circle driven at `cfg.step` m per frame (radius
    step * period / 2 pi = 6.1 m for the defaults, 3.75 deg of yaw per frame, >= 3.9 m clear of every box) with
    periodic wobble, so pose[period] == pose[0] and every consecutive relative motion is (nearly) the same twist: the
    steady-driving regime in which a constant-velocity initial guess is right, for arbitrarily long runs."""
    radius = cfg.step * period / (2.0 * np.pi)
    poses = np.zeros((period, 4, 4))
    for f in range(period):
        a = 2.0 * np.pi * f / period
        roll = 0.002 * np.sin(5.0 * a)
        pitch = 0.002 * np.sin(3.0 * a + 1.0)
        z = 0.02 * np.sin(4.0 * a)
        poses[f] = pose_matrix(np.array([centre[0] + radius * np.sin(a), centre[1] - radius * np.cos(a), z,
                                         roll, pitch, a]))
    return poses


def _ray_box_exit(o, d, box):
    """Distance at which rays starting INSIDE the box leave it (slab method)."""
    lo = np.array([box[0], box[2], box[4]])
    hi = np.array([box[1], box[3], box[5]])
    with np.errstate(divide="ignore", invalid="ignore"):
        t1 = (lo - o) / d
        t2 = (hi - o) / d
    return np.min(np.maximum(t1, t2), axis=1)


def _ray_box_entry(o, d, box):
    """Distance at which rays starting OUTSIDE the box enter it (inf if missed)."""
    lo = np.array([box[0], box[2], box[4]])
    hi = np.array([box[1], box[3], box[5]])
    with np.errstate(divide="ignore", invalid="ignore"):
        t1 = (lo - o) / d
        t2 = (hi - o) / d
    tn = np.max(np.minimum(t1, t2), axis=1)
    tf = np.min(np.maximum(t1, t2), axis=1)
    hit = (tn <= tf) & (tn > 0.0)
    return np.where(hit, tn, np.inf)


def render_scan(cfg: SceneConfig, pose: np.ndarray, frame: int, dirs: Optional[np.ndarray] = None) -> np.ndarray:
    """One scan [H*W, 3] float32 in the sensor frame taken from absolute pose `pose` (noise rng = seed + frame)."""
    if dirs is None:
        dirs = ray_directions(cfg)
    o = pose[:3, 3]
    dw = rotate_rows(dirs, pose[:3, :3])  # (not `dirs @ R.T`: see rotate_rows)
    t = _ray_box_exit(o, dw, cfg.room)
    for b in cfg.boxes:
        t = np.minimum(t, _ray_box_entry(o, dw, b))
    pts = dirs * t[:, None]
    rng = np.random.default_rng(cfg.seed + frame)
    pts = pts + rng.normal(0.0, cfg.noise_sigma, size=pts.shape)
    return pts.astype(np.float32)


def make_sequence(cfg: SceneConfig, num_frames: int):
    """Returns (scans: list of [H*W,3] f32, absolute gt poses [F,4,4] f64)."""
    dirs = ray_directions(cfg)
    poses = trajectory(cfg, num_frames)
    return [render_scan(cfg, poses[f], f, dirs) for f in range(num_frames)], poses


def make_fixed_map(cfg: SceneConfig, scans, poses, ref_frame: int, num_points: int = 100_000, voxel: float = 0.25,
                   seed: int = 7) -> np.ndarray:
    """A fixed local map of exactly `num_points` points expressed in the frame of `poses[ref_frame]`.

    Union of voxel-subsampled (one point per `voxel` cell, first occurrence) scans, randomly sub-selected with
    rng(seed) — the C2 workload of SURVEY.md §8(d) (loaded through the `set_map_pointcloud` path,
    reference slam/odometry/local_map.py:289-299).
    """
    inv_ref = np.linalg.inv(poses[ref_frame])
    clouds = []
    for s, p in zip(scans, poses):
        rel = inv_ref @ p
        pts = rotate_rows(s.astype(np.float64), rel[:3, :3]) + rel[:3, 3]
        keys = np.round(pts / voxel).astype(np.int64)
        _, first = np.unique(keys, axis=0, return_index=True)
        clouds.append(pts[np.sort(first)])
    cloud = np.concatenate(clouds, axis=0)
    rng = np.random.default_rng(seed)
    if cloud.shape[0] < num_points:
        raise ValueError(f"only {cloud.shape[0]} candidate map points (< {num_points}); add scans or shrink voxel")
    sel = np.sort(rng.choice(cloud.shape[0], size=num_points, replace=False))
    return cloud[sel].astype(np.float32)


# ----------------------------------------------------------------------------------------------------------------------
# The benchmark's C2 workload (bench.py): scans to track, ground truth, a fixed 100 000-point map none of the
# tracked scans is part of.  Lives here — numpy only — so that several sequences can be generated by worker PROCESSES.
# ----------------------------------------------------------------------------------------------------------------------
LOOP_PERIOD = 96


def make_c2_workload(seq: int, trajectory_name: str, frames_needed: int):
    """(scans: dict frame -> [N,3] f32, ground-truth poses, the fixed 100k-point map in the frame of `start`, the order
    in which the frames are visited, start).  No tracked scan contributes to the map (except `pingpong_r01`)."""
    seed = 1234 + 1000 * seq

    def into_frame(model, poses, src, dst):
        rel = np.linalg.inv(poses[dst]) @ poses[src]
        return (rotate_rows(model.astype(np.float64), rel[:3, :3]) + rel[:3, 3]).astype(np.float32)  # (BLAS-free: same bits on every host)

    if trajectory_name == "pingpong":
        # half-steps: even poses = the mapping pass (never tracked), odd poses = the tracked sequence
        cfg = SceneConfig(height=64, width=2048, seed=seed, step=0.2, yaw_rate=0.005)
        scans, poses = make_sequence(cfg, 16)
        even = list(range(0, 16, 2))
        model = make_fixed_map(cfg, [scans[f] for f in even], poses[even], ref_frame=0, num_points=100_000)
        order = list(range(3, 16, 2)) + list(range(13, 0, -2))  # 3,5,..,15,13,..,1 then repeats
        return {f: scans[f] for f in range(1, 16, 2)}, poses, into_frame(model, poses, 0, 1), order, 1
    if trajectory_name == "pingpong_r01":
        cfg = SceneConfig(height=64, width=2048, seed=seed)
        scans, poses = make_sequence(cfg, 8)
        model = make_fixed_map(cfg, scans, poses, ref_frame=0, num_points=100_000)
        return dict(enumerate(scans)), poses, model, list(range(1, 8)) + list(range(6, -1, -1)), 0
    # loop: 192 half-steps around the circuit; odd poses tracked (96, 0.4 m apart), 8 even poses mapped
    cfg = SceneConfig(height=64, width=2048, seed=seed, step=0.2)
    poses = loop_trajectory(cfg, 2 * LOOP_PERIOD)
    order = list(range(3, 2 * LOOP_PERIOD, 2)) + [1]
    map_frames = list(range(0, 2 * LOOP_PERIOD, 2 * LOOP_PERIOD // 8))
    dirs = ray_directions(cfg)
    tracked = order[:min(frames_needed, LOOP_PERIOD)]
    scans = {f: render_scan(cfg, poses[f], f, dirs) for f in sorted(set(tracked) | set(map_frames))}
    model = make_fixed_map(cfg, [scans[f] for f in map_frames], poses[map_frames], ref_frame=0, num_points=100_000)
    return {f: scans[f] for f in tracked}, poses, into_frame(model, poses, 0, 1), order, 1


def make_c2_workloads(seqs, trajectory_name: str, frames_needed: int, workers: int = 0):
    """`make_c2_workload` for several sequences.  workers > 1: by that many worker PROCESSES at a time (`python -m
    pylidar_slam_amd.synthetic ...`: fresh interpreters that import numpy and this module only — no fork of a process that
    holds a GPU context, no re-import of the caller's main module; every worker is single-threaded numpy, so the bits are
    those of the sequential generation — bench.py found THREADS of one process not to be, on a 256-thread host, while the
    generator still called BLAS).  A worker that fails has its sequence generated here."""
    seqs = list(seqs)
    out = {}
    if workers > 1 and len(seqs) > 1:
        import os
        import pickle
        import subprocess
        import sys
        import tempfile
        pkg_parent = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env = dict(os.environ, PYTHONPATH=pkg_parent + os.pathsep + os.environ.get("PYTHONPATH", ""), OMP_NUM_THREADS="1",
                   OPENBLAS_NUM_THREADS="1")
        with tempfile.TemporaryDirectory(prefix="c2_workloads_") as tmp:
            todo, running = list(seqs), []
            while todo or running:
                while todo and len(running) < workers:
                    sq = todo.pop(0)
                    path = os.path.join(tmp, f"{sq}.pkl")
                    try:
                        proc = subprocess.Popen([sys.executable, "-m", "pylidar_slam_amd.synthetic", str(sq), trajectory_name,
                                                 str(frames_needed), path], env=env, stdout=subprocess.DEVNULL,
                                                stderr=subprocess.DEVNULL)
                        running.append((sq, proc, path))
                    except Exception:
                        pass  # (generated below)
                if not running:
                    break
                sq, proc, path = running.pop(0)
                try:
                    if proc.wait(timeout=600) == 0:
                        with open(path, "rb") as f:
                            out[sq] = pickle.load(f)
                except Exception:
                    proc.kill()
    return [out[sq] if sq in out else make_c2_workload(sq, trajectory_name, frames_needed) for sq in seqs]


if __name__ == "__main__":  # worker of make_c2_workloads: SEQ TRAJECTORY FRAMES OUTPATH
    import pickle
    import sys
    _seq, _traj, _frames, _path = int(sys.argv[1]), sys.argv[2], int(sys.argv[3]), sys.argv[4]
    with open(_path, "wb") as _f:
        pickle.dump(make_c2_workload(_seq, _traj, _frames), _f, protocol=pickle.HIGHEST_PROTOCOL)
