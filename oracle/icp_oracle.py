"""CPU ORACLE for the pyLiDAR-SLAM frame-to-model ICP hot path — TEST INFRASTRUCTURE, NOT THE PRODUCT.

A numpy (+ scipy.spatial.cKDTree) restatement of the reference algorithm, function by function, each citing the
reference file:line (relative to /root/reference) it follows. Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import this module; the product (`pylidar-slam_amd/`) never does.

Pinning status: the reference holds NO golden vectors for this path. This restatement is pinned by
  * `tests/golden/*.npz`, produced by running the reference's OWN `ICPFrameToModel` / `GridSample` /
    `SphericalProjector` / `GaussNewton` unmodified (through the import shims in `oracle/shims/`) on seeded synthetic
    scans — generator script `oracle/make_golden.py`;
  * the reference's property tests restated in `tests/test_oracle.py` (tests/test_optimization.py:9-32 least-square
    variant, tests/test_pointcloud.py:7-25).
Unpinned third-party arithmetic (stated, not hidden): `pykdtree` (exact kNN; stood in for by cKDTree — same answer
except on exact distance ties), numba `fastmath` code generation for `voxelise`, LAPACK `sgesdd` rounding for normals.

Everything is float32 where the reference is float32; poses accumulate in float64 (icp_odometry.py:200-202).
"""
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np
from scipy.spatial import cKDTree

F32 = np.float32
F64 = np.float64


# ======================================================================================================================
# a1 / a2  spherical projection + z-buffered projection map
# ======================================================================================================================
def spherical_projection(pc: np.ndarray, height: int, width: int, up_fov: float, down_fov: float):
    """reference slam/common/projection.py:11-73 (`torch__spherical_projection`), called with
    (min_vertical_fov=up_fov, max_vertical_fov=down_fov) by SphericalProjector.project_pointcloud (:474-476).

    pc [N, 3] f32 -> (rows, cols, r) f32; invalid (r == 0) points get row = col = -1, r = 0.
    """
    pc = np.asarray(pc, dtype=F32)
    fov_up = up_fov / 180.0 * np.pi  # :47
    fov_down = down_fov / 180.0 * np.pi  # :48
    fov = abs(fov_down) + abs(fov_up)  # :49
    r = np.sqrt((pc * pc).sum(axis=1, dtype=F32)).astype(F32)  # :52
    mask_0 = (r == 0.0).astype(F32)  # :55
    mask_valid = F32(1.0) - mask_0
    r = mask_0 * F32(0.001) + mask_valid * r  # :57
    x, y, z = pc[:, 0], pc[:, 1], pc[:, 2]
    theta = -np.arctan2(y, x)  # :64
    phi = np.arcsin(z / r)  # :65
    proj_col = F32(0.5) * (theta / F32(np.pi) + F32(1.0))  # :67
    proj_row = F32(1.0) - (phi + F32(abs(fov_down))) / F32(fov)  # :68
    proj_col = proj_col * F32(width)  # :70
    proj_row = proj_row * F32(height)  # :71
    return (proj_row * mask_valid - mask_0).astype(F32), (proj_col * mask_valid - mask_0).astype(F32), \
        (r * mask_valid).astype(F32)  # :73


def build_projection_map(pc: np.ndarray, height: int, width: int, up_fov: float, down_fov: float,
                         channels: Optional[np.ndarray] = None, return_index: bool = False):
    """reference slam/common/projection.py:331-418 (`Projector.build_projection_map`), batch of 1.

    Nearest point wins each pixel: points are written in order of DESCENDING range, last write wins (:405-415).
    Ties in range at one pixel are unspecified in the reference (unstable torch.argsort); this oracle fixes them as
    "stable descending sort, last write wins" = the HIGHEST original index among equal ranges wins.

    Returns vmap [C, H, W] f32 (zeros where empty) and optionally the winning point index per pixel (-1 = empty).
    """
    pc = np.asarray(pc, dtype=F32)
    image_channels = pc[:, :3] if channels is None else np.asarray(channels, dtype=F32)
    rows, cols, _ = spherical_projection(pc[:, :3], height, width, up_fov, down_fov)
    r = np.sqrt((pc * pc).sum(axis=1, dtype=F32)).astype(F32)  # :394 (norm over ALL given channels of `pointcloud`)
    prow = np.round(rows)  # :395 round-half-even
    pcol = np.round(cols)  # :396
    invalid = ~((prow >= 0.0) & (prow <= height - 1) & (pcol >= 0.0) & (pcol <= width - 1))  # :398-401
    r = r.copy()
    r[invalid] = -1.0  # :404
    order = np.argsort(-r, kind="stable")  # :405 descending
    order = order[r[order] > 0.0]  # :409-411
    ir = prow[order].astype(np.int64)
    ic = pcol[order].astype(np.int64)
    c_dest = image_channels.shape[1]
    dest = np.zeros((c_dest, height, width), dtype=F32)
    index = np.full((height, width), -1, dtype=np.int64)
    # sequential "last write wins" (:415). numpy fancy assignment with repeated indices keeps the last value.
    dest[:, ir, ic] = image_channels[order].T
    index[ir, ic] = order
    if return_index:
        return dest, index
    return dest


def vertex_map_to_points(vmap: np.ndarray) -> np.ndarray:
    """reference slam/common/geometry.py:181-204 (`projection_map_to_points`, dim=0): [C,H,W] -> [H*W, C]."""
    c = vmap.shape[0]
    return np.ascontiguousarray(vmap.reshape(c, -1).T)


# ======================================================================================================================
# a4-a6 voxel grid sampling
# ======================================================================================================================
def voxelise(pc: np.ndarray, voxel: float) -> np.ndarray:
    """reference slam/common/pointcloud.py:54-79. numba types `pointcloud[i, 0] / voxel_x` as f32 / f64 -> f64,
    rounds half-to-even (`np.round_`) and truncates with `int()`; out is int64 [n, 3]."""
    q = np.asarray(pc).astype(np.float64) / np.float64(voxel)
    return np.round(q).astype(np.int64)


def voxel_hashing(voxels: np.ndarray) -> np.ndarray:
    """reference slam/common/pointcloud.py:13-23,40-51: 73856093 x + 19349669 y + 83492791 z in wrapping int64."""
    v = voxels.astype(np.int64)
    with np.errstate(over="ignore"):
        return np.int64(73856093) * v[:, 0] + np.int64(19349669) * v[:, 1] + np.int64(83492791) * v[:, 2]


def sample_from_hashes(pc: np.ndarray, hashes: np.ndarray):
    """reference slam/common/pointcloud.py:170-179: np.unique(return_index) = first occurrence per distinct hash,
    ordered by ascending int64 hash."""
    _, idx = np.unique(hashes, return_index=True)
    return pc[idx], idx


def voxel_normal_distribution(pc: np.ndarray, hashes: np.ndarray):
    """reference slam/common/pointcloud.py:83-167: voxels = runs of equal hash in sorted order; per voxel the point
    count, the mean and the UNNORMALISED covariance sum (p - mean)(p - mean)^T in the dtype of the points; voxel id of
    a point = rank of its hash among the distinct hashes.  (The order of the points inside a voxel — the reference's
    argsort is not stable — only moves the float32 sums by rounding.)"""
    uniq, inv, counts = np.unique(hashes, return_inverse=True, return_counts=True)
    order = np.argsort(hashes, kind="stable")
    starts = np.concatenate([[0], np.cumsum(counts)[:-1]])
    means = np.zeros((uniq.shape[0], 3), pc.dtype)
    covs = np.zeros((uniq.shape[0], 3, 3), pc.dtype)
    for v, (b, c) in enumerate(zip(starts, counts)):
        pts = pc[order[b:b + c]]
        mean = (pts.sum(axis=0).reshape(1, 3) / c).astype(pc.dtype)
        cen = (pts - mean).astype(pc.dtype)
        means[v] = mean[0]
        covs[v] = (cen.reshape(-1, 3, 1) * cen.reshape(-1, 1, 3)).sum(axis=0)
    return counts.astype(np.int64), means, covs, inv.astype(np.int64)


def grid_sample(pc: np.ndarray, voxel: float):
    """reference slam/common/pointcloud.py:182-195 and `GridSample.filter` slam/preprocessing.py:213-226."""
    return sample_from_hashes(pc, voxel_hashing(voxelise(pc, voxel)))


def kitti_correct_scan(scan: np.ndarray) -> np.ndarray:
    """`KITTIOdometrySequence.correct_scan` (slam/dataset/kitti_dataset.py:202-231): every point is rotated by 0.205 deg
    about u = (p x e_z) / |p x e_z|.  u and the outer products u u^T are float32 as in the reference; cos/sin are float64
    scalars, so (numpy >= 2 promotion, the stack this oracle is pinned on) the rotation and the result are float64."""
    p = np.asarray(scan)[:, :3].astype(F32)
    u = np.stack([p[:, 1], -p[:, 0], np.zeros(p.shape[0], F32)], axis=1)     # p x (0,0,1)
    u = (u / np.sqrt((u * u).sum(axis=1, keepdims=True, dtype=F32))).astype(F32)
    theta = 0.205 * np.pi / 180.0
    c, s = np.float64(np.cos(theta)), np.float64(np.sin(theta))
    outer = (u[:, :, None] * u[:, None, :]).astype(F32)                       # float32 products (:216)
    skew = np.zeros((p.shape[0], 3, 3), F32)
    skew[:, 0, 1], skew[:, 0, 2] = -u[:, 2], u[:, 1]
    skew[:, 1, 0], skew[:, 1, 2] = u[:, 2], -u[:, 0]
    skew[:, 2, 0], skew[:, 2, 1] = -u[:, 1], u[:, 0]
    rot = c * np.eye(3)[None] + s * skew.astype(F64) + (1.0 - c) * outer.astype(F64)
    return (rot * p.astype(F64)[:, None, :]).sum(axis=2)


def distort(pc: np.ndarray, timestamps: np.ndarray, rpose: np.ndarray) -> np.ndarray:
    """reference slam/preprocessing.py:144-191 (`Distortion.filter`): every point is moved by the
    fraction alpha = (t - t_min) / (t_max - t_min) of the initial motion estimate: rotation slerp(I, R, alpha)
    (scipy `Slerp`, i.e. exp(alpha * log R)) and translation alpha * t.  float64 out, like the reference's einsum."""
    ts = np.asarray(timestamps, dtype=np.float64).reshape(-1)
    diff = ts.max() - ts.min()
    alpha = ts * 0 if diff == 0.0 else (ts - ts.min()) / diff  # :177-179
    rot = np.asarray(rpose)[:3, :3].astype(np.float64)
    # log map: (R - R^T) / 2 = sin(theta) [axis]x, trace = 1 + 2 cos(theta)   (theta < pi)
    v = 0.5 * np.array([rot[2, 1] - rot[1, 2], rot[0, 2] - rot[2, 0], rot[1, 0] - rot[0, 1]])
    nv = np.linalg.norm(v)
    theta = np.arctan2(nv, 0.5 * (np.trace(rot) - 1.0))
    axis = v / nv if nv > 0 else np.zeros(3)
    phi = alpha * theta
    p = np.asarray(pc).astype(np.float64)
    c, s_ = np.cos(phi)[:, None], np.sin(phi)[:, None]
    rotated = p * c + np.cross(axis[None, :], p) * s_ + axis[None, :] * (p @ axis)[:, None] * (1.0 - c)
    return rotated + alpha[:, None] * np.asarray(rpose)[:3, 3].astype(np.float64)[None, :]  # :183-185


# ======================================================================================================================
# a9 / a17  pose parametrisation (euler xyz, R = Rz Ry Rx), float32 like the reference's torch tensors
# ======================================================================================================================
def euler_to_mat(angles, dtype=F32) -> np.ndarray:
    """reference slam/common/rotation.py:144-150 (`torch_euler_to_mat`): Rz(ez) @ Ry(ey) @ Rx(ex)."""
    a = np.asarray(angles, dtype=dtype)
    c, s = np.cos(a), np.sin(a)
    one, zero = dtype(1), dtype(0)
    rx = np.array([[one, zero, zero], [zero, c[0], -s[0]], [zero, s[0], c[0]]], dtype=dtype)
    ry = np.array([[c[1], zero, s[1]], [zero, one, zero], [-s[1], zero, c[1]]], dtype=dtype)
    rz = np.array([[c[2], -s[2], zero], [s[2], c[2], zero], [zero, zero, one]], dtype=dtype)
    return (rz @ ry @ rx).astype(dtype)


def build_pose_matrix(params, dtype=F32) -> np.ndarray:
    """reference slam/common/pose.py:120-144: [tx,ty,tz,ex,ey,ez] -> 4x4."""
    p = np.asarray(params, dtype=dtype).reshape(6)
    t = np.eye(4, dtype=dtype)
    t[:3, :3] = euler_to_mat(p[3:], dtype)
    t[:3, 3] = p[:3]
    return t


def mat_to_euler(rot: np.ndarray, eps: float = 1.0e-6) -> np.ndarray:
    """reference slam/common/rotation.py:253-270 (`torch_mat_to_euler`)."""
    dtype = rot.dtype.type
    sy = np.sqrt(rot[0, 0] * rot[0, 0] + rot[1, 0] * rot[1, 0])
    if not sy < eps:
        x = np.arctan2(rot[2, 1], rot[2, 2])
        y = np.arctan2(-rot[2, 0], sy)
        z = np.arctan2(rot[1, 0], rot[0, 0])
    else:
        x = np.arctan2(-rot[1, 2], rot[1, 1])
        y = np.arctan2(-rot[2, 0], sy)
        z = dtype(0)
    return np.array([x, y, z], dtype=dtype)


def from_pose_matrix(mat: np.ndarray) -> np.ndarray:
    """reference slam/common/pose.py:188-207."""
    return np.concatenate([mat[:3, 3], mat_to_euler(mat[:3, :3])]).astype(mat.dtype)


def apply_transformation(points: np.ndarray, mat: np.ndarray) -> np.ndarray:
    """reference slam/common/pose.py:169-186: p' = p @ R^T + t (float32)."""
    return (points @ mat[:3, :3].T + mat[:3, 3][None, :]).astype(points.dtype)


# ======================================================================================================================
# a15  robust weighting schemes + one Gauss-Newton step
# ======================================================================================================================
def ls_weights(scheme: str, sigma: float, res: np.ndarray, tgt: Optional[np.ndarray] = None,
               ref: Optional[np.ndarray] = None, eps: float = 1.0e-4) -> np.ndarray:
    """reference slam/common/optimization.py:18-226: w = sqrt(cost(r)) / clamp(|r|, eps) (:45-50);
    `least_square`/`default` short-circuit to w = 1 (:70-72)."""
    res = res.astype(F32)
    if scheme in ("default", "least_square"):
        return np.ones((1,), dtype=F32)
    s = F32(sigma)
    if scheme == "huber":  # :87-97
        a = np.abs(res)
        sq = a < s
        cost = sq * (res * res) + (~sq) * (F32(2) * s * a - s * s)
    elif scheme == "exp":  # :110-117
        cost = (res * res) * np.exp(-(res ** 2) / (s * s))
    elif scheme == "neighborhood":  # :132-145
        d = tgt.astype(F32) - ref.astype(F32)
        nrm = np.sqrt((d * d).sum(axis=-1, dtype=F32))
        cost = res * res * np.exp(-(nrm ** 2) / (s * s))
    elif scheme == "geman_mcclure":  # :158-166
        r2 = res ** 2
        cost = s * r2 / (s + r2)
    elif scheme == "square_geman_mcclure":  # :179-187
        r2 = res ** 2
        cost = r2 * (s / (s + r2)) ** 2
    elif scheme == "cauchy":  # :200-208
        cost = np.log(F32(1) + (res / s) ** 2)
    else:
        raise AssertionError(f"unknown scheme {scheme}")
    clamped = np.clip(np.abs(res), F32(eps), None)
    return (np.sqrt(cost.astype(F32)) / clamped).astype(F32)


@dataclass
class GNStep:
    dx: np.ndarray  # [6] f32
    loss: float  # sum (w r)^2
    H: np.ndarray  # [6,6]
    g: np.ndarray  # [6] = J^T (w r)
    stopped: bool  # residual norm < 1e-7 early return


def point_to_plane_rows(tgt: np.ndarray, ref: np.ndarray, normals: np.ndarray):
    """reference slam/common/optimization.py:356-435 evaluated at x0 = 0 (alignment.py:100-113):
    r_i = (p_i - q_i) . n_i (:427-431) ; J_i = [n_i, p_i x n_i] (:378-390 with dR/de_k at 0, rotation.py:166-184)."""
    p = tgt.astype(F32)
    n = normals.astype(F32)
    res = ((p - ref.astype(F32)) * n).sum(axis=-1, dtype=F32)
    jac = np.concatenate([n, np.cross(p, n).astype(F32)], axis=1)
    return res, jac


def gauss_newton_step(tgt: np.ndarray, ref: np.ndarray, normals: np.ndarray, scheme: str = "default",
                      sigma: float = 0.5, accumulate=F32) -> GNStep:
    """reference slam/common/optimization.py:296-344 with max_iters = 1 (alignment.py:77) from x0 = 0.

    `accumulate` = float32 restates the reference arithmetic; float64 is the "exact" variant the HIP path is compared
    with at tight tolerance (the device accumulates the normal equations in f64).
    Raises RuntimeError("Invalid Jacobian in Gauss Newton minimization") when |det H| < 1e-7 (:334-336).
    """
    res, jac = point_to_plane_rows(tgt, ref, normals)
    if np.sqrt((res.astype(np.float64) ** 2).sum()) < 1.0e-7:  # :323-327
        return GNStep(np.zeros(6, F32), float((res * res).sum()), np.zeros((6, 6)), np.zeros(6), True)
    w = ls_weights(scheme, sigma, res, tgt, ref)  # :328
    res = (res * w).astype(F32)  # :329
    jac = (jac * w.reshape(-1, 1)).astype(F32)  # :330
    ja = jac.astype(accumulate)
    ra = res.astype(accumulate)
    H = ja.T @ ja  # :332-333
    if abs(np.linalg.det(H)) < 1.0e-7:  # :334
        raise RuntimeError("Invalid Jacobian in Gauss Newton minimization")
    g = ja.T @ ra
    dx = -(np.linalg.inv(H) @ g)  # :338
    return GNStep(dx.astype(F32), float((ra * ra).sum()), H, g, False)


# ======================================================================================================================
# §8f-4  point-to-point Gauss-Newton step and weighted Procrustes
# ======================================================================================================================
def euler_jacobian(angles, dtype=F32) -> np.ndarray:
    """reference slam/common/rotation.py:166-187: [3,3,3] = d(Rz Ry Rx)/d(ex, ey, ez)."""
    ex, ey, ez = (dtype(a) for a in angles)
    cx, sx, cy, sy, cz, sz = np.cos(ex), np.sin(ex), np.cos(ey), np.sin(ey), np.cos(ez), np.sin(ez)
    z, o = dtype(0), dtype(1)
    rx = np.array([[o, z, z], [z, cx, -sx], [z, sx, cx]], dtype)
    ry = np.array([[cy, z, sy], [z, o, z], [-sy, z, cy]], dtype)
    rz = np.array([[cz, -sz, z], [sz, cz, z], [z, z, o]], dtype)
    jx = np.array([[z, z, z], [z, -sx, -cx], [z, cx, -sx]], dtype)
    jy = np.array([[-sy, z, cy], [z, z, z], [-cy, z, -sy]], dtype)
    jz = np.array([[-sz, -cz, z], [cz, -sz, z], [z, z, z]], dtype)
    return np.stack([rz @ ry @ jx, rz @ jy @ rx, jz @ ry @ rx]).astype(dtype)


def point_to_point_step(tgt: np.ndarray, ref: np.ndarray, x0=None, scheme: str = "default", sigma: float = 0.5,
                        accumulate=F32):
    """`GaussNewtonPointToPointAlignment.align` (slam/odometry/alignment.py:143-189) = one `GaussNewton.compute`
    iteration (optimization.py:296-344) of `PointToPointCost` (:458-560) linearised at x0:
    r_i = ||T(x0) p_i - q_i||, J_ik = (dT/dx_k p_i) . (T(x0) p_i - q_i) — the reference's own unnormalised form.
    Returns (pose [4,4] f32, params [6] f32 = x0 + dx, loss)."""
    x0 = np.zeros(6, F32) if x0 is None else np.asarray(x0, F32)
    p, q = tgt.astype(F32), ref.astype(F32)
    T0 = build_pose_matrix(x0)
    d = (apply_transformation(p, T0) - q).astype(F32)
    res = np.sqrt((d * d).sum(axis=-1, dtype=F32))
    dR = euler_jacobian(x0[3:])
    jac = np.concatenate([d, np.stack([((p @ dR[k].T).astype(F32) * d).sum(axis=-1, dtype=F32) for k in range(3)], 1)],
                         axis=1).astype(F32)
    if np.sqrt((res.astype(np.float64) ** 2).sum()) < 1.0e-7:
        return T0, x0, float((res * res).sum())
    w = ls_weights(scheme, sigma, res, p, q)  # neighborhood sees the raw target points (alignment.py:183)
    res = (res * w).astype(F32)
    jac = (jac * w.reshape(-1, 1)).astype(F32)
    ja, ra = jac.astype(accumulate), res.astype(accumulate)
    H = ja.T @ ja
    if abs(np.linalg.det(H)) < 1.0e-7:
        raise RuntimeError("Invalid Jacobian in Gauss Newton minimization")
    dx = -(np.linalg.inv(H) @ (ja.T @ ra))
    params = (x0 + dx.astype(F32)).astype(F32)
    return build_pose_matrix(params), params, float((ra * ra).sum())


def weighted_procrustes(tgt: np.ndarray, ref: np.ndarray, weights: Optional[np.ndarray] = None) -> np.ndarray:
    """reference slam/common/registration.py:15-74 (numpy branch): weighted centroids in the dtype of the points, the
    cross-covariance of the centred clouds in float64 WITHOUT the weights (as there), R = U S V^T, t = mu_ref - R mu_tgt."""
    dt = tgt.dtype
    w = np.ones((tgt.shape[0], 1), dt) if weights is None else np.asarray(weights, dt).reshape(-1, 1)
    aw = w / w.sum(axis=0)
    mu_t = (tgt * aw).sum(axis=0).reshape(1, 3)
    mu_r = (ref * aw).sum(axis=0).reshape(1, 3)
    C = (ref - mu_r).T.astype(np.float64) @ (tgt - mu_t).astype(np.float64)
    U, _, Vt = np.linalg.svd(C)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        S[-1, -1] = -1
    T = np.eye(4)
    T[:3, :3] = U @ S @ Vt
    T[:3, 3] = mu_r.astype(np.float64).reshape(3) - T[:3, :3] @ mu_t.astype(np.float64).reshape(3)
    return T


# ======================================================================================================================
# a10-a12  kd-tree local map
# ======================================================================================================================
def knn_normals(model: np.ndarray, tree: cKDTree, idx: np.ndarray, k: int = 10) -> np.ndarray:
    """reference slam/odometry/local_map.py:397-422 for the map points `idx`: k+1 NN, drop the first (:405-407),
    covariance centred on the query point (:411-413), normal = last right-singular vector (:414-416)."""
    pts = model[idx]
    _, nb = tree.query(pts.astype(np.float64), k=k + 1)
    nb = nb[:, 1:]
    centered = (model[nb.reshape(-1)].reshape(-1, k, 3) - pts.reshape(-1, 1, 3)).astype(F32)
    covs = (centered[:, :, :, None] * centered[:, :, None, :]).mean(axis=1).astype(F32)
    _, _, vh = np.linalg.svd(covs)
    return vh[:, 2, :].astype(F32)


class KdTreeLocalMapOracle:
    """reference slam/odometry/local_map.py:254-427 (`KdTreeLocalMap`)."""

    def __init__(self, local_map_size: int = 20, num_neighbors_normals: int = 10, workers: int = -1):
        self.local_map_size = local_map_size
        self.k = num_neighbors_normals
        self.workers = workers
        self.init()

    def init(self):  # :279-288
        self.local_map: Optional[np.ndarray] = None
        self.num_elements: List[int] = []
        self.model: Optional[np.ndarray] = None
        self.normals: Optional[np.ndarray] = None
        self.tree: Optional[cKDTree] = None

    def set_map_pointcloud(self, pc: np.ndarray):  # :289-299
        self.init()
        self.local_map = np.asarray(pc, dtype=F32)
        self.build_model()

    def build_model(self):  # :365-369  (tree rebuilt and normal cache zeroed EVERY update)
        self.model = self.local_map
        self.normals = np.zeros((self.model.shape[0], 4), dtype=F32)
        self.tree = cKDTree(self.model.astype(np.float64))

    def update(self, rel_pose: np.ndarray, new_pc: Optional[np.ndarray] = None,
               new_vertex_map: Optional[np.ndarray] = None):  # :302-362
        numpy_pc = None
        num = 0
        if new_pc is not None:
            numpy_pc = np.asarray(new_pc, dtype=F32).reshape(-1, 3)
        elif new_vertex_map is not None:
            pts = vertex_map_to_points(new_vertex_map)
            nrm = np.sqrt((pts * pts).sum(axis=1, dtype=F32))
            numpy_pc = pts[nrm > 0.01]  # :324
        if numpy_pc is not None:
            numpy_pc = numpy_pc[~np.isnan(numpy_pc).any(axis=1)]  # :327 remove_nan
            num = numpy_pc.shape[0]
        rel_pose = np.asarray(rel_pose, dtype=F32).reshape(4, 4)
        if self.local_map is None:
            self.local_map = numpy_pc
            self.num_elements.append(num)
        else:
            inv = np.linalg.inv(rel_pose)  # :346 (float32 LAPACK)
            moved = (np.einsum("ij,nj->ni", inv[:3, :3], self.local_map) + inv[:3, 3].reshape(1, 3)).astype(F32)
            if numpy_pc is not None:
                self.local_map = np.concatenate([moved, numpy_pc], axis=0)
                self.num_elements.append(num)
            else:
                self.local_map = moved
            if len(self.num_elements) > self.local_map_size:  # :356-360
                first = self.num_elements.pop(0)
                self.local_map = self.local_map[first:]
        self.build_model()

    def nearest_neighbor_search(self, pts: np.ndarray):  # :372-395
        _, idx = self.tree.query(pts.astype(np.float64), workers=self.workers)
        neighbors = self.model[idx]
        normals = self.get_normals(idx)
        return neighbors, normals, idx

    def get_normals(self, idx: np.ndarray) -> np.ndarray:  # :397-422
        todo = idx[self.normals[idx, 3] == 0.0]
        if todo.shape[0] > 0:
            todo = np.unique(todo)  # the reference recomputes duplicates; results are identical
            self.normals[todo, :3] = knn_normals(self.model, self.tree, todo, self.k)
            self.normals[todo, 3] = 1.0
        return self.normals[idx, :3]


# ======================================================================================================================
# a19  projective local map (the reference's "GPU" variant): normal maps, per-pixel association over K maps
# ======================================================================================================================
def _box_sum(img: np.ndarray, k: int) -> np.ndarray:
    """zero-padded k x k box sum of [C, H, W] (Fnn.conv2d with a ones kernel, geometry.py:262-270)."""
    c, h, w = img.shape
    r = k // 2
    pad = np.zeros((c, h + 2 * r, w + 2 * r), dtype=img.dtype)
    pad[:, r:r + h, r:r + w] = img
    out = np.zeros_like(img)
    for dy in range(k):
        for dx in range(k):
            out += pad[:, dy:dy + h, dx:dx + w]
    return out


def compute_normal_map(vmap: np.ndarray, kernel_size: int = 5, dtype=F32) -> np.ndarray:
    """reference slam/common/geometry.py:240-295: n ~ (sum_box p p^T)^-1 (sum_box p), adjugate inverse (:63-98),
    zero where |det| <= 1e-6 or the pixel is null, normalised.  vmap [3,H,W] f32 -> [3,H,W].
    `dtype` = float32 restates the reference arithmetic (noise-dominated, see tests); float64 is the exact value the
    HIP kernel is compared with."""
    F32 = dtype  # noqa: N806 (shadows the module constant on purpose)
    v = np.asarray(vmap, dtype=F32)
    _, h, w = v.shape
    cov = (v[None, :, :, :] * v[:, None, :, :]).reshape(9, h, w)  # cov[i*3+j] = v_i v_j
    s = _box_sum(v, kernel_size)  # [3,H,W]
    a = _box_sum(cov, kernel_size).reshape(3, 3, h, w)
    A = np.moveaxis(a, (0, 1), (2, 3))  # [H,W,3,3]
    S = np.moveaxis(s, 0, 2)  # [H,W,3]
    adj = np.empty_like(A)
    for i in range(3):
        adj[..., i, :] = np.cross(A[..., i - 2, :], A[..., i - 1, :])  # :74-76
    det = (adj * A).sum(axis=-1).mean(axis=-1)  # :88
    ok = np.abs(det) > 1.0e-6
    safe = np.where(ok, det, F32(1.0))
    inv_t = adj / safe[..., None, None]
    inv_t[~ok] = 0.0
    inv = np.swapaxes(inv_t, -1, -2)  # :109-110
    n = np.einsum("...ij,...j->...i", inv, S).astype(F32)
    norms = np.sqrt((n * n).sum(axis=-1, keepdims=True, dtype=F32))
    n = np.where(norms > 0, n / np.where(norms > 0, norms, F32(1.0)), F32(0.0)).astype(F32)
    null = np.sqrt((v * v).sum(axis=0, dtype=F32)) == 0.0
    n[null] = 0.0
    return np.ascontiguousarray(np.moveaxis(n, 2, 0))


def compute_neighbors(vm_target: np.ndarray, vm_reference: np.ndarray, reference_fields: Optional[np.ndarray] = None):
    """reference slam/common/geometry.py:397-439: per pixel, the closest of the K reference vertex maps (inf where the
    target or the reference pixel is null; ties -> the first).  vm_target [3,H,W], vm_reference [K,3,H,W]."""
    t = np.asarray(vm_target, dtype=F32)
    r = np.asarray(vm_reference, dtype=F32)
    mask_t = np.abs(t).max(axis=0) > 0  # [H,W]
    mask_r = np.abs(r).max(axis=1) > 0  # [K,H,W]
    diff = np.sqrt(((t[None] - r) ** 2).sum(axis=1, dtype=F32))
    diff = np.where(mask_r & mask_t[None], diff, np.inf)
    idx = diff.argmin(axis=0)  # first minimum
    nb = np.take_along_axis(r, idx[None, None], axis=0)[0]
    nb = nb * mask_t[None]
    fields = None
    if reference_fields is not None:
        fields = np.take_along_axis(np.asarray(reference_fields, dtype=F32), idx[None, None], axis=0)[0]
    return nb.astype(F32), fields


class ProjectiveLocalMapOracle:
    """reference slam/odometry/local_map.py:91-240 (`ProjectiveLocalMap`)."""

    def __init__(self, height, width, up_fov, down_fov, local_map_size: int = 20, normals_kernel_size: int = 5,
                 normals_dtype=F32):
        self.h, self.w, self.up, self.down = height, width, up_fov, down_fov
        self.size, self.ks = local_map_size, normals_kernel_size
        self.normals_dtype = normals_dtype
        self.init()

    def init(self):  # :113-119
        self.vmaps: List[np.ndarray] = []
        self.nmaps: List[np.ndarray] = []
        self.masks: List[np.ndarray] = []
        self.poses: List[np.ndarray] = []
        self.model_vmap = None
        self.model_nmap = None

    def update(self, rel_pose: np.ndarray, new_vertex_map: Optional[np.ndarray] = None):  # :122-174
        rel_pose = np.asarray(rel_pose, dtype=F32).reshape(4, 4)
        if new_vertex_map is not None:
            v = np.asarray(new_vertex_map, dtype=F32).reshape(3, self.h, self.w)
            nm = compute_normal_map(v, self.ks, self.normals_dtype).astype(F32)
            mask = np.abs(v).max(axis=0) > 0
        if not self.vmaps:
            self.vmaps, self.nmaps, self.masks, self.poses = [v], [nm], [mask], [rel_pose]
        else:
            inv = np.linalg.inv(rel_pose)
            self.poses = [(inv @ p).astype(F32) for p in self.poses]  # :149
            if new_vertex_map is not None:
                self.poses.append(np.eye(4, dtype=F32))
                self.vmaps.append(v)
                self.nmaps.append(nm)
                self.masks.append(mask)
            if len(self.poses) > self.size:  # :166-171
                self.vmaps, self.nmaps, self.masks, self.poses = self.vmaps[1:], self.nmaps[1:], self.masks[1:], \
                    self.poses[1:]
        self.build_model()

    def build_model(self):  # :177-202
        mv, mn = [], []
        for v, nm, mask, pose in zip(self.vmaps, self.nmaps, self.masks, self.poses):
            pts = apply_transformation(vertex_map_to_points(v), pose)
            nrm = (vertex_map_to_points(nm) @ pose[:3, :3].T).astype(F32)  # apply_rotation, pose.py:154-167
            m = mask.reshape(-1, 1).astype(F32)
            pts, nrm = pts * m, nrm * m
            six = build_projection_map(pts, self.h, self.w, self.up, self.down,
                                       channels=np.concatenate([pts, nrm], axis=1))
            mv.append(six[:3])
            mn.append(six[3:6])
        self.model_vmap = np.stack(mv)
        self.model_nmap = np.stack(mn)

    def nearest_neighbor_search(self, pts: np.ndarray):  # :205-235
        tv = build_projection_map(pts, self.h, self.w, self.up, self.down)
        nb_v, nb_n = compute_neighbors(tv, self.model_vmap, self.model_nmap)
        new_points = vertex_map_to_points(tv)
        nb_points = vertex_map_to_points(nb_v)
        mask = (np.abs(new_points).max(axis=1) > 0) & (np.abs(nb_points).max(axis=1) > 0)
        return nb_points[mask], vertex_map_to_points(nb_n)[mask], new_points[mask]


class ICPProjectiveOracle:
    """`ICPFrameToModel` (icp_odometry.py:72-381) with the projective local map and vertex-map input
    (data_key = "vertex_map", the reference's default): targets = non-null pixels of the scan's own projection."""

    def __init__(self, config: "ICPOracleConfig", normals_dtype=F32):
        self.config = config
        c = config
        self.local_map = ProjectiveLocalMapOracle(c.height, c.width, c.up_fov, c.down_fov, c.local_map_size,
                                                  normals_dtype=normals_dtype)
        self.init()

    def init(self):
        self.local_map.init()
        self.relative_poses: List[np.ndarray] = []
        self._iter = 0
        self._delta = np.eye(4, dtype=F32)
        self.traces: List["FrameTrace"] = []

    def process_next_frame(self, vmap: np.ndarray, init_rpose: Optional[np.ndarray] = None):
        c = self.config
        vmap = np.asarray(vmap, dtype=F32).reshape(3, c.height, c.width)
        if self._iter == 0:
            self.local_map.update(np.eye(4, dtype=F32), vmap)
            self.relative_poses.append(np.eye(4, dtype=F32))
            self._iter += 1
            return None
        pts = vertex_map_to_points(vmap)
        target = pts[np.sqrt((pts * pts).sum(axis=1, dtype=F32)) > 0.0]  # sample_points, :303-305
        pose = np.eye(4, dtype=F32) if init_rpose is None else np.asarray(init_rpose).astype(F32)
        params = np.zeros(6, dtype=F32)
        trace = FrameTrace()
        for _ in range(c.max_num_alignments):
            p = apply_transformation(target, pose)
            q, n, t = self.local_map.nearest_neighbor_search(p)
            step = gauss_newton_step(t, q, n, c.scheme, c.sigma, c.accumulate)
            trace.dx.append(step.dx)
            trace.loss.append(step.loss)
            if np.sqrt((step.dx.astype(F32) ** 2).sum(dtype=F32)) < c.threshold_delta_pose:
                break
            params = from_pose_matrix((build_pose_matrix(step.dx) @ pose).astype(F32))
            pose = build_pose_matrix(params)
        trace.params = params
        self.traces.append(trace)
        new_delta = (self._delta @ pose).astype(F32)
        dp = from_pose_matrix(new_delta)
        if np.linalg.norm(dp[:3]) > c.threshold_trans or np.linalg.norm(dp[3:]) * 180 / np.pi > c.threshold_rot:
            self.local_map.update(pose, vmap)
            self._delta = np.eye(4, dtype=F32)
        else:
            self.local_map.update(pose)
            self._delta = new_delta
        self.relative_poses.append(pose)
        self._iter += 1
        return pose


# ======================================================================================================================
# a7, a8, a17, a18  the frame-to-model ICP driver
# ======================================================================================================================
@dataclass
class ICPOracleConfig:
    """Mirrors ICPFrameToModelConfig (icp_odometry.py:29-64) + the sub-configs that matter numerically."""
    max_num_alignments: int = 100
    threshold_delta_pose: float = 1.0e-4
    threshold_trans: float = 0.1
    threshold_rot: float = 0.3
    local_map_size: int = 20
    num_neighbors_normals: int = 10
    scheme: str = "default"  # ConfigStore default of alignment = plain least squares (alignment.py:77)
    sigma: float = 0.5
    height: int = 64
    width: int = 1024
    up_fov: float = 3.0
    down_fov: float = -24.0
    accumulate: type = F32
    # "point_to_plane" | "point_to_point": the RIGID_ALIGNMENT mode (alignment.py:200-208); point to point = one
    # `GaussNewtonPointToPointAlignment.align` step from x0 = 0 per iteration (:143-189), no normals
    alignment: str = "point_to_plane"


@dataclass
class FrameTrace:
    dx: List[np.ndarray] = field(default_factory=list)
    loss: List[float] = field(default_factory=list)
    params: Optional[np.ndarray] = None


class ICPFrameToModelOracle:
    """reference slam/odometry/icp_odometry.py:72-381 (`ICPFrameToModel`) with the kd-tree local map."""

    def __init__(self, config: ICPOracleConfig):
        self.config = config
        self.local_map = KdTreeLocalMapOracle(config.local_map_size, config.num_neighbors_normals)
        self.init()

    def init(self):  # :128-145
        self.relative_poses: List[np.ndarray] = []
        self.absolute_poses: List[np.ndarray] = []
        self.local_map.init()
        self._iter = 0
        self._sample_pointcloud = False
        self._delta = np.eye(4, dtype=F32)
        self.traces: List[FrameTrace] = []

    # a7 ---------------------------------------------------------------------------------------------------------------
    def _read_input(self, data, is_numpy: bool):  # :319-358
        c = self.config
        if data.ndim == 2:
            if is_numpy:
                self._sample_pointcloud = True  # :330 sticky
            pc = np.asarray(data, dtype=F32)
            vmap = build_projection_map(pc, c.height, c.width, c.up_fov, c.down_fov)  # :333 / :349
        else:
            vmap = np.asarray(data, dtype=F32).reshape(3, c.height, c.width)
            pc = vertex_map_to_points(vmap)
            pc = pc[np.abs(pc).max(axis=1) > 0]  # :343-344 mask_not_null
        nan_px = np.isnan(vmap).any(axis=0)
        vmap = vmap.copy()
        vmap[:, nan_px] = 0.0  # :356 modify_nan_pmap
        pc = pc[~np.isnan(pc).any(axis=1)]  # :357 remove_nan
        self._tgt_vmap, self._tgt_pc = vmap, pc

    # a8 ---------------------------------------------------------------------------------------------------------------
    def sample_points(self) -> np.ndarray:  # :301-308
        if not self._sample_pointcloud:
            pts = vertex_map_to_points(self._tgt_vmap)
            nrm = np.sqrt((pts * pts).sum(axis=1, dtype=F32))
            return pts[nrm > 0.0]
        return self._tgt_pc

    # a17 --------------------------------------------------------------------------------------------------------------
    def register_new_frame(self, target: np.ndarray, init: np.ndarray):  # :248-299
        c = self.config
        pose = init.astype(F32)
        params = np.zeros(6, dtype=F32)
        trace = FrameTrace()
        for _ in range(c.max_num_alignments):
            p = apply_transformation(target, pose)  # :275
            q, n, _ = self.local_map.nearest_neighbor_search(p)  # :278
            if c.alignment == "point_to_point":
                _, dx, loss = point_to_point_step(p, q, None, c.scheme, c.sigma, c.accumulate)
                step = GNStep(dx, loss, None, None, False)
            else:
                step = gauss_newton_step(p, q, n, c.scheme, c.sigma, c.accumulate)  # :284-287
            trace.dx.append(step.dx)
            trace.loss.append(step.loss)
            if np.sqrt((step.dx.astype(F32) ** 2).sum(dtype=F32)) < c.threshold_delta_pose:  # :292
                break
            delta = build_pose_matrix(step.dx)
            params = from_pose_matrix((delta @ pose).astype(F32))  # :296
            pose = build_pose_matrix(params)  # :297
        trace.params = params
        self.traces.append(trace)
        return params, pose

    # a18 --------------------------------------------------------------------------------------------------------------
    def process_next_frame(self, data, init_rpose: Optional[np.ndarray] = None, is_numpy: bool = True):  # :157-246
        """Returns the relative pose [4,4] f32 (None for frame 0, which writes nothing: :171-181)."""
        self._read_input(data, is_numpy)
        if self._iter == 0:
            eye = np.eye(4, dtype=F32)
            self.local_map.update(eye, new_vertex_map=self._tgt_vmap)  # :176
            self.relative_poses.append(eye)
            self.absolute_poses.append(np.eye(4))
            self._iter += 1
            return None
        init = np.eye(4, dtype=F32) if init_rpose is None else np.asarray(init_rpose).astype(F32)  # :147-154
        params, pose = self.register_new_frame(self.sample_points(), init)
        self._update_map(pose)
        self.relative_poses.append(pose)
        self.absolute_poses.append(self.absolute_poses[-1] @ build_pose_matrix(params.astype(np.float64), np.float64))
        self._iter += 1
        return pose

    def _update_map(self, new_rpose: np.ndarray):  # :360-380
        c = self.config
        new_delta = (self._delta @ new_rpose).astype(F32)
        dp = from_pose_matrix(new_delta)
        if np.linalg.norm(dp[:3]) > c.threshold_trans or np.linalg.norm(dp[3:]) * 180 / np.pi > c.threshold_rot:
            self.local_map.update(new_rpose, new_pc=self._tgt_pc)
            self._delta = np.eye(4, dtype=F32)
        else:
            self.local_map.update(new_rpose)
            self._delta = new_delta

    def get_relative_poses(self) -> np.ndarray:  # :310-314
        return np.stack(self.relative_poses, axis=0)


# ======================================================================================================================
# a20  constant-velocity initialisation
# ======================================================================================================================
class ConstantVelocityOracle:
    """reference slam/initialization.py:103-119: the initial guess is the last estimated relative pose."""

    def __init__(self):
        self.last = None

    def next_initial_pose(self):
        return self.last

    def save_real_motion(self, pose):
        self.last = pose


# ======================================================================================================================
# exact brute-force nearest neighbour (small cases; independent of cKDTree)
# ======================================================================================================================
def brute_force_nn(queries: np.ndarray, model: np.ndarray, chunk: int = 2048) -> Tuple[np.ndarray, np.ndarray]:
    q = queries.astype(np.float64)
    m = model.astype(np.float64)
    idx = np.empty(q.shape[0], dtype=np.int64)
    d2 = np.empty(q.shape[0], dtype=np.float64)
    for s in range(0, q.shape[0], chunk):
        d = ((q[s:s + chunk, None, :] - m[None, :, :]) ** 2).sum(axis=2)
        idx[s:s + chunk] = d.argmin(axis=1)
        d2[s:s + chunk] = d.min(axis=1)
    return idx, d2


def pose_error(a: np.ndarray, b: np.ndarray) -> Tuple[float, float]:
    """(translation error in m, rotation angle of Ra^T Rb in rad) — the parity metric of BASELINE.json.

    The angle is atan2(|vee(R - R^T)| / 2, (tr R - 1) / 2): well conditioned near zero (arccos of the trace is not)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    dt = float(np.linalg.norm(a[:3, 3] - b[:3, 3]))
    r = a[:3, :3].T @ b[:3, :3]
    v = 0.5 * np.array([r[2, 1] - r[1, 2], r[0, 2] - r[2, 0], r[1, 0] - r[0, 1]])
    ang = float(np.arctan2(np.linalg.norm(v), (np.trace(r) - 1.0) / 2.0))
    return dt, ang
