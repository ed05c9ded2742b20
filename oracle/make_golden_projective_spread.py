"""The reference's OWN numerical spread on the projective local-map path.  TEST INFRASTRUCTURE.

`compute_normal_map` (slam/common/geometry.py:240-295) box-filters p and p p^T in float32 and inverts the 3x3 sum by
its adjugate.  With coordinates of ~10-20 m the 25-term float32 sums of p p^T carry an absolute rounding error of
~1e-3 m^2 while the quantity the plane fit lives on (the thickness variance of a 1 cm-noise surface patch) is ~1e-4 m^2:
the reference's normals — and through them its poses — depend on the ORDER in which its own conv2d adds the 25 taps.

This script measures that dependence with the reference's code only: the full `ICPFrameToModel` + `ProjectiveLocalMap`
sequence of make_golden_projective.py is run again with `compute_normal_map`'s two `conv2d` calls evaluated on a
mirrored image and mirrored back (the all-ones kernel is symmetric, so in exact arithmetic nothing changes; in float32
the taps are added in the opposite order along the mirrored axis).  Variants: baseline (= tests/golden/projective.npz),
mirrored along W, along H, along both, and — as the anchor the variants scatter around — the same two convolutions
carried out in float64.  Stored: the relative poses of every variant, for both runs of the projective golden.

    python oracle/make_golden_projective_spread.py    # writes tests/golden/projective_spread.npz

tests/test_gpu_parity.py::test_projective_icp_sequence asserts that the HIP path (float64 window sums) deviates from
the reference's baseline by no more than the reference's mirrored runs deviate from it.
"""
import logging
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "oracle", "shims"), "/root/reference", os.path.join(ROOT, "pylidar-slam_amd")]
logging.disable(logging.WARNING)

import numpy as np  # noqa: E402
import torch  # noqa: E402

torch.set_num_threads(1)
import slam.common.geometry as geometry  # noqa: E402
from slam.common.pose import Pose  # noqa: E402
from slam.common.projection import SphericalProjector  # noqa: E402
from slam.odometry.alignment import GaussNewtonPointToPlaneConfig  # noqa: E402
from slam.odometry.icp_odometry import ICPFrameToModel, ICPFrameToModelConfig  # noqa: E402
from slam.odometry.local_map import ProjectiveLocalMapConfig  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
VARIANTS = ("baseline", "mirror_w", "mirror_h", "mirror_hw", "float64")


class _ConvVariant:
    """Stands in for the `Fnn` module inside slam.common.geometry: conv2d on a mirrored image / in float64."""

    def __init__(self, real, variant):
        self._real, self._variant = real, variant

    def __getattr__(self, name):
        return getattr(self._real, name)

    def conv2d(self, x, k, **kw):
        v = self._variant
        if v == "baseline":
            return self._real.conv2d(x, k, **kw)
        if v == "float64":
            return self._real.conv2d(x.double(), k.double(), **kw).float()
        dims = {"mirror_w": (-1,), "mirror_h": (-2,), "mirror_hw": (-2, -1)}[v]
        return self._real.conv2d(x.flip(dims).contiguous(), k, **kw).flip(dims).contiguous()


def run(vmaps, h, w, scheme, sigma, iters, thr):
    proj = SphericalProjector(h, w, 3, 3.0, -24.0)
    cfg = ICPFrameToModelConfig(
        max_num_alignments=iters, threshold_delta_pose=thr, data_key="vertex_map",
        local_map=ProjectiveLocalMapConfig(local_map_size=4),
        alignment=GaussNewtonPointToPlaneConfig(gauss_newton_config=dict(max_iters=1, scheme=scheme, sigma=sigma)))
    odo = ICPFrameToModel(cfg, projector=proj, pose=Pose("euler"), device=torch.device("cpu"))
    odo.init()
    rel, last = [], None
    for vm in vmaps:
        d = {"vertex_map": torch.from_numpy(vm), "init_rpose": last}
        odo.process_next_frame(d)
        if "odometry_pose" in d:
            rel.append(d["odometry_pose"].copy())
            last = d["odometry_pose"].astype(np.float64)
        else:
            rel.append(np.eye(4, dtype=np.float32))
    return np.stack(rel)


def pose_error(a, b):
    d = np.linalg.inv(a.astype(np.float64)) @ b.astype(np.float64)
    c = np.clip((np.trace(d[:3, :3]) - 1.0) / 2.0, -1.0, 1.0)
    return float(np.linalg.norm(a[:3, 3].astype(np.float64) - b[:3, 3].astype(np.float64))), float(np.arccos(c))


def main():
    g = np.load(os.path.join(OUT, "projective.npz"))
    h, w = (int(v) for v in g["hw"])
    vmaps = g["vmaps"]
    real = geometry.Fnn
    out = dict(variants=np.array(VARIANTS))
    for name in ("ls", "nbh"):
        scheme, sigma, iters, thr = (str(v) for v in g[f"{name}_cfg"])
        poses = {}
        for v in VARIANTS:
            geometry.Fnn = _ConvVariant(real, v)
            try:
                poses[v] = run(vmaps, h, w, scheme, float(sigma), int(iters), float(thr))
            finally:
                geometry.Fnn = real
            out[f"{name}_{v}_rel"] = poses[v]
        assert np.array_equal(poses["baseline"], g[f"{name}_rel"]), "baseline must reproduce projective.npz"
        for v in VARIANTS[1:]:
            errs = [pose_error(poses["baseline"][f], poses[v][f]) for f in range(1, len(vmaps))]
            print(f"{name} {v:10s} vs baseline: max |dt| = {max(e[0] for e in errs):.2e} m, "
                  f"max |dr| = {max(e[1] for e in errs):.2e} rad")
    np.savez_compressed(os.path.join(OUT, "projective_spread.npz"), **out)
    print("projective_spread.npz", os.path.getsize(os.path.join(OUT, "projective_spread.npz")), "bytes")


if __name__ == "__main__":
    main()
