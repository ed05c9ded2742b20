"""Golden vectors for the projective local map path, produced by the reference's own
`compute_normal_map` / `compute_neighbors` / `ProjectiveLocalMap` / `ICPFrameToModel` (imported from /root/reference
through oracle/shims; torch pinned to one thread).  TEST INFRASTRUCTURE.

    python oracle/make_golden_projective.py      # writes tests/golden/projective.npz
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
from slam.common.geometry import compute_neighbors, compute_normal_map  # noqa: E402
from slam.common.pose import Pose  # noqa: E402
from slam.common.projection import SphericalProjector  # noqa: E402
from slam.odometry.alignment import GaussNewtonPointToPlaneConfig  # noqa: E402
from slam.odometry.icp_odometry import ICPFrameToModel, ICPFrameToModelConfig  # noqa: E402
from slam.odometry.local_map import ProjectiveLocalMapConfig  # noqa: E402

from pylidar_slam_amd.synthetic import SceneConfig, make_sequence  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def main():
    h, w, n = 32, 256, 6
    scans, gt = make_sequence(SceneConfig(height=h, width=w), n)
    proj = SphericalProjector(h, w, 3, 3.0, -24.0)
    vmaps = [proj.build_projection_map(torch.from_numpy(s).unsqueeze(0))[0].numpy() for s in scans]
    out = dict(hw=np.array([h, w]), vmaps=np.stack(vmaps), gt=gt)
    # components
    out["nmap0"] = compute_normal_map(torch.from_numpy(vmaps[0]).unsqueeze(0), kernel_size=5)[0].numpy()
    rng = np.random.default_rng(3)
    tgt = torch.from_numpy(vmaps[1]).unsqueeze(0)
    ref = torch.from_numpy(np.stack([vmaps[0], vmaps[2], vmaps[3]]))
    fld = torch.from_numpy(rng.normal(size=(3, 3, h, w)).astype(np.float32))
    nb, nf = compute_neighbors(tgt, ref, reference_fields=fld)
    out.update(cn_fields=fld.numpy(), cn_neighbors=nb[0].numpy(), cn_neighbor_fields=nf[0].numpy())
    # full odometry with the projective map, vertex-map input (the reference's default data_key), forced iterations
    for name, scheme, sigma, iters, thr in (("ls", "default", 0.5, 12, 0.0), ("nbh", "neighborhood", 0.2, 15, 1e-4)):
        cfg = ICPFrameToModelConfig(
            max_num_alignments=iters, threshold_delta_pose=thr, data_key="vertex_map",
            local_map=ProjectiveLocalMapConfig(local_map_size=4),
            alignment=GaussNewtonPointToPlaneConfig(gauss_newton_config=dict(max_iters=1, scheme=scheme, sigma=sigma)))
        odo = ICPFrameToModel(cfg, projector=proj, pose=Pose("euler"), device=torch.device("cpu"))
        odo.init()
        rel, its = [], []
        orig = odo.rigid_alignment.align
        count = [0]

        def align(a, b, c, **kw):
            count[0] += 1
            return orig(a, b, c, **kw)
        odo.rigid_alignment.align = align
        last = None
        for f in range(n):
            d = {"vertex_map": torch.from_numpy(vmaps[f]), "init_rpose": last}
            count[0] = 0
            odo.process_next_frame(d)
            its.append(count[0])
            if "odometry_pose" in d:
                rel.append(d["odometry_pose"].copy())
                last = d["odometry_pose"].astype(np.float64)
            else:
                rel.append(np.eye(4, dtype=np.float32))
        out[f"{name}_rel"] = np.stack(rel)
        out[f"{name}_iters"] = np.array(its)
        out[f"{name}_cfg"] = np.array([scheme, str(sigma), str(iters), str(thr)])
        if name == "ls":
            out["ls_model_vmap"] = odo.local_map._model_vmap.numpy()
            out["ls_model_nmap"] = odo.local_map._model_nmap.numpy()
        err = [np.linalg.norm((np.linalg.inv(gt[f - 1]) @ gt[f])[:3, 3] - rel[f][:3, 3]) for f in range(1, n)]
        print(name, "iters", its, "max |t - t_gt|", max(err))
    np.savez_compressed(os.path.join(OUT, "projective.npz"), **out)
    print("projective.npz", os.path.getsize(os.path.join(OUT, "projective.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
