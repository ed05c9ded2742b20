"""Writes /tmp/probe.bin for tools/search_probe.hip: the C2 map + a converged-pose scan (dev tool)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pylidar-slam_amd"))
from pylidar_slam_amd.synthetic import SceneConfig, make_fixed_map, make_sequence
cfg = SceneConfig(height=64, width=2048)
scans, poses = make_sequence(cfg, 8)
model = make_fixed_map(cfg, scans, poses, 0)
rel = np.linalg.inv(poses[0]) @ poses[1]
q = (scans[1].astype(np.float64) @ rel[:3, :3].T + rel[:3, 3]).astype(np.float32)
with open("/tmp/probe.bin", "wb") as f:
    np.array([model.shape[0], q.shape[0]], np.int32).tofile(f)
    np.array([0.33], np.float32).tofile(f)
    model.tofile(f); q.tofile(f)
print("wrote /tmp/probe.bin", model.shape, q.shape)
