"""numba stand-in (TEST INFRASTRUCTURE ONLY): @njit runs the plain Python body; prange = range.

The reference's kernels (slam/common/pointcloud.py:13-153) are fully specified by their Python source, so running
them un-jitted gives the same integers. NumPy 2 removed `np.round_` (pointcloud.py:73-75 uses it): alias it back.
"""
import numpy as np

if not hasattr(np, "round_"):
    np.round_ = np.round
prange = range


def njit(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]

    def deco(fn):
        return fn
    return deco


jit = njit
