"""MI355X-native frame-to-model ICP odometry hot path of pyLiDAR-SLAM (gfx950, HIP).

The heavy lifting lives in `_lib/libicp_mi355x.so` (C ABI: include/icp_mi355x.h, sources: pylidar-slam_amd/csrc);
this package is the Python host side mirroring the reference's plugin surface.  Importing it does not touch the GPU;
creating any context / odometry object does, and raises if the library or the device is missing (no CPU fallback).
"""
from ._lib import IcpLibraryError, library_path, load_library  # noqa: F401
from . import synthetic  # noqa: F401

__version__ = "0.1.0"


def __getattr__(name):
    # lazy: `engine` / `odometry` import torch
    if name in ("IcpContext", "InvalidJacobianError", "RegisterResult"):
        from . import engine
        return getattr(engine, name)
    if name in ("MI355XICPFrameToModel", "MI355XICPConfig", "HashGridLocalMap", "PointToPlaneAlignment",
                "SphericalProjector", "GridSample", "GridSampleConfig", "grid_sample",
                "ConstantVelocityInitialization", "OdometryAlgorithm"):
        from . import odometry
        return getattr(odometry, name)
    raise AttributeError(name)
