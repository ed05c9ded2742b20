"""Registers the MI355X plugins inside an importable pyLiDAR-SLAM checkout, without editing it.

The reference resolves its plugins through enums of `(class, config dataclass)` pairs:
  ODOMETRY         selector `algorithm`    slam/odometry/__init__.py:23-32   (loader slam/common/utils.py:266-302)
  DATASET          selector `dataset`      slam/dataset/__init__.py:15-38
  FILTER           selector `filter_name`  slam/preprocessing.py:230-252
  LOCAL_MAP        selector `type`         slam/odometry/local_map.py:437-445   (inner seam of ICPFrameToModel)
  RIGID_ALIGNMENT  selector `mode`         slam/odometry/alignment.py:200-208   (inner seam of ICPFrameToModel)
and hydra's ConfigStore (slam/odometry/icp_odometry.py:67-68).  An Enum cannot be extended in place, so
`register_with_reference()` builds new enums with the same members plus the MI355X ones and swaps them in wherever
the reference imported them.  A maintainer would instead add the lines shown in INTEGRATION.md to those modules.
"""
import sys
from enum import Enum

ALGORITHM_NAME = "icp_F2M_mi355x"
DATASET_NAMES = ("synthetic_mi355x", "kitti_mi355x")
FILTER_NAMES = ("grid_sample_mi355x", "distortion_mi355x", "voxelization_mi355x", "to_device_mi355x", "to_tensor_mi355x")
LOCAL_MAP_NAMES = ("hashgrid_local_map_mi355x", "projective_local_map_mi355x")
ALIGNMENT_NAMES = ("point_to_plane_gauss_newton_mi355x", "point_to_point_gauss_newton_mi355x")


def _swap_enum(owner, attr: str, extra: dict, mixins: tuple, namespace: dict):
    """New enum = members of `owner.attr` + `extra`, installed in every loaded slam.* module that holds the old one."""
    current = getattr(owner, attr)
    if all(name in current.__members__ for name in extra):
        return current
    members = {m.name: m.value for m in current}
    members.update(extra)
    base = type(f"_{attr}Base", mixins, dict(namespace)) if (mixins or namespace) else None
    patched = Enum(attr, members, type=base) if base is not None else Enum(attr, members)
    patched.__doc__ = current.__doc__
    for name, mod in list(sys.modules.items()):
        if name.startswith("slam") and getattr(mod, attr, None) is current:
            setattr(mod, attr, patched)
    return patched


def register_with_reference():
    """Call once, before `SLAM.init()` / `run.py`'s hydra main builds the pipeline. Returns the patched ODOMETRY enum
    (`icp_F2M_mi355x`); DATASET gains `synthetic_mi355x` / `kitti_mi355x`, FILTER gains `grid_sample_mi355x` /
    `distortion_mi355x` / `voxelization_mi355x` / `to_device_mi355x` / `to_tensor_mi355x`, LOCAL_MAP gains
    `hashgrid_local_map_mi355x` / `projective_local_map_mi355x` and RIGID_ALIGNMENT gains
    `point_to_plane_gauss_newton_mi355x` / `point_to_point_gauss_newton_mi355x` (so the reference's OWN
    `ICPFrameToModel` can run on the MI355X local map / alignment through its inner seams)."""
    import slam.dataset as ref_dataset
    import slam.odometry as ref_odometry
    import slam.preprocessing as ref_pre
    from slam.common.utils import ObjectLoaderEnum

    from .dataset import KITTIConfig, KITTIDatasetLoader, SyntheticDatasetConfig, SyntheticDatasetLoader
    import slam.odometry.alignment as ref_alignment
    import slam.odometry.local_map as ref_local_map
    from .odometry import (Distortion, DistortionConfig, GridSample, GridSampleConfig, HashGridLocalMap,
                           HashGridLocalMapConfig, MI355XICPConfig, MI355XICPFrameToModel, PointToPlaneAlignment,
                           PointToPlaneAlignmentConfig, PointToPointAlignment, PointToPointAlignmentConfig,
                           ProjectiveLocalMap, ProjectiveLocalMapConfig, ToDevice, ToDeviceConfig, ToTensor,
                           ToTensorConfig, Voxelization, VoxelizationConfig)

    odometry = _swap_enum(ref_odometry, "ODOMETRY", {ALGORITHM_NAME: (MI355XICPFrameToModel, MI355XICPConfig)},
                          (ObjectLoaderEnum,), {"type_name": classmethod(lambda cls: "algorithm")})
    _swap_enum(ref_dataset, "DATASET", {DATASET_NAMES[0]: (SyntheticDatasetLoader, SyntheticDatasetConfig),
                                         DATASET_NAMES[1]: (KITTIDatasetLoader, KITTIConfig)},
               (ObjectLoaderEnum,), {"type_name": classmethod(lambda cls: "dataset")})

    def _load_filter(config, **kwargs):  # slam/preprocessing.py:243-252, against the patched enum
        name = config.filter_name
        flt = ref_pre.FILTER
        assert name in flt.__members__, f"unknown filter {name}"
        _class, _config = flt[name].value
        return _class(_config(**config), **kwargs)

    _swap_enum(ref_pre, "FILTER", {FILTER_NAMES[0]: (GridSample, GridSampleConfig),
                                   FILTER_NAMES[1]: (Distortion, DistortionConfig),
                                   FILTER_NAMES[2]: (Voxelization, VoxelizationConfig),
                                   FILTER_NAMES[3]: (ToDevice, ToDeviceConfig),
                                   FILTER_NAMES[4]: (ToTensor, ToTensorConfig)},
               (), {"load": staticmethod(_load_filter)})
    _swap_enum(ref_local_map, "LOCAL_MAP", {LOCAL_MAP_NAMES[0]: (HashGridLocalMap, HashGridLocalMapConfig),
                                            LOCAL_MAP_NAMES[1]: (ProjectiveLocalMap, ProjectiveLocalMapConfig)},
               (ObjectLoaderEnum,), {"type_name": classmethod(lambda cls: "type")})
    _swap_enum(ref_alignment, "RIGID_ALIGNMENT",
               {ALIGNMENT_NAMES[0]: (PointToPlaneAlignment, PointToPlaneAlignmentConfig),
                ALIGNMENT_NAMES[1]: (PointToPointAlignment, PointToPointAlignmentConfig)},
               (ObjectLoaderEnum,), {"type_name": classmethod(lambda cls: "mode")})
    try:  # hydra group entries, so `slam/odometry=icp_odometry_mi355x` / `dataset=synthetic_mi355x` resolve
        from hydra.core.config_store import ConfigStore
        cs = ConfigStore.instance()
        cs.store(name="icp_odometry_mi355x", group="slam/odometry", node=MI355XICPConfig)
        cs.store(name=DATASET_NAMES[0], group="dataset", node=SyntheticDatasetConfig)
        cs.store(name=DATASET_NAMES[1], group="dataset", node=KITTIConfig)
        cs.store(name="hashgrid_mi355x", group="slam/odometry/local_map", node=HashGridLocalMapConfig)
        cs.store(name="projective_mi355x", group="slam/odometry/local_map", node=ProjectiveLocalMapConfig)
        cs.store(name="point_to_plane_GN_mi355x", group="slam/odometry/alignment", node=PointToPlaneAlignmentConfig)
        cs.store(name="point_to_point_GN_mi355x", group="slam/odometry/alignment", node=PointToPointAlignmentConfig)
    except Exception:  # hydra absent: the enum patches are all the loaders need
        pass
    return odometry
