"""Registers the MI355X odometry inside an importable pyLiDAR-SLAM checkout, without editing it.

The reference resolves `slam.odometry.algorithm` through the `ODOMETRY` enum of `(class, config dataclass)` pairs
(slam/odometry/__init__.py:23-32, loader slam/common/utils.py:266-302) and hydra's ConfigStore
(slam/odometry/icp_odometry.py:67-68).  An Enum cannot be extended in place, so `register_with_reference()` builds
a new enum with the same members plus `icp_F2M_mi355x` and swaps it in wherever the reference imported it.
A maintainer would instead add the two lines shown in INTEGRATION.md to slam/odometry/__init__.py.
"""
from enum import Enum

ALGORITHM_NAME = "icp_F2M_mi355x"


def register_with_reference():
    """Call once, before `SLAM.init()` / `run.py`'s hydra main builds the odometry. Returns the patched enum."""
    import slam.odometry as ref_odometry
    from slam.common.utils import ObjectLoaderEnum
    from .odometry import MI355XICPConfig, MI355XICPFrameToModel

    current = ref_odometry.ODOMETRY
    if ALGORITHM_NAME in current.__members__:
        return current
    members = {m.name: m.value for m in current}
    members[ALGORITHM_NAME] = (MI355XICPFrameToModel, MI355XICPConfig)

    class _Base(ObjectLoaderEnum):
        @classmethod
        def type_name(cls):
            return "algorithm"

    patched = Enum("ODOMETRY", members, type=_Base)
    patched.__doc__ = current.__doc__
    ref_odometry.ODOMETRY = patched
    import sys
    for name, mod in list(sys.modules.items()):
        if name.startswith("slam.") and getattr(mod, "ODOMETRY", None) is current:
            setattr(mod, "ODOMETRY", patched)
    try:  # hydra group entry, so `slam/odometry=icp_odometry_mi355x` resolves
        from hydra.core.config_store import ConfigStore
        ConfigStore.instance().store(name="icp_odometry_mi355x", group="slam/odometry", node=MI355XICPConfig)
    except Exception:  # hydra absent: the enum patch is all `ODOMETRY.load` needs
        pass
    return patched
