"""Minimal stand-in for `omegaconf` — TEST INFRASTRUCTURE ONLY (oracle harness).

Lets the reference's own modules import in this container, where omegaconf is not installed.
Only the surface the ICP hot path touches is provided.
"""
import dataclasses

MISSING = "???"


class DictConfig(dict):
    """dict with attribute access, enough for ObjectLoaderEnum.load (reference slam/common/utils.py:271-298)."""

    def __getattr__(self, item):
        try:
            return self[item]
        except KeyError as e:
            raise AttributeError(item) from e

    def __setattr__(self, key, value):
        self[key] = value


class ListConfig(list):
    pass


class OmegaConf:
    @staticmethod
    def create(obj=None):
        if obj is None:
            return DictConfig()
        if dataclasses.is_dataclass(obj):
            obj = dataclasses.asdict(obj() if isinstance(obj, type) else obj)
        if isinstance(obj, dict):
            return DictConfig({k: OmegaConf.create(v) if isinstance(v, dict) else v for k, v in obj.items()})
        return obj

    @staticmethod
    def get_type(obj):
        return type(obj)

    @staticmethod
    def to_yaml(obj):
        import yaml
        if dataclasses.is_dataclass(obj):
            obj = dataclasses.asdict(obj)
        return yaml.safe_dump(dict(obj))

    @staticmethod
    def load(path):
        import yaml
        with open(path) as f:
            return OmegaConf.create(yaml.safe_load(f))

    @staticmethod
    def to_container(obj, **kwargs):
        return dict(obj)

    @staticmethod
    def structured(obj):
        return obj() if isinstance(obj, type) else obj
