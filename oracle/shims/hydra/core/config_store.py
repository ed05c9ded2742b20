"""ConfigStore stand-in: stores *instantiated* dataclass nodes; `load("<group>/<name>.yaml").node` returns them."""
import copy
import dataclasses


class _Node:
    def __init__(self, node):
        self.node = node


class ConfigStore:
    _instance = None

    def __init__(self):
        self.repo = {}

    @staticmethod
    def instance():
        if ConfigStore._instance is None:
            ConfigStore._instance = ConfigStore()
        return ConfigStore._instance

    def store(self, name, node, group=None, package=None, **kwargs):
        key = f"{group}/{name}.yaml" if group else f"{name}.yaml"
        self.repo[key] = node

    def load(self, config_path):
        node = self.repo.get(config_path)
        if node is None:
            return None
        if isinstance(node, type) and dataclasses.is_dataclass(node):
            node = node()
        else:
            node = copy.deepcopy(node)
        return _Node(node)
