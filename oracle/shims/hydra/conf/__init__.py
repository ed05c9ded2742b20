from dataclasses import dataclass, field  # noqa: F401
from omegaconf import MISSING  # noqa: F401
from hydra.core.config_store import ConfigStore  # noqa: F401
