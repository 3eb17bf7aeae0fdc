from .config import Config, ConfigDict
from .registry import Registry, build_from_cfg

__all__ = ['Config', 'ConfigDict', 'Registry', 'build_from_cfg']
