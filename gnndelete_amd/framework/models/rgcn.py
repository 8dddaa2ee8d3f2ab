"""Import-path compatibility with the reference (framework/models/rgcn.py)."""
from .backbones import RGCN  # noqa: F401
