"""Import-path compatibility with the reference (framework/models/gcn.py)."""
from .backbones import GCN  # noqa: F401
