"""Import-path compatibility with the reference (framework/models/gat.py)."""
from .backbones import GAT  # noqa: F401
