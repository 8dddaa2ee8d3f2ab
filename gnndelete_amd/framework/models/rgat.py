"""Import-path compatibility with the reference (framework/models/rgat.py)."""
from ...nn import RGATConv  # noqa: F401
from .backbones import RGAT  # noqa: F401
