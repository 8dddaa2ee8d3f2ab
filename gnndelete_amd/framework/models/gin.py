"""Import-path compatibility with the reference (framework/models/gin.py)."""
from .backbones import GIN  # noqa: F401
