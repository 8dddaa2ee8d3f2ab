"""Host-side mirror of the reference's ``framework`` package for the Del hot path."""
from .models import GAT, GCN, GIN, RGCN, GATDelete, GCNDelete, GINDelete, RGCNDelete  # noqa: F401
