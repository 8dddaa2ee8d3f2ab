from .backbones import GAT, GCN, GIN, RGAT, RGCN, SAGE
from .deletion import (DeletionLayer, DeletionLayerKG, GATDelete, GCNDelete, GINDelete, RGATDelete, RGCNDelete,
                       SAGEDelete)

__all__ = ['GCN', 'GAT', 'GIN', 'RGCN', 'RGAT', 'DeletionLayer', 'DeletionLayerKG', 'GCNDelete', 'GATDelete',
           'GINDelete', 'RGCNDelete', 'RGATDelete', 'SAGE', 'SAGEDelete']
