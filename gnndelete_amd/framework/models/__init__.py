from .backbones import GAT, GCN, GIN, RGCN
from .deletion import DeletionLayer, DeletionLayerKG, GATDelete, GCNDelete, GINDelete, RGCNDelete

__all__ = ['GCN', 'GAT', 'GIN', 'RGCN', 'DeletionLayer', 'DeletionLayerKG', 'GCNDelete', 'GATDelete',
           'GINDelete', 'RGCNDelete']
