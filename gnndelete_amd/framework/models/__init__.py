from .backbones import GAT, GCN, GIN, RGCN, SAGE
from .deletion import DeletionLayer, DeletionLayerKG, GATDelete, GCNDelete, GINDelete, RGCNDelete, SAGEDelete

__all__ = ['GCN', 'GAT', 'GIN', 'RGCN', 'DeletionLayer', 'DeletionLayerKG', 'GCNDelete', 'GATDelete',
           'GINDelete', 'RGCNDelete', 'SAGE', 'SAGEDelete']
