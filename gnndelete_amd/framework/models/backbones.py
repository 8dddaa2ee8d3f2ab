"""The reference's two-layer backbones on the HIP message-passing layers.

Mirrors framework/models/gcn.py:7-36, gat.py:7-36, gin.py:7-46 and rgcn.py:9-47 of the
reference: same constructor (``args`` with in_dim / hidden_dim / out_dim), same attribute names
(conv1, conv2, node_emb, W - load-bearing for checkpoints and for the ``'del' in name``
optimizer filter of delete_gnn.py:217), same forward / decode signatures."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import ops
from ...nn import GATConv, GCNConv, GINConv, RGATConv, RGCNConv, SAGEConv


class _Homogeneous(nn.Module):
    """x1 = conv1(x, E); x2 = conv2(relu(x1), E); dot-product link decoder."""

    def _make_convs(self, args):
        raise NotImplementedError

    def __init__(self, args, **kwargs):
        super().__init__()
        self.conv1, self.conv2 = self._make_convs(args)

    def forward(self, x, edge_index, return_all_emb=False):
        x1 = self.conv1(x, edge_index)
        x2 = self.conv2(F.relu(x1), edge_index)
        return (x1, x2) if return_all_emb else x2

    def decode(self, z, pos_edge_index, neg_edge_index=None):
        ei = pos_edge_index if neg_edge_index is None else torch.cat([pos_edge_index, neg_edge_index], dim=-1)
        return ops.edge_dot(z, ei[0], ei[1])


class GCN(_Homogeneous):
    def _make_convs(self, args):
        return GCNConv(args.in_dim, args.hidden_dim), GCNConv(args.hidden_dim, args.out_dim)


class GAT(_Homogeneous):
    def _make_convs(self, args):
        return GATConv(args.in_dim, args.hidden_dim), GATConv(args.hidden_dim, args.out_dim)


class GIN(_Homogeneous):
    def _make_convs(self, args):
        return (GINConv(nn.Linear(args.in_dim, args.hidden_dim)),
                GINConv(nn.Linear(args.hidden_dim, args.out_dim)))


class SAGE(_Homogeneous):
    """2-layer GraphSAGE (mean) - not in the reference's registry; added because BASELINE.json's
    config 3 names it.  Same wiring and decoder as GCN / GAT / GIN."""

    def _make_convs(self, args):
        return SAGEConv(args.in_dim, args.hidden_dim), SAGEConv(args.hidden_dim, args.out_dim)


class RGCN(nn.Module):
    """nn.Embedding -> RGCNConv x2 (2R relation types, 4 blocks when R > 20) + DistMult decoder."""

    def __init__(self, args, num_nodes, num_edge_type, **kwargs):
        super().__init__()
        self.args = args
        self.num_edge_type = num_edge_type
        self.node_emb = nn.Embedding(num_nodes, args.in_dim)
        blocks = 4 if num_edge_type > 20 else None
        self.conv1 = RGCNConv(args.in_dim, args.hidden_dim, num_edge_type * 2, num_blocks=blocks)
        self.conv2 = RGCNConv(args.hidden_dim, args.out_dim, num_edge_type * 2, num_blocks=blocks)
        self.relu = nn.ReLU()
        self.W = nn.Parameter(torch.empty(num_edge_type, args.out_dim))
        nn.init.xavier_uniform_(self.W, gain=nn.init.calculate_gain('relu'))

    def forward(self, x, edge, edge_type, return_all_emb=False):
        x = self.node_emb(x)
        x1 = self.conv1(x, edge, edge_type)
        x2 = self.conv2(self.relu(x1), edge, edge_type)
        return (x1, x2) if return_all_emb else x2

    def decode(self, z, edge_index, edge_type):
        return ops.edge_dot(z, edge_index[0], edge_index[1], self.W, edge_type)


class RGAT(RGCN):
    """nn.Embedding -> RGATConv x2 + DistMult decoder (framework/models/rgat.py:353-391)."""

    def __init__(self, args, num_nodes, num_edge_type, **kwargs):
        nn.Module.__init__(self)
        self.args = args
        self.num_edge_type = num_edge_type
        self.node_emb = nn.Embedding(num_nodes, args.in_dim)
        blocks = 4 if num_edge_type > 20 else None
        self.conv1 = RGATConv(args.in_dim, args.hidden_dim, num_edge_type * 2, num_blocks=blocks)
        self.conv2 = RGATConv(args.hidden_dim, args.out_dim, num_edge_type * 2, num_blocks=blocks)
        self.relu = nn.ReLU()
        self.W = nn.Parameter(torch.empty(num_edge_type, args.out_dim))
        nn.init.xavier_uniform_(self.W, gain=nn.init.calculate_gain('relu'))
