"""The Del operator and the *Delete model wrappers (reference: framework/models/deletion.py).

DeletionLayer (deletion.py:8-29): ``new = x.clone(); new[mask] = new[mask] @ W_D`` with
W_D = ones(dim, dim) / 1000.  Here the boolean mask is turned ONCE into a sorted int32 row list
that lives on the device (the reference keeps the mask on the CPU and re-uploads it inside every
forward), and the gather / [S,d]x[d,d] GEMM / scatter run as one MFMA kernel.

*Delete wrappers (deletion.py:52-163): conv1 -> Del1 -> (z1) -> relu -> conv2 -> Del2 -> z2.
Build semantics: the backbone is truly frozen - conv1 runs under no_grad for every architecture.
Upstream does so for GAT/GIN/RGCN; for GCN the no_grad is commented out (deletion.py:62), which
only adds a wasted conv1 weight-gradient and makes GCN + both_layerwise crash;
the Del-weight gradients are identical (tests/test_oracle_golden.py)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import ops
from .backbones import GAT, GCN, GIN, RGAT, RGCN, SAGE


class _RowList:
    """Boolean node mask -> cached sorted int32 index list on a given device."""

    def __init__(self):
        self._key = None
        self._idx = None

    def get(self, mask, device):
        key = (id(mask), mask._version, str(device))
        if self._key != key:
            self._idx = mask.nonzero().flatten().to(device=device, dtype=torch.int32)
            self._key = key
            self._mask_ref = mask          # keep the id stable while cached
        return self._idx


class DeletionLayer(nn.Module):
    def __init__(self, dim, mask):
        super().__init__()
        self.dim = dim
        self.mask = mask
        self.deletion_weight = nn.Parameter(torch.ones(dim, dim) / 1000)
        self._rows = _RowList()
        self._rows_override = _RowList()

    def forward(self, x, mask=None):
        '''Only apply deletion operator to the local nodes identified by mask'''
        if mask is None:
            mask, rows = self.mask, self._rows
        else:
            rows = self._rows_override
        if mask is None:
            return x
        idx = rows.get(mask, x.device)
        return ops.del_rows(x, self.deletion_weight, idx)


DeletionLayerKG = DeletionLayer


def _with_deletion(base):
    relational = base in (RGCN, RGAT)

    class _Delete(base):
        def __init__(self, args, *pos, mask_1hop=None, mask_2hop=None, **kwargs):
            if relational:
                # RGCNDelete(args, num_nodes, num_edge_type, mask_1hop=None, mask_2hop=None)
                names = ['num_nodes', 'num_edge_type', 'mask_1hop', 'mask_2hop']
            else:
                # GCNDelete(args, mask_1hop=None, mask_2hop=None)
                names = ['mask_1hop', 'mask_2hop']
            merged = dict(zip(names, pos))
            if 'mask_1hop' in merged:
                mask_1hop = merged.pop('mask_1hop')
            if 'mask_2hop' in merged:
                mask_2hop = merged.pop('mask_2hop')
            kwargs.update(merged)
            if relational:
                super().__init__(args, kwargs['num_nodes'], kwargs['num_edge_type'])
            else:
                super().__init__(args)
            self.deletion1 = DeletionLayer(args.hidden_dim, mask_1hop)
            self.deletion2 = DeletionLayer(args.out_dim, mask_2hop)
            for frozen in ([self.node_emb] if relational else []) + [self.conv1, self.conv2]:
                frozen.requires_grad = False      # plain attribute, as upstream (no effect on params)

        if relational:
            def forward(self, x, edge_index, edge_type, mask_1hop=None, mask_2hop=None, return_all_emb=False):
                with torch.no_grad():
                    p1 = self.conv1(self.node_emb(x), edge_index, edge_type)
                x1 = self.deletion1(p1, mask_1hop)
                x2 = self.deletion2(self.conv2(F.relu(x1), edge_index, edge_type), mask_2hop)
                return (x1, x2) if return_all_emb else x2

            def get_original_embeddings(self, x, edge_index, edge_type, return_all_emb=False):
                return base.forward(self, x, edge_index, edge_type, return_all_emb)
        else:
            def forward(self, x, edge_index, mask_1hop=None, mask_2hop=None, return_all_emb=False):
                with torch.no_grad():
                    p1 = self.conv1(x, edge_index)
                x1 = self.deletion1(p1, mask_1hop)
                x2 = self.deletion2(self.conv2(F.relu(x1), edge_index), mask_2hop)
                return (x1, x2) if return_all_emb else x2

            def get_original_embeddings(self, x, edge_index, return_all_emb=False):
                return base.forward(self, x, edge_index, return_all_emb)

    _Delete.__name__ = _Delete.__qualname__ = base.__name__ + 'Delete'
    return _Delete


GCNDelete = _with_deletion(GCN)
GATDelete = _with_deletion(GAT)
GINDelete = _with_deletion(GIN)
RGCNDelete = _with_deletion(RGCN)
RGATDelete = _with_deletion(RGAT)
SAGEDelete = _with_deletion(SAGE)
