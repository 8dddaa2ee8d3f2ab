"""Message-passing layers on the HIP kernels, with torch_geometric's parameter names so the
reference's checkpoints (state_dict keys ``conv1.lin.weight``, ``conv1.bias``,
``conv1.lin_src.weight``, ``conv1.att_src``, ``conv1.nn.weight``, ``conv1.weight``/``root`` ...)
load unchanged (delete_gnn.py:206-207, strict=False).

Semantics follow the torch_geometric convs the reference instantiates
(framework/models/gcn.py:11-12, gat.py:11-12, gin.py:11-12, rgcn.py:17-22); they are pinned by
the dense known-answer tests of the oracle and by HIP-vs-oracle parity tests.  The dense
feature transforms (x @ W^T, their input and weight gradients) run on the fp32 matrix-core kernels of
libgnndelete_hip.so too (ops.dense: whole-weight-in-LDS row kernel, K-tiled kernel for the wide
bag-of-words inputs), like the sparse aggregation, attention softmax and typed mean."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .graph import TypedNodeCSR, build_typed_csr, graph_for


def _glorot(*shape):
    t = torch.empty(*shape)
    bound = math.sqrt(6.0 / (t.size(-2) + t.size(-1)))
    return nn.Parameter(t.uniform_(-bound, bound))


class _Weight(nn.Module):
    """Bias-free linear map stored as ``weight`` [out, in] (torch_geometric Linear)."""

    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.weight = _glorot(out_dim, in_dim)

    def forward(self, x, const_x=False):
        return ops.dense(x, self.weight, None, const_x)


class GCNConv(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.lin = _Weight(in_channels, out_channels)
        self.bias = nn.Parameter(torch.zeros(out_channels))

    def forward(self, x, edge_index):
        g = graph_for(edge_index, x.shape[0], 'gcn')
        return ops.spmm(self.lin(x, const_x=not x.requires_grad), g, self.bias)


class GATConv(nn.Module):
    """heads = 1, negative_slope = 0.2, self loops, bias; lin_src and lin_dst share one weight."""

    def __init__(self, in_channels, out_channels, negative_slope=0.2):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.negative_slope = negative_slope
        self.lin_src = _Weight(in_channels, out_channels)
        self.lin_dst = self.lin_src
        self.att_src = _glorot(1, 1, out_channels)
        self.att_dst = _glorot(1, 1, out_channels)
        self.bias = nn.Parameter(torch.zeros(out_channels))

    def forward(self, x, edge_index):
        g = graph_for(edge_index, x.shape[0], 'gat')
        h = self.lin_src(x, const_x=not x.requires_grad)
        a_src = (h * self.att_src.view(1, -1)).sum(-1)
        a_dst = (h * self.att_dst.view(1, -1)).sum(-1)
        return ops.gat_aggregate(h, a_src, a_dst, g, self.bias, self.negative_slope)


class GINConv(nn.Module):
    """GINConv(nn.Linear), eps = 0: Linear(x_i + sum_j x_j).  When the layer narrows the features
    the linear map is applied BEFORE the aggregation (W(x_i + sum x_j) = Wx_i + sum Wx_j): same
    result up to fp32 rounding, in_dim/out_dim times less SpMM traffic."""

    def __init__(self, lin, eps=0.0):
        super().__init__()
        self.nn = lin
        self.eps = eps

    def forward(self, x, edge_index):
        g = graph_for(edge_index, x.shape[0], 'sum')
        if isinstance(self.nn, nn.Linear) and self.nn.out_features <= self.nn.in_features:
            return ops.spmm(ops.dense(x, self.nn.weight, None, not x.requires_grad), g, self.nn.bias, 1.0 + self.eps)
        agg = ops.spmm(x, g, None, 1.0 + self.eps)
        return ops.dense(agg, self.nn.weight, self.nn.bias) if isinstance(self.nn, nn.Linear) else self.nn(agg)


class SAGEConv(nn.Module):
    """GraphSAGE, mean aggregation: W_l mean_j x_j + b_l + W_r x_i (PyG parameter names lin_l /
    lin_r).  Extension: the reference has no SAGE model, BASELINE.json's config 3 names one.  The
    neighbour transform is applied before the (linear) mean when the layer narrows."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.lin_l = nn.Linear(in_channels, out_channels)
        self.lin_r = nn.Linear(in_channels, out_channels, bias=False)

    def forward(self, x, edge_index):
        g = graph_for(edge_index, x.shape[0], 'mean')
        const = not x.requires_grad
        if self.out_channels <= self.in_channels:
            agg = ops.spmm(ops.dense(x, self.lin_l.weight, None, const), g, self.lin_l.bias)
        else:
            agg = ops.dense(ops.spmm(x, g), self.lin_l.weight, self.lin_l.bias)
        return agg + ops.dense(x, self.lin_r.weight, None, const)


def _tensor_key(*tensors):
    """Cache key of device tensors that also changes when one of them is edited IN PLACE (storage address, shape
    and torch's version counter) - an identity check alone would hand back a stale CSR after `edge_index[...] = ...`."""
    return tuple((t.data_ptr(), tuple(t.shape), t._version) for t in tensors)


class RGCNConv(nn.Module):
    """aggr='mean', root_weight, bias; dense [R, in, out] or block-diagonal
    [R, num_blocks, in/nb, out/nb] relation weights."""

    def __init__(self, in_channels, out_channels, num_relations, num_blocks=None):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.num_relations, self.num_blocks = num_relations, num_blocks
        if num_blocks is None:
            self.weight = _glorot(num_relations, in_channels, out_channels)
        else:
            assert in_channels % num_blocks == 0 and out_channels % num_blocks == 0
            self.weight = _glorot(num_relations, num_blocks, in_channels // num_blocks, out_channels // num_blocks)
        self.root = _glorot(in_channels, out_channels)
        self.bias = nn.Parameter(torch.zeros(out_channels))
        self._typed = None
        self._typed_nodes = None

    def _typed_csr(self, edge_index, edge_type, n):
        c, key = self._typed, (_tensor_key(edge_index, edge_type), n)
        if c is None or c[0] != key:
            # the entry keeps the key tensors alive, so their storage cannot be recycled for another edge list
            c = (key, build_typed_csr(edge_index, edge_type, n, self.num_relations), edge_index, edge_type)
            self._typed = c
        return c[1]

    def _typed_node_csr(self, edge_index, edge_type, n):
        c, key = self._typed_nodes, (_tensor_key(edge_index, edge_type), n)
        if c is None or c[0] != key:
            c = (key, TypedNodeCSR(edge_index, edge_type, n, self.num_relations), edge_index, edge_type)
            self._typed_nodes = c
        return c[1]

    def forward(self, x, edge_index, edge_type):
        n = x.shape[0]
        # Relation weights that need no gradient - evaluation, or a conv the *Delete wrapper marked
        # frozen (framework/models/deletion.py sets the same plain `requires_grad = False` attribute
        # upstream does, rgcn.py / deletion.py:145-163) - take the fused kernel: no [R, n, d] tensor.
        frozen = (not torch.is_grad_enabled()) or getattr(self, 'requires_grad', True) is False \
            or not self.weight.requires_grad
        nb = 1 if self.num_blocks is None else self.num_blocks
        even = (self.in_channels // nb) % 2 == 0 and (self.out_channels // nb) % 2 == 0
        if frozen and even and self.in_channels <= 128 and self.out_channels <= 128:
            tg = self._typed_node_csr(edge_index, edge_type, n)
            return ops.rgcn_conv_frozen(x, tg, self.weight, self.root, self.bias, nb)
        if even and self.in_channels <= 128 and self.out_channels <= 128:
            # trainable relation weights (original-model / retrain training, base.py:394-493, retrain.py:235-339): the same
            # typed conv kernels forward and for the input gradient, the weight gradient straight from the relation-major
            # edge list (gd_typed_wgrad_f32) - no [R, n, in] tensor of per-relation means, no torch.einsum
            tg = self._typed_node_csr(edge_index, edge_type, n)
            return ops.typed_conv(x, tg, self.weight, nb) + ops.dense(x, self.root.t(), self.bias)
        # widths the typed kernels do not have (odd block widths, above 128): per-relation means + einsum
        typed = self._typed_csr(edge_index, edge_type, n)
        m = ops.rgcn_mean(x, typed, self.num_relations, n)                  # [R, n, in]
        if self.num_blocks is None:
            out = torch.einsum('rni,rio->no', m, self.weight)
        else:
            mb = m.view(self.num_relations, n, self.num_blocks, -1)
            out = torch.einsum('rnbi,rbio->nbo', mb, self.weight).reshape(n, self.out_channels)
        return out + ops.dense(x, self.root.t(), self.bias)


FastRGCNConv = RGCNConv


class RGATConv(nn.Module):
    """The reference's RGATConv (framework/models/rgat.py:24-351) in the configuration the reference builds
    (rgat.py:361-366): heads = dim = 1, additive self-attention, softmax across relations, no `mod`, no edge
    features.  Parameter names / shapes as upstream registers them (w, l1, b1, l2, b2 are only read by the `mod`
    variants but are part of every checkpoint).

    The attention logit of an edge (j -> i, r) is  leaky_relu(x_i W_r q + x_j W_r k) = A[i, r] + B[j, r]  with
    A = x (W q)^T, B = x (W k)^T two [N, R] tables (one small GEMM each) instead of two per-edge transforms;
    the softmax across the relations of a target node is gd_segment_softmax_f32 (forward and backward);
    the aggregation  y_i = sum_e alpha_e x_j W_r  is the typed conv kernel with alpha as edge weights - under no_grad
    (evaluation) directly, with gradients through ops.typed_conv (input gradient = the transposed typed conv, relation-weight
    gradient and alpha's gradient straight from the edge lists: gd_typed_wgrad_f32 / gd_typed_edge_dot_f32); widths the typed
    kernels do not have fall back to a weighted typed SpMM over the relation-major CSR + the relation-wise transform.
    No Python loop over relations, no host synchronisation."""

    def __init__(self, in_channels, out_channels, num_relations, num_blocks=None, negative_slope=0.2):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.num_relations, self.num_blocks, self.negative_slope = num_relations, num_blocks, negative_slope
        self.q = _glorot(out_channels, 1)
        self.k = _glorot(out_channels, 1)
        self.bias = nn.Parameter(torch.zeros(out_channels))
        if num_blocks is None:
            self.weight = _glorot(num_relations, in_channels, out_channels)
        else:
            assert in_channels % num_blocks == 0 and out_channels % num_blocks == 0
            self.weight = _glorot(num_relations, num_blocks, in_channels // num_blocks, out_channels // num_blocks)
        self.w = nn.Parameter(torch.ones(out_channels))
        self.l1 = nn.Parameter(torch.ones(1, out_channels))
        self.b1 = nn.Parameter(torch.zeros(1, out_channels))
        self.l2 = nn.Parameter(torch.full((out_channels, out_channels), 1.0 / out_channels))
        self.b2 = nn.Parameter(torch.zeros(1, out_channels))
        self._typed_nodes = None

    def _typed_node_csr(self, edge_index, edge_type, n):
        c, key = self._typed_nodes, (_tensor_key(edge_index, edge_type), n)
        if c is None or c[0] != key:
            c = (key, TypedNodeCSR(edge_index, edge_type, n, self.num_relations), edge_index, edge_type)
            self._typed_nodes = c
        return c[1]

    def _relation_vectors(self, v):
        """W_r v for every relation: [R, in]."""
        if self.num_blocks is None:
            return (self.weight @ v).squeeze(-1)
        vb = v.view(self.num_blocks, -1)
        return torch.einsum('rbio,bo->rbi', self.weight, vb).reshape(self.num_relations, self.in_channels)

    def _edge_orders(self, edge_index, edge_type, n):
        """Index structure of one (edge_index, edge_type) pair, built once on the device and cached: the edges grouped by
        target (softmax segments), the relation-major typed CSR (rows r * n + i) and its transpose by source."""
        c, key = getattr(self, '_orders', None), (_tensor_key(edge_index, edge_type), n)
        if c is not None and c[0] == key:
            return c[1]
        dev = edge_index.device
        src, dst, et = edge_index[0].long(), edge_index[1].long(), edge_type.long()
        r = self.num_relations
        if r * n >= 2 ** 31:
            raise ValueError('num_relations * num_nodes overflows int32')
        ord_d = torch.argsort(dst, stable=True)
        rowptr_d = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        rowptr_d[1:] = torch.cumsum(torch.bincount(dst, minlength=n), 0)
        vrow = et * n + dst
        ord_v = torch.argsort(vrow * n + src)
        rowptr_v = torch.zeros(r * n + 1, dtype=torch.int64, device=dev)
        rowptr_v[1:] = torch.cumsum(torch.bincount(vrow, minlength=r * n), 0)
        ord_t = torch.argsort(src * (r * n) + vrow)
        rowptr_t = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        rowptr_t[1:] = torch.cumsum(torch.bincount(src, minlength=n), 0)
        pos_v = torch.empty_like(ord_v)
        pos_v[ord_v] = torch.arange(ord_v.numel(), device=dev)
        i32 = lambda t: t.to(torch.int32).contiguous()
        tc = dict(n=n, n_vrows=r * n, ord_d=ord_d, rowptr_d=i32(rowptr_d), ord_v=ord_v, rowptr_v=i32(rowptr_v),
                  col_v=i32(src[ord_v]), vrow_v=i32(vrow[ord_v]), rowptr_t=i32(rowptr_t), col_t=i32(vrow[ord_t]),
                  v_of_t=pos_v[ord_t], keep=(edge_index, edge_type))
        inv_d = torch.empty_like(ord_d)
        inv_d[ord_d] = torch.arange(ord_d.numel(), device=dev)
        tc['inv_d'] = inv_d
        self._orders = (key, tc)
        return tc

    def forward(self, x, edge_index, edge_type):
        n = x.shape[0]
        src, dst, et = edge_index[0], edge_index[1], edge_type
        a = ops.dense(x, self._relation_vectors(self.q))                # [N, R]: x (W_r q) for every relation, the library's GEMM
        b = ops.dense(x, self._relation_vectors(self.k))
        e = F.leaky_relu(a[dst, et] + b[src, et], self.negative_slope)
        nb = 1 if self.num_blocks is None else self.num_blocks
        if not x.is_cuda:
            raise ops._lib.GnnDeleteHipError('RGATConv needs CUDA(HIP) tensors (no CPU fallback)')
        if e.numel() == 0:              # a sampled batch without edges: every node aggregates nothing (rgat.py: out = 0 + bias)
            return (a.sum() + b.sum()) * 0.0 + self.bias.expand(n, self.out_channels) + x.new_zeros(n, self.out_channels)
        tc = self._edge_orders(edge_index, edge_type, n)
        # softmax across ALL in-edges of a target node, whatever their relation (rgat.py: attention_mode
        # 'additive-self-attention', across-relation): segments = the edges grouped by target
        alpha = ops.segment_softmax(e[tc['ord_d']], tc['rowptr_d'])[tc['inv_d']]
        even = (self.in_channels // nb) % 2 == 0 and (self.out_channels // nb) % 2 == 0
        if not torch.is_grad_enabled() and even and self.in_channels <= 128 and self.out_channels <= 128:
            tg = self._typed_node_csr(edge_index, edge_type, n)
            return ops.rgat_aggregate_nograd(x, tg, alpha, self.weight, self.bias, nb, self.out_channels)
        if even and self.in_channels <= 128 and self.out_channels <= 128:
            # with gradients: the typed conv with alpha as per-edge coefficients, differentiable in x, in the relation weights
            # (gd_typed_wgrad_f32) and in alpha (gd_typed_edge_dot_f32) - no [R, n, in] tensor, no torch.einsum
            tg = self._typed_node_csr(edge_index, edge_type, n)
            return ops.typed_conv(x, tg, self.weight, nb, alpha) + self.bias
        m = ops.typed_weighted_sum(x, alpha[tc['ord_v']], tc).view(self.num_relations, n, self.in_channels)
        if self.num_blocks is None:
            out = torch.einsum('rni,rio->no', m, self.weight)
        else:
            out = torch.einsum('rnbi,rbio->nbo', m.view(self.num_relations, n, nb, -1), self.weight).reshape(n, self.out_channels)
        return out + self.bias
