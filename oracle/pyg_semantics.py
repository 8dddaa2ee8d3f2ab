"""CPU restatement of the torch_geometric arithmetic the reference calls.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  **Parity unpinned**: the
reference depends on an un-vendored, un-pinned ``torch_geometric`` (API usage implies
2.0.4 - 2.2; no requirements file) and holds no tests at this boundary.  What is
restated here is the published PyG algorithm for each call site:

  GCNConv           framework/models/gcn.py:11-12      (gcn_norm + propagate, bias)
  GATConv           framework/models/gat.py:11-12      (heads=1, slope .2, self loops)
  GINConv(Linear)   framework/models/gin.py:11-12      (eps=0, sum aggregation)
  RGCNConv          framework/models/rgcn.py:17-22     (mean aggr, block-diag, root, bias)
  k_hop_subgraph    delete_gnn.py:128-140
  to_undirected     delete_gnn.py:175
  is_undirected     delete_gnn.py:156,182

and is pinned by the dense closed-form known-answer tests in
tests/test_oracle_kat.py.  Everything is plain index_add_/scatter arithmetic on CPU
tensors (the same gather + scatter-add PyG's CPU path performs), differentiable
through torch autograd.  Edge convention: ``edge_index[0]`` = source j,
``edge_index[1]`` = target i (flow source_to_target).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


# ----------------------------------------------------------------------------
# graph utilities
# ----------------------------------------------------------------------------
def with_single_self_loops(edge_index, num_nodes):
    """PyG ``add_remaining_self_loops`` for unweighted graphs: drop every existing
    self loop, keep all other edges in order, append exactly one loop per node."""
    keep = edge_index[0] != edge_index[1]
    loops = torch.arange(num_nodes, dtype=edge_index.dtype, device=edge_index.device)
    return torch.cat([edge_index[:, keep], torch.stack([loops, loops])], dim=1)


def gcn_norm(edge_index, num_nodes, dtype=torch.float32):
    """Symmetric normalisation D^-1/2 (A+I) D^-1/2; degree = weighted IN-degree
    (scatter over the target column) including the loop; multi-edges count."""
    ei = with_single_self_loops(edge_index, num_nodes)
    src, dst = ei[0], ei[1]
    w = torch.ones(ei.shape[1], dtype=dtype, device=ei.device)
    deg = torch.zeros(num_nodes, dtype=dtype, device=ei.device).index_add_(0, dst, w)
    dis = deg.pow(-0.5)
    dis[torch.isinf(dis)] = 0.0
    return ei, dis[src] * w * dis[dst]


def scatter_rows(msg, dst, num_nodes):
    out = torch.zeros(num_nodes, msg.shape[1], dtype=msg.dtype, device=msg.device)
    return out.index_add(0, dst, msg)


def k_hop_subgraph(node_idx, num_hops, edge_index, num_nodes):
    """PyG k_hop_subgraph(relabel_nodes=False, flow='source_to_target').

    Each hop looks at the PREVIOUS frontier only and adds the *sources* of edges
    whose *target* is in it.  Returns (subset, induced edge_index, edge_mask)."""
    src, dst = edge_index[0], edge_index[1]
    node_idx = torch.as_tensor(node_idx, dtype=torch.long).flatten()
    frontier = node_idx
    collected = [node_idx]
    for _ in range(num_hops):
        in_frontier = torch.zeros(num_nodes, dtype=torch.bool)
        in_frontier[frontier] = True
        frontier = src[in_frontier[dst]]
        collected.append(frontier)
    subset = torch.cat(collected).unique()
    member = torch.zeros(num_nodes, dtype=torch.bool)
    member[subset] = True
    edge_mask = member[src] & member[dst]
    return subset, edge_index[:, edge_mask], edge_mask


def coalesce(edge_index, attrs, num_nodes):
    """Sort by (row, col), merge duplicates, sum integer/float attributes."""
    key = edge_index[0] * num_nodes + edge_index[1]
    uniq, inv = torch.unique(key, sorted=True, return_inverse=True)
    out_index = torch.stack([uniq // num_nodes, uniq % num_nodes])
    out_attrs = []
    for a in attrs:
        acc = torch.zeros(uniq.shape[0], dtype=a.dtype)
        out_attrs.append(acc.index_add_(0, inv, a))
    return out_index, out_attrs


def to_undirected(edge_index, attrs, num_nodes):
    """PyG to_undirected(edge_index, [attrs], reduce='add'): concatenate the reversed
    edges, duplicate the attributes, coalesce."""
    both = torch.cat([edge_index, edge_index.flip(0)], dim=1)
    return coalesce(both, [torch.cat([a, a]) for a in attrs], num_nodes)


def saint_subgraph(data, node_idx, num_nodes=None):
    """One batch of torch_geometric's GraphSAINTSampler for the (sorted, unique) node set ``node_idx`` [PyG-mem]:
    the induced subgraph in the adjacency's (row, col) order - SparseTensor(row=edge_index[0], col=edge_index[1],
    value=arange(E)).saint_subgraph - with endpoints relabelled to positions in ``node_idx``; tensors whose first
    dimension is N are sliced by ``node_idx``, those whose first dimension is E by the kept edges, the rest are
    passed through.  ``data`` is a dict; -> dict with num_nodes = len(node_idx)."""
    ei = data['edge_index']
    n = int(data['num_nodes'] if num_nodes is None else num_nodes)
    e = ei.shape[1]
    member = torch.zeros(n, dtype=torch.bool)
    member[node_idx] = True
    order = torch.argsort(ei[0] * n + ei[1], stable=True)
    keep = order[(member[ei[0]] & member[ei[1]])[order]]
    relabel = torch.full((n,), -1, dtype=torch.long)
    relabel[node_idx] = torch.arange(node_idx.numel())
    out = {'num_nodes': int(node_idx.numel()), 'edge_index': relabel[ei[:, keep]]}
    for k, v in data.items():
        if k in ('edge_index', 'num_nodes'):
            continue
        if torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == n:
            out[k] = v[node_idx]
        elif torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == e:
            out[k] = v[keep]
        else:
            out[k] = v
    return out


def is_undirected(edge_index, num_nodes):
    a, _ = coalesce(edge_index, [], num_nodes)
    b, _ = coalesce(edge_index.flip(0), [], num_nodes)
    return a.shape == b.shape and bool((a == b).all())


# ----------------------------------------------------------------------------
# conv arithmetic (functional)
# ----------------------------------------------------------------------------
def gcn_conv(x, edge_index, weight, bias):
    """out = D^-1/2 (A+I) D^-1/2 (x W^T) + b ; weight is [out, in] (PyG Linear)."""
    n = x.shape[0]
    ei, w = gcn_norm(edge_index, n, x.dtype)
    h = x @ weight.t()
    out = scatter_rows(h[ei[0]] * w[:, None], ei[1], n)
    return out + bias if bias is not None else out


def segment_softmax(score, dst, num_nodes):
    """PyG utils.softmax: per-target max-shifted exp / (sum + 1e-16)."""
    smax = torch.full((num_nodes,), -math.inf, dtype=score.dtype, device=score.device)
    smax = smax.scatter_reduce(0, dst, score.detach(), reduce='amax', include_self=True)
    e = (score - smax[dst]).exp()
    denom = torch.zeros(num_nodes, dtype=score.dtype, device=score.device).index_add(0, dst, e)
    return e / (denom[dst] + 1e-16)


def gat_conv(x, edge_index, weight, att_src, att_dst, bias, negative_slope=0.2):
    """GATConv heads=1: h = xW^T; e_ij = LeakyReLU(a_s.h_j + a_d.h_i); softmax over the
    in-edges of i (existing loops removed, one loop per node appended); out = sum a_ij h_j + b."""
    n = x.shape[0]
    ei = with_single_self_loops(edge_index, n)
    src, dst = ei[0], ei[1]
    h = x @ weight.t()
    a_s = (h * att_src.view(1, -1)).sum(-1)
    a_d = (h * att_dst.view(1, -1)).sum(-1)
    alpha = segment_softmax(F.leaky_relu(a_s[src] + a_d[dst], negative_slope), dst, n)
    out = scatter_rows(h[src] * alpha[:, None], dst, n)
    return out + bias if bias is not None else out


def gin_conv(x, edge_index, weight, bias, eps=0.0):
    """GINConv(nn.Linear): Linear((1+eps) x_i + sum_{j->i} x_j); no loops, no norm."""
    n = x.shape[0]
    agg = scatter_rows(x[edge_index[0]], edge_index[1], n)
    return F.linear((1.0 + eps) * x + agg, weight, bias)


def sage_conv(x, edge_index, w_l, b_l, w_r):
    """SAGEConv (aggr='mean', root_weight=True, bias on lin_l only, no loops, no normalisation):
    out_i = W_l mean_{j->i} x_j + b_l + W_r x_i ; a node without in-edges aggregates zeros.
    NOT used by the reference (it has no SAGE model; BASELINE.json config 3 names one) - an
    extension pinned only by the dense known-answer test."""
    n = x.shape[0]
    src, dst = edge_index[0], edge_index[1]
    cnt = torch.zeros(n, dtype=x.dtype, device=x.device).index_add_(0, dst, torch.ones(src.shape[0], dtype=x.dtype, device=x.device))
    mean = scatter_rows(x[src], dst, n) / cnt.clamp(min=1.0)[:, None]
    return F.linear(mean, w_l, b_l) + F.linear(x, w_r)


def rgcn_conv(x, edge_index, edge_type, weight, root, bias, num_blocks=None):
    """RGCNConv(aggr='mean', root_weight=True): per relation r, mean of x_j over the
    in-edges of type r, times W_r (dense [in,out] or block-diagonal
    [num_blocks, in/nb, out/nb]); plus x @ root + bias."""
    n = x.shape[0]
    num_rel = weight.shape[0]
    out_dim = root.shape[1]
    out = torch.zeros(n, out_dim, dtype=x.dtype, device=x.device)
    for r in range(num_rel):
        sel = edge_type == r
        src, dst = edge_index[0, sel], edge_index[1, sel]
        cnt = torch.zeros(n, dtype=x.dtype, device=x.device).index_add_(0, dst, torch.ones(src.shape[0], dtype=x.dtype, device=x.device))
        h = scatter_rows(x[src], dst, n) / cnt.clamp(min=1.0)[:, None]
        if num_blocks is not None:
            hb = h.view(n, num_blocks, -1)
            out = out + torch.einsum('abc,bcd->abd', hb, weight[r]).reshape(n, out_dim)
        else:
            out = out + h @ weight[r]
    out = out + x @ root
    return out + bias if bias is not None else out


def rgat_conv(x, edge_index, edge_type, weight, q, k, bias, num_blocks=None, negative_slope=0.2):
    """The reference's RGATConv (framework/models/rgat.py:24-351, itself torch_geometric's) in the only
    configuration the reference instantiates (rgat.py:361-366): heads = dim = 1, additive self-attention,
    across-relation softmax, mod = None, no edge features.  Per edge e = (j -> i, type r):
        out_i = x_i W_r, out_j = x_j W_r  (W_r dense [in,out] or block-diagonal [nb, in/nb, out/nb], :188-206)
        alpha_e = softmax over ALL in-edges of i of leaky_relu(out_i q + out_j k)   (:208, :226-241)
        y_i = sum_e alpha_e out_j + bias                                            (:322-323, :330-337)"""
    n = x.shape[0]
    src, dst = edge_index[0], edge_index[1]
    if num_blocks is not None:
        w = weight[edge_type]                                           # [E, nb, in/nb, out/nb]
        xi = x[dst].view(-1, num_blocks, w.shape[2])
        xj = x[src].view(-1, num_blocks, w.shape[2])
        outi = torch.einsum('ebi,ebio->ebo', xi, w).reshape(src.shape[0], -1)
        outj = torch.einsum('ebi,ebio->ebo', xj, w).reshape(src.shape[0], -1)
    else:
        w = weight[edge_type]                                           # [E, in, out]
        outi = torch.bmm(x[dst].unsqueeze(1), w).squeeze(1)
        outj = torch.bmm(x[src].unsqueeze(1), w).squeeze(1)
    e = F.leaky_relu((outi @ q + outj @ k).squeeze(-1), negative_slope)
    alpha = segment_softmax(e, dst, n)
    out = scatter_rows(alpha[:, None] * outj, dst, n)
    return out + bias if bias is not None else out


# ----------------------------------------------------------------------------
# nn.Module wrappers with PyG's parameter names (state_dict keys are part of the
# checkpoint contract: delete_gnn.py:206-207 loads model_best.pt with strict=False)
# ----------------------------------------------------------------------------
def glorot_(t):
    fan = t.size(-2) + t.size(-1)
    bound = math.sqrt(6.0 / fan)
    with torch.no_grad():
        return t.uniform_(-bound, bound)


class _BiaslessLinear(nn.Module):
    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.weight = nn.Parameter(glorot_(torch.empty(out_dim, in_dim)))


class GCNConv(nn.Module):
    """keys: lin.weight [out,in], bias [out]"""
    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.lin = _BiaslessLinear(in_dim, out_dim)
        self.bias = nn.Parameter(torch.zeros(out_dim))

    def forward(self, x, edge_index):
        return gcn_conv(x, edge_index, self.lin.weight, self.bias)


class GATConv(nn.Module):
    """keys: lin_src.weight (shared with lin_dst), att_src, att_dst [1,1,out], bias"""
    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.lin_src = _BiaslessLinear(in_dim, out_dim)
        self.lin_dst = self.lin_src
        self.att_src = nn.Parameter(glorot_(torch.empty(1, 1, out_dim)))
        self.att_dst = nn.Parameter(glorot_(torch.empty(1, 1, out_dim)))
        self.bias = nn.Parameter(torch.zeros(out_dim))

    def forward(self, x, edge_index):
        return gat_conv(x, edge_index, self.lin_src.weight, self.att_src, self.att_dst, self.bias)


class GINConv(nn.Module):
    """keys: nn.weight, nn.bias (the wrapped nn.Linear)"""
    def __init__(self, lin):
        super().__init__()
        self.nn = lin
        self.eps = 0.0

    def forward(self, x, edge_index):
        return gin_conv(x, edge_index, self.nn.weight, self.nn.bias, self.eps)


class SAGEConv(nn.Module):
    """keys: lin_l.weight, lin_l.bias, lin_r.weight"""
    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.lin_l = nn.Linear(in_dim, out_dim)
        self.lin_r = nn.Linear(in_dim, out_dim, bias=False)

    def forward(self, x, edge_index):
        return sage_conv(x, edge_index, self.lin_l.weight, self.lin_l.bias, self.lin_r.weight)


class RGCNConv(nn.Module):
    """keys: weight [R,in,out] or [R,nb,in/nb,out/nb], root [in,out], bias [out]"""
    def __init__(self, in_dim, out_dim, num_relations, num_blocks=None):
        super().__init__()
        self.num_blocks = num_blocks
        if num_blocks is None:
            self.weight = nn.Parameter(glorot_(torch.empty(num_relations, in_dim, out_dim)))
        else:
            self.weight = nn.Parameter(glorot_(torch.empty(
                num_relations, num_blocks, in_dim // num_blocks, out_dim // num_blocks)))
        self.root = nn.Parameter(glorot_(torch.empty(in_dim, out_dim)))
        self.bias = nn.Parameter(torch.zeros(out_dim))

    def forward(self, x, edge_index, edge_type):
        return rgcn_conv(x, edge_index, edge_type, self.weight, self.root, self.bias, self.num_blocks)


FastRGCNConv = RGCNConv


class RGATConv(nn.Module):
    """keys as the reference's RGATConv registers them (rgat.py:96-149): q, k [out,1], bias [out], weight
    [R,in,out] or [R,nb,in/nb,out/nb], and w, l1, b1, l2, b2, which only the `mod` variants read."""
    def __init__(self, in_dim, out_dim, num_relations, num_blocks=None, negative_slope=0.2):
        super().__init__()
        self.num_blocks, self.negative_slope = num_blocks, negative_slope
        self.q = nn.Parameter(glorot_(torch.empty(out_dim, 1)))
        self.k = nn.Parameter(glorot_(torch.empty(out_dim, 1)))
        self.bias = nn.Parameter(torch.zeros(out_dim))
        if num_blocks is None:
            self.weight = nn.Parameter(glorot_(torch.empty(num_relations, in_dim, out_dim)))
        else:
            self.weight = nn.Parameter(glorot_(torch.empty(
                num_relations, num_blocks, in_dim // num_blocks, out_dim // num_blocks)))
        self.w = nn.Parameter(torch.ones(out_dim))
        self.l1 = nn.Parameter(torch.ones(1, out_dim))
        self.b1 = nn.Parameter(torch.zeros(1, out_dim))
        self.l2 = nn.Parameter(torch.full((out_dim, out_dim), 1.0 / out_dim))
        self.b2 = nn.Parameter(torch.zeros(1, out_dim))

    def forward(self, x, edge_index, edge_type):
        return rgat_conv(x, edge_index, edge_type, self.weight, self.q, self.k, self.bias, self.num_blocks,
                         self.negative_slope)
