"""Host/device graph preprocessing the reference takes from torch_geometric.utils
(k_hop_subgraph, to_undirected, is_undirected, negative_sampling: delete_gnn.py:9,128-140,175;
gnndelete_nodeemb.py:182-185).  One-off integer/boolean work per unlearning request - plain
PyTorch on whatever device the tensors live on, bit-exact against the reference's masks
(tests/test_host_utils.py)."""
import torch


def k_hop_subgraph(node_idx, num_hops, edge_index, relabel_nodes=False, num_nodes=None, flow='source_to_target'):
    """Returns (subset, edge_index[:, edge_mask], inverse-of-seeds, edge_mask) like PyG.
    flow='source_to_target': a hop moves from the current frontier to the SOURCES of the edges
    that point INTO it; only the previous hop's frontier is expanded."""
    assert not relabel_nodes, 'relabel_nodes is not used by the reference'
    n = int(edge_index.max()) + 1 if num_nodes is None else int(num_nodes)
    if flow == 'source_to_target':
        walk_to, walk_from = edge_index[0], edge_index[1]
    else:
        walk_to, walk_from = edge_index[1], edge_index[0]
    dev = edge_index.device
    if not torch.is_tensor(node_idx):
        node_idx = torch.tensor(node_idx, device=dev)
    seeds = node_idx.to(dev).flatten().long()
    reached = torch.zeros(n, dtype=torch.bool, device=dev)
    reached[seeds] = True
    frontier = reached.clone()
    for _ in range(num_hops):
        nxt = torch.zeros(n, dtype=torch.bool, device=dev)
        nxt[walk_to[frontier[walk_from]]] = True
        reached |= nxt
        frontier = nxt
    subset = reached.nonzero().flatten()
    edge_mask = reached[edge_index[0]] & reached[edge_index[1]]
    inv = torch.searchsorted(subset, seeds)
    return subset, edge_index[:, edge_mask], inv, edge_mask


def coalesce(edge_index, edge_attr=None, num_nodes=None):
    """Sort by (row, col), merge duplicate edges, add their attributes."""
    n = int(edge_index.max()) + 1 if num_nodes is None else int(num_nodes)
    key = edge_index[0] * n + edge_index[1]
    key_sorted, perm = torch.sort(key, stable=True)
    first = torch.ones_like(key_sorted, dtype=torch.bool)
    first[1:] = key_sorted[1:] != key_sorted[:-1]
    out_index = torch.stack([key_sorted[first] // n, key_sorted[first] % n])
    if edge_attr is None:
        return out_index
    seg = torch.cumsum(first, 0) - 1
    single = torch.is_tensor(edge_attr)
    merged = []
    for a in ([edge_attr] if single else edge_attr):
        acc = torch.zeros((out_index.shape[1],) + tuple(a.shape[1:]), dtype=a.dtype, device=a.device)
        merged.append(acc.index_add_(0, seg, a[perm]))
    return out_index, (merged[0] if single else merged)


def to_undirected(edge_index, edge_attr=None, num_nodes=None, reduce='add'):
    assert reduce == 'add'
    both = torch.cat([edge_index, edge_index.flip(0)], dim=1)
    if edge_attr is None:
        return coalesce(both, None, num_nodes)
    single = torch.is_tensor(edge_attr)
    doubled = [torch.cat([a, a], 0) for a in ([edge_attr] if single else edge_attr)]
    idx, attrs = coalesce(both, doubled, num_nodes)
    return idx, (attrs[0] if single else attrs)


def is_undirected(edge_index, num_nodes=None):
    n = int(edge_index.max()) + 1 if num_nodes is None else int(num_nodes)
    a = coalesce(edge_index, None, n)
    b = coalesce(edge_index.flip(0), None, n)
    return a.shape == b.shape and bool((a == b).all())


def negative_sampling(edge_index, num_nodes=None, num_neg_samples=None, generator=None):
    """Uniform random node pairs that are not edges of ``edge_index`` (and not self loops).
    PyG draws these from Python's `random`, so the sample itself can never match the
    reference's; what is matched is the contract: [2, num_neg_samples] int64, no positives."""
    n = int(edge_index.max()) + 1 if num_nodes is None else int(num_nodes)
    want = int(edge_index.shape[1] if num_neg_samples is None else num_neg_samples)
    dev = edge_index.device
    pos = torch.unique(edge_index[0] * n + edge_index[1])
    got = torch.empty(0, dtype=torch.long, device=dev)
    tries = 0
    while got.numel() < want and tries < 64:
        m = int((want - got.numel()) * 1.2) + 16
        cand = torch.randint(0, n * n, (m,), generator=generator, device='cpu').to(dev)
        ok = (cand // n != cand % n)
        loc = torch.searchsorted(pos, cand).clamp(max=max(pos.numel() - 1, 0))
        if pos.numel():
            ok &= pos[loc] != cand
        got = torch.cat([got, cand[ok]])
        tries += 1
    got = got[:want]
    return torch.stack([got // n, got % n])
