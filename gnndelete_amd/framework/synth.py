"""Seeded synthetic stand-ins for the reference's datasets (no dataset is reachable offline;
prepare_dataset.py:141-150,270-302 download them).  Same shapes and the same preparation recipe
as prepare_dataset.py:31-136,186-264: unique ``row < col`` edges, randperm split with test and
validation edges taken first, negatives that avoid the positives, and the IN / OUT Df candidate
masks = inside / outside the 2-hop enclosing subgraph of the test edges.

The graph model is a degree-corrected stochastic block model: heavy-tailed expected degrees
(w_i ~ rank^-1/2, like citation / co-authorship graphs) and communities that carry both the
majority of the edges and a feature mean, so link prediction on it is learnable.  Node ids are
randomly permuted: no locality is baked into the numbering."""
import math

import torch

from .data import Data
from .graph_utils import k_hop_subgraph, negative_sampling
from .utils import negative_sampling_kg

# name -> (num_nodes, num_features, unique row<col edges, feature style)
SHAPES = {
    'synth-cora': (19793, 8710, 63421, 'sparse'),
    'synth-dblp': (17716, 1639, 52867, 'sparse'),
    'synth-collab': (235868, 128, 1179052, 'dense'),
    'synth-tiny': (600, 32, 2400, 'dense'),
    'synth-small': (5000, 64, 30000, 'dense'),
}


def _sample_by_weight(cdf, k, gen):
    u = torch.rand(k, generator=gen, dtype=torch.float64)
    return torch.searchsorted(cdf, u).clamp(max=cdf.numel() - 1)


def dcsbm_edges(n, m, seed, comm_size=128, p_in=0.8, gamma=0.5):
    """m unique undirected edges (row < col) of a degree-corrected SBM on n nodes."""
    gen = torch.Generator().manual_seed(seed)
    n_comm = max(1, n // comm_size)
    rank = torch.arange(n, dtype=torch.float64)
    w = (rank + 1.0).pow(-gamma)
    cdf = torch.cumsum(w / w.sum(), 0)
    per_comm = (n + n_comm - 1) // n_comm
    wr = (torch.arange(per_comm, dtype=torch.float64) * n_comm + 1.0).pow(-gamma)
    cdf_r = torch.cumsum(wr / wr.sum(), 0)
    relabel = torch.randperm(n, generator=gen)
    keys = torch.empty(0, dtype=torch.long)
    while keys.numel() < m:
        k = int((m - keys.numel()) * 1.3) + 1024
        u = _sample_by_weight(cdf, k, gen)                     # latent index; community = u % n_comm
        inside = torch.rand(k, generator=gen) < p_in
        v_in = (_sample_by_weight(cdf_r, k, gen) * n_comm + u % n_comm).clamp(max=n - 1)
        v = torch.where(inside, v_in, _sample_by_weight(cdf, k, gen))
        a, b = relabel[u], relabel[v]
        lo, hi = torch.minimum(a, b), torch.maximum(a, b)
        new = (lo * n + hi)[lo != hi]
        keys = torch.unique(torch.cat([keys, new]))
    keys = keys[torch.randperm(keys.numel(), generator=gen)[:m]]
    community = torch.empty(n, dtype=torch.long)
    community[relabel] = torch.arange(n) % n_comm
    return torch.stack([keys // n, keys % n]), community


def _features(n, f, style, community, gen):
    if style == 'dense':
        n_comm = int(community.max()) + 1
        means = torch.randn(n_comm, f, generator=gen) * 0.1
        return means[community] + torch.randn(n, f, generator=gen) * 0.1
    # ~1 % dense bag-of-words rows, row-normalised (T.NormalizeFeatures, prepare_dataset.py:142)
    nnz_per_row = max(1, int(0.01 * f))
    x = torch.zeros(n, f)
    cols = torch.randint(0, f, (n, nnz_per_row), generator=gen)
    x.scatter_(1, cols, torch.rand(n, nnz_per_row, generator=gen) + 0.05)
    return x / x.sum(1, keepdim=True).clamp(min=1e-12)


def split_linkpred(x, edges, n, gen, val_ratio=0.05, test_ratio=0.05, two_hop_degree=None):
    """The reference's split recipe (prepare_dataset.py:31-136 train_test_split_edges_no_neg_adj_mask,
    :205-214 IN / OUT masks) on unique ``row < col`` edges: randperm, test then validation edges first,
    negatives that avoid the positives, Df candidates inside / outside the 2-hop enclosing subgraph of
    the test edges.  two_hop_degree (per edge, optional): edges whose two-hop degree is below 50 are drawn first,
    so that they make up the test / validation sets - what upstream does for ogbl-* (:54-64, :186-189).
    -> (data, {'in': mask, 'out': mask})"""
    m = edges.shape[1]
    n_v, n_t = int(math.floor(val_ratio * m)), int(math.floor(test_ratio * m))
    if two_hop_degree is not None:
        low_mask = two_hop_degree < 50
        low, high = low_mask.nonzero().flatten(), (~low_mask).nonzero().flatten()
        low = low[torch.randperm(low.shape[0], generator=gen)]
        high = high[torch.randperm(high.shape[0], generator=gen)]
        perm = torch.cat([low, high])
    else:
        perm = torch.randperm(m, generator=gen)
    edges = edges[:, perm]
    test_pos, val_pos, train = edges[:, :n_t], edges[:, n_t:n_t + n_v], edges[:, n_t + n_v:]
    data = Data(x=x, num_nodes=n, num_features=int(x.shape[1]) if x.dim() == 2 else 0, train_pos_edge_index=train,
                test_pos_edge_index=test_pos, val_pos_edge_index=val_pos,
                test_neg_edge_index=negative_sampling(test_pos, n, n_t, generator=gen),
                val_neg_edge_index=negative_sampling(val_pos, n, n_v, generator=gen))
    _, _, _, local = k_hop_subgraph(test_pos.flatten().unique(), 2, train, num_nodes=n)
    return data, {'in': local, 'out': ~local}


def make_linkpred_dataset(name='synth-collab', seed=42, val_ratio=0.05, test_ratio=0.05, shape=None):
    """-> (data, df_masks) where data has x, num_nodes, train_pos_edge_index (directed row<col),
    {val,test}_{pos,neg}_edge_index and df_masks = {'in': mask, 'out': mask} over the train edges
    (the content of the reference's d_<seed>.pkl and df_<seed>.pt)."""
    n, f, m, style = shape if shape is not None else SHAPES[name]
    gen = torch.Generator().manual_seed(seed)
    edges, community = dcsbm_edges(n, m, seed)
    x = _features(n, f, style, community, gen)
    return split_linkpred(x, edges, n, gen, val_ratio, test_ratio)


# name -> (num_nodes, num_relations, unique triples)
KG_SHAPES = {
    'synth-kg-tiny': (500, 4, 3000),
    'synth-kg-small': (4000, 25, 60000),          # > 20 relation types: block-diagonal relation weights like ogbl-biokg
    'synth-wn18': (40943, 18, 151442),
    'synth-biokg': (93773, 51, 4762678),
}


def make_kg_dataset(name='synth-kg-tiny', seed=42, val_ratio=0.05, test_ratio=0.05, shape=None):
    """Knowledge-graph stand-in in the layout prepare_dataset.py:266-399 writes: x = arange(N)
    (entity ids for the embedding table), directed train triples (head < tail) with relation
    types whose sizes follow a Zipf law, val / test triples with per-relation head-shuffled
    negatives (framework/utils.py:46-58), and the IN / OUT Df candidate masks."""
    n, r, m = shape if shape is not None else KG_SHAPES[name]
    gen = torch.Generator().manual_seed(seed)
    edges, _ = dcsbm_edges(n, m, seed, comm_size=64, p_in=0.6)
    m = edges.shape[1]
    w = 1.0 / torch.arange(1, r + 1, dtype=torch.float64)
    rel = torch.searchsorted(torch.cumsum(w / w.sum(), 0), torch.rand(m, generator=gen, dtype=torch.float64)).clamp(max=r - 1)
    state = torch.get_rng_state()
    torch.manual_seed(seed)
    data, masks = split_kg(torch.arange(n), edges, rel, n, val_ratio, test_ratio)
    torch.set_rng_state(state)
    return data, masks


def split_kg(x, edges, edge_type, n, val_ratio=0.05, test_ratio=0.05):
    """The KG branch of train_test_split_edges_no_neg_adj_mask (prepare_dataset.py:31-136, kg=True) on directed
    triples: the edges are permuted with torch.randperm on the GLOBAL generator, test then validation triples are
    taken first - and the relation types are sliced WITHOUT the permutation, exactly as upstream does (:79, :104,
    :119; pinned by tests/golden/split.npz) - negatives by per-relation head shuffling (framework/utils.py:46-58,
    test first, then validation), Df candidates inside / outside the 2-hop enclosing subgraph of the test triples."""
    m = edges.shape[1]
    n_v, n_t = int(math.floor(val_ratio * m)), int(math.floor(test_ratio * m))
    edges = edges[:, torch.randperm(m)]
    data = Data(x=x, num_nodes=n, num_features=0,
                train_pos_edge_index=edges[:, n_t + n_v:], train_edge_type=edge_type[n_t + n_v:],
                test_pos_edge_index=edges[:, :n_t], test_edge_type=edge_type[:n_t],
                val_pos_edge_index=edges[:, n_t:n_t + n_v], val_edge_type=edge_type[n_t:n_t + n_v])
    data.test_neg_edge_index = negative_sampling_kg(data.test_pos_edge_index, data.test_edge_type)
    data.val_neg_edge_index = negative_sampling_kg(data.val_pos_edge_index, data.val_edge_type)
    _, _, _, local = k_hop_subgraph(data.test_pos_edge_index.flatten().unique(), 2, data.train_pos_edge_index, num_nodes=n)
    return data, {'in': local, 'out': ~local}


def make_nodecls_dataset(name='synth-dblp', seed=42, num_classes=4, shape=None):
    """Node-classification stand-in for delete_node.py's CitationFull-DBLP + RandomNodeSplit
    (delete_node.py:54-64): undirected edge_index, features, labels tied to the planted
    communities, train_rest split with 500 validation and 1000 test nodes (scaled down for
    graphs smaller than 3000 nodes)."""
    n, f, m, style = shape if shape is not None else SHAPES[name]
    gen = torch.Generator().manual_seed(seed)
    edges, community = dcsbm_edges(n, m, seed)
    x = _features(n, f, style, community, gen)
    und = torch.cat([edges, edges.flip(0)], dim=1)
    order = torch.argsort(und[0] * n + und[1])
    y = community % num_classes
    num_val, num_test = (500, 1000) if n >= 3000 else (n // 6, n // 3)
    perm = torch.randperm(n, generator=gen)
    val_mask = torch.zeros(n, dtype=torch.bool)
    test_mask = torch.zeros(n, dtype=torch.bool)
    val_mask[perm[:num_val]] = True
    test_mask[perm[num_val:num_val + num_test]] = True
    return Data(x=x, y=y, num_nodes=n, num_features=f, num_classes=num_classes, edge_index=und[:, order],
                train_mask=~(val_mask | test_mask), val_mask=val_mask, test_mask=test_mask)
