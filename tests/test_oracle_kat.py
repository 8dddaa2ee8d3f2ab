"""Closed-form dense known-answer tests pinning oracle.pyg_semantics (the PyG boundary the
reference leaves unpinned).  Every expected value is built from dense matrices in fp64,
independently of the gather/scatter code under test.  CPU only."""
import numpy as np
import pytest
import torch

from oracle import pyg_semantics as pyg


def rand_graph(n, m, seed, loops=True, dups=True):
    g = torch.Generator().manual_seed(seed)
    ei = torch.randint(0, n, (2, m), generator=g)
    if not loops:
        ei = ei[:, ei[0] != ei[1]]
    if dups:
        ei = torch.cat([ei, ei[:, :5]], 1)
    return ei


def dense_adj(ei, n, drop_loops=False):
    a = torch.zeros(n, n, dtype=torch.float64)
    for s, d in ei.t().tolist():
        if drop_loops and s == d:
            continue
        a[d, s] += 1.0            # row = target, col = source
    return a


@pytest.mark.parametrize('seed', [0, 1, 2])
def test_gcn_conv_is_sym_normalised_dense_product(seed):
    n, f, o = 17, 5, 4
    ei = rand_graph(n, 40, seed)
    g = torch.Generator().manual_seed(seed + 100)
    x = torch.randn(n, f, generator=g, dtype=torch.float64)
    w = torch.randn(o, f, generator=g, dtype=torch.float64)
    b = torch.randn(o, generator=g, dtype=torch.float64)
    a = dense_adj(ei, n, drop_loops=True) + torch.eye(n, dtype=torch.float64)
    deg = a.sum(1)                                   # in-degree incl. the single loop
    want = (a / deg.sqrt()[:, None] / deg.sqrt()[None, :]) @ (x @ w.t()) + b
    got = pyg.gcn_conv(x, ei, w, b)
    assert torch.allclose(got, want, atol=1e-12)


def test_gcn_isolated_node_gets_only_its_loop():
    ei = torch.tensor([[0, 1], [1, 0]])
    x = torch.eye(3, dtype=torch.float64)
    got = pyg.gcn_conv(x, ei, torch.eye(3, dtype=torch.float64), None)
    assert torch.allclose(got[2], torch.tensor([0, 0, 1.0], dtype=torch.float64))
    assert torch.allclose(got[0], torch.tensor([0.5, 0.5, 0], dtype=torch.float64))


@pytest.mark.parametrize('seed', [0, 1])
def test_gat_conv_is_masked_dense_softmax_attention(seed):
    n, f, o = 13, 6, 5
    ei = rand_graph(n, 30, seed, dups=False)
    ei = torch.unique(ei, dim=1)                      # simple graph for the dense form
    g = torch.Generator().manual_seed(seed + 7)
    x = torch.randn(n, f, generator=g, dtype=torch.float64)
    w = torch.randn(o, f, generator=g, dtype=torch.float64)
    a_s = torch.randn(o, generator=g, dtype=torch.float64)
    a_d = torch.randn(o, generator=g, dtype=torch.float64)
    b = torch.randn(o, generator=g, dtype=torch.float64)
    h = x @ w.t()
    e = torch.nn.functional.leaky_relu((h @ a_d)[:, None] + (h @ a_s)[None, :], 0.2)   # [target, source]
    mask = (dense_adj(ei, n, drop_loops=True) + torch.eye(n, dtype=torch.float64)) > 0
    e = e.masked_fill(~mask, -float('inf'))
    want = torch.softmax(e, dim=1) @ h + b
    got = pyg.gat_conv(x, ei, w, a_s.view(1, 1, -1), a_d.view(1, 1, -1), b)
    assert torch.allclose(got, want, atol=1e-10)


def test_gat_multi_edges_count_twice_in_softmax():
    ei = torch.tensor([[1, 1, 2], [0, 0, 0]])
    x = torch.tensor([[0.0], [1.0], [2.0]], dtype=torch.float64)
    one = torch.ones(1, 1, dtype=torch.float64)
    got = pyg.gat_conv(x, ei, one, torch.zeros(1, 1, 1, dtype=torch.float64),
                       torch.zeros(1, 1, 1, dtype=torch.float64), None)
    # uniform attention over {1, 1, 2, self}: (1 + 1 + 2 + 0) / 4
    assert torch.allclose(got[0], torch.tensor([1.0], dtype=torch.float64))


def test_gin_conv_is_adj_plus_identity_then_linear():
    n, f, o = 11, 4, 3
    ei = rand_graph(n, 25, 3)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, f, generator=g, dtype=torch.float64)
    w = torch.randn(o, f, generator=g, dtype=torch.float64)
    b = torch.randn(o, generator=g, dtype=torch.float64)
    want = ((dense_adj(ei, n) + torch.eye(n, dtype=torch.float64)) @ x) @ w.t() + b   # loops kept, no norm
    assert torch.allclose(pyg.gin_conv(x, ei, w, b), want, atol=1e-12)


@pytest.mark.parametrize('blocks', [None, 2])
def test_rgcn_conv_is_rownormalised_per_relation(blocks):
    n, f, o, r = 9, 4, 6, 3
    g = torch.Generator().manual_seed(11)
    ei = torch.randint(0, n, (2, 30), generator=g)
    et = torch.randint(0, r, (30,), generator=g)
    x = torch.randn(n, f, generator=g, dtype=torch.float64)
    root = torch.randn(f, o, generator=g, dtype=torch.float64)
    b = torch.randn(o, generator=g, dtype=torch.float64)
    if blocks:
        wt = torch.randn(r, blocks, f // blocks, o // blocks, generator=g, dtype=torch.float64)
        dense_w = [torch.block_diag(*[wt[k, j] for j in range(blocks)]) for k in range(r)]
    else:
        wt = torch.randn(r, f, o, generator=g, dtype=torch.float64)
        dense_w = [wt[k] for k in range(r)]
    want = x @ root + b
    for k in range(r):
        a = dense_adj(ei[:, et == k], n)
        a = a / a.sum(1).clamp(min=1)[:, None]
        want = want + a @ x @ dense_w[k]
    got = pyg.rgcn_conv(x, ei, et, wt, root, b, blocks)
    assert torch.allclose(got, want, atol=1e-12)


def test_k_hop_subgraph_equals_boolean_reachability():
    n = 14
    g = torch.Generator().manual_seed(4)
    ei = torch.randint(0, n, (2, 22), generator=g)
    a = dense_adj(ei, n) > 0                           # a[target, source]
    seeds = torch.tensor([3, 7])
    reach = torch.zeros(n, dtype=torch.bool)
    reach[seeds] = True
    frontier = reach.clone()
    for hops in (1, 2, 3):
        frontier = (a[frontier].any(0))                # sources of edges whose target is in the frontier
        reach = reach | frontier
        subset, sub_ei, mask = pyg.k_hop_subgraph(seeds, hops, ei, n)
        assert torch.equal(subset, reach.nonzero().flatten())
        assert torch.equal(mask, reach[ei[0]] & reach[ei[1]])
        assert torch.equal(sub_ei, ei[:, mask])


def test_k_hop_walks_only_towards_sources_on_a_path():
    ei = torch.tensor([[0, 1, 2], [1, 2, 3]])          # 0->1->2->3
    subset, _, mask = pyg.k_hop_subgraph(torch.tensor([2]), 1, ei, 4)
    assert subset.tolist() == [1, 2] and mask.tolist() == [False, True, False]
    subset, _, _ = pyg.k_hop_subgraph(torch.tensor([2]), 2, ei, 4)
    assert subset.tolist() == [0, 1, 2]


def test_to_undirected_sorts_and_sums_attributes():
    ei = torch.tensor([[0, 2, 1], [3, 3, 2]])
    und, (a,) = pyg.to_undirected(ei, [torch.tensor([1, 0, 1])], 4)
    assert und.tolist() == [[0, 1, 2, 2, 3, 3], [3, 2, 1, 3, 0, 2]]
    assert a.tolist() == [1, 1, 1, 0, 1, 0]
    both = torch.tensor([[0, 1], [1, 0]])              # already symmetric -> attrs add up
    und, (a,) = pyg.to_undirected(both, [torch.tensor([1, 1])], 2)
    assert und.tolist() == [[0, 1], [1, 0]] and a.tolist() == [2, 2]
    assert pyg.is_undirected(und, 2) and not pyg.is_undirected(ei, 4)


def test_module_state_dict_keys_are_pygs():
    import torch.nn as nn
    assert set(pyg.GCNConv(3, 2).state_dict()) == {'lin.weight', 'bias'}
    assert set(pyg.GATConv(3, 2).state_dict()) == {'lin_src.weight', 'lin_dst.weight', 'att_src', 'att_dst', 'bias'}
    assert set(pyg.GINConv(nn.Linear(3, 2)).state_dict()) == {'nn.weight', 'nn.bias'}
    assert set(pyg.RGCNConv(4, 4, 6, 2).state_dict()) == {'weight', 'root', 'bias'}
    assert pyg.RGCNConv(4, 4, 6, 2).weight.shape == (6, 2, 2, 2)


def test_sage_conv_is_row_mean_plus_root():
    n, f, o = 12, 5, 3
    ei = rand_graph(n, 28, 9)
    g = torch.Generator().manual_seed(6)
    x = torch.randn(n, f, generator=g, dtype=torch.float64)
    wl = torch.randn(o, f, generator=g, dtype=torch.float64)
    wr = torch.randn(o, f, generator=g, dtype=torch.float64)
    bl = torch.randn(o, generator=g, dtype=torch.float64)
    a = dense_adj(ei, n)
    a = a / a.sum(1).clamp(min=1)[:, None]
    want = (a @ x) @ wl.t() + bl + x @ wr.t()
    assert torch.allclose(pyg.sage_conv(x, ei, wl, bl, wr), want, atol=1e-12)
    assert set(pyg.SAGEConv(3, 2).state_dict()) == {'lin_l.weight', 'lin_l.bias', 'lin_r.weight'}
