"""Host-side preprocessing of gnndelete_amd.framework vs the oracle and vs what the reference's
delete_gnn.main() produced (tests/golden/prep_*.npz).  Bit-exact (integer / boolean work)."""
import os

import pytest
import torch

from helpers import load_golden, t
from oracle import gnndelete_ref as R
from oracle import pyg_semantics as pyg

from gnndelete_amd.framework import graph_utils as G
from gnndelete_amd.framework.data import Data, prepare_edge_deletion, resolve_df_size
from gnndelete_amd.framework.utils import get_link_labels, negative_sampling_kg


@pytest.mark.parametrize('name', ['prep_gcn_out.npz', 'prep_gcn_in.npz', 'prep_gat_out.npz', 'prep_gat_in.npz'])
def test_prepare_edge_deletion_reproduces_reference_main(name):
    fx = load_golden(name)
    E, n = t(fx['in::train']), int(fx['in::num_nodes'])
    data = Data(train_pos_edge_index=E, num_nodes=n)
    torch.manual_seed(int(fx['in::seed']))
    size = resolve_df_size(float(fx['in::df_size']), E.shape[1])
    prepare_edge_deletion(data, t(fx['in::cand']), size)
    for k in ['train_pos_edge_index', 'df_mask', 'dr_mask', 'sdf_mask', 'sdf_node_1hop_mask', 'sdf_node_2hop_mask',
              'directed_df_edge_index']:
        assert torch.equal(data[k], t(fx[f'out::{k}'])), k
    assert data.edge_index is data.train_pos_edge_index


def test_prepare_edge_deletion_relational_branch():
    g = torch.Generator().manual_seed(0)
    n, m, r = 40, 120, 3
    lo = torch.randint(0, n - 1, (m,), generator=g)
    E = torch.unique(torch.stack([lo, lo + 1 + torch.randint(0, 5, (m,), generator=g)]).clamp(max=n - 1), dim=1)
    E = E[:, E[0] < E[1]]
    et = torch.randint(0, r, (E.shape[1],), generator=g)
    data = Data(train_pos_edge_index=E, train_edge_type=et, num_nodes=n)
    torch.manual_seed(1)
    prepare_edge_deletion(data, torch.ones(E.shape[1], dtype=torch.bool), 7, relational=True, num_edge_type=r)
    m1 = E.shape[1]
    assert data.edge_index.shape[1] == 2 * m1 and torch.equal(data.edge_index[:, m1:], E.flip(0))
    assert torch.equal(data.edge_type[m1:], et + r)
    assert int(data.df_mask.sum()) == 14 and torch.equal(data.df_mask[:m1], data.df_mask[m1:])
    assert torch.equal(data.dr_mask, ~data.df_mask)
    assert torch.equal(data.directed_df_edge_type, et[data.df_mask[:m1]])


@pytest.mark.parametrize('seed', range(4))
def test_k_hop_and_to_undirected_match_oracle(seed):
    g = torch.Generator().manual_seed(seed)
    n = 50
    ei = torch.randint(0, n, (2, 160), generator=g)
    seeds = torch.randperm(n, generator=g)[:4]
    for hops in (1, 2, 3):
        s0, e0, m0 = pyg.k_hop_subgraph(seeds, hops, ei, n)
        s1, e1, inv, m1 = G.k_hop_subgraph(seeds, hops, ei, num_nodes=n)
        assert torch.equal(s0, s1) and torch.equal(e0, e1) and torch.equal(m0, m1)
        assert torch.equal(s1[inv], seeds)
    a = torch.randint(0, 2, (160,), generator=g).int()
    u0, (a0,) = pyg.to_undirected(ei, [a], n)
    u1, [a1] = G.to_undirected(ei, [a], n)
    assert torch.equal(u0, u1) and torch.equal(a0, a1)
    assert torch.equal(G.to_undirected(ei, num_nodes=n), u0)
    assert G.is_undirected(u1, n) and (G.is_undirected(ei, n) == pyg.is_undirected(ei, n))


def test_negative_sampling_contract():
    n = 30
    ei = torch.randint(0, n, (2, 200), generator=torch.Generator().manual_seed(3))
    neg = G.negative_sampling(ei, n, 150)
    assert neg.shape == (2, 150) and neg.dtype == torch.long
    pos = set((ei[0] * n + ei[1]).tolist())
    assert not (set((neg[0] * n + neg[1]).tolist()) & pos)
    assert bool((neg[0] != neg[1]).all())
    assert G.negative_sampling(ei, n, 0).shape == (2, 0)


def test_negative_sampling_kg_matches_reference_golden():
    fx = load_golden('neg_kg.npz')
    torch.manual_seed(int(fx['seed']))
    assert torch.equal(negative_sampling_kg(t(fx['edge_index']), t(fx['edge_type'])), t(fx['neg']))


def test_labels_and_data_bag():
    lab = get_link_labels(torch.zeros(2, 3, dtype=torch.long), torch.zeros(2, 2, dtype=torch.long))
    assert lab.tolist() == [1, 1, 1, 0, 0]
    d = Data(x=torch.zeros(2, 2), num_nodes=2)
    assert not hasattr(d, 'dtrain_mask') and d['x'] is d.x
    d.foo = torch.ones(1)
    assert 'foo' in d and d.to('cpu') is d
    assert R.df_size_from_arg(2.5, 1000) == resolve_df_size(2.5, 1000) == 25
    assert resolve_df_size(100, 1000) == 100


def test_raw_readers_and_prepare_dataset_on_files_in_the_upstream_layouts(tmp_path):
    """prepare_dataset.py on raw files laid out like the reference's downloads (CitationFull .npz,
    ogbl-collab csv.gz): self loops and duplicate / reversed edges collapse to unique row<col pairs, features
    are binarised and row-normalised, and the result goes through the reference's split recipe."""
    import gzip
    import subprocess
    import sys
    import numpy as np
    import scipy.sparse as sp
    import torch
    from gnndelete_amd.framework.raw_readers import load_raw
    rng = np.random.default_rng(0)
    n, f = 120, 30
    a = sp.random(n, n, density=0.05, random_state=1, format='csr')
    a = a + a.T + sp.eye(n)                                   # symmetric with self loops, like the raw files
    a = sp.csr_matrix(a)
    x = sp.random(n, f, density=0.2, random_state=2, format='csr')
    raw = tmp_path / 'data' / 'DBLP' / 'raw'
    raw.mkdir(parents=True)
    np.savez(raw / 'dblp.npz', adj_data=a.data, adj_indices=a.indices, adj_indptr=a.indptr, adj_shape=a.shape,
             attr_data=x.data, attr_indices=x.indices, attr_indptr=x.indptr, attr_shape=x.shape,
             labels=rng.integers(0, 4, n))
    xr, edges, y = load_raw('DBLP', str(tmp_path / 'data'))
    dense = np.asarray(a.todense()) != 0
    np.fill_diagonal(dense, False)
    want = torch.from_numpy(np.stack(np.nonzero(np.triu(dense, 1))))
    assert torch.equal(edges, want) and y.shape == (n,)
    xd = (np.asarray(x.todense()) > 0).astype(np.float32)
    np.testing.assert_allclose(xr.numpy(), xd / np.maximum(xd.sum(1, keepdims=True), 1), rtol=1e-6)

    craw = tmp_path / 'data' / 'ogbl_collab' / 'raw'
    craw.mkdir(parents=True)
    e = rng.integers(0, n, (400, 2))
    with gzip.open(craw / 'edge.csv.gz', 'wt') as fh:
        fh.write('\n'.join(f'{u},{v}' for u, v in e))
    feat = rng.standard_normal((n, 8)).astype(np.float32)
    with gzip.open(craw / 'node-feat.csv.gz', 'wt') as fh:
        fh.write('\n'.join(','.join(repr(float(v)) for v in row) for row in feat))
    xc, ec, _ = load_raw('ogbl-collab', str(tmp_path / 'data'))
    keys = {(min(u, v), max(u, v)) for u, v in e.tolist() if u != v}
    assert ec.shape[1] == len(keys) and bool((ec[0] < ec[1]).all()) and xc.shape == (n, 8)

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'prepare_dataset.py'), '--dataset', 'DBLP', '--data_dir',
                        str(tmp_path / 'data'), '--seeds', '42'], capture_output=True, text=True,
                       env=dict(os.environ, PYTHONPATH=root))
    assert r.returncode == 0, r.stderr[-800:]
    d = torch.load(tmp_path / 'data' / 'DBLP' / 'd_42.pt')
    m = edges.shape[1]
    assert d['train_pos_edge_index'].shape[1] == m - 2 * int(0.05 * m) and d['val_pos_edge_index'].shape[1] == int(0.05 * m)
    r = subprocess.run([sys.executable, os.path.join(root, 'prepare_dataset.py'), '--dataset', 'PubMed', '--data_dir',
                        str(tmp_path / 'data')], capture_output=True, text=True, env=dict(os.environ, PYTHONPATH=root))
    assert r.returncode != 0 and 'no raw files' in r.stderr + r.stdout


@pytest.mark.parametrize('tag', ['plain', 'degree', 'kg'])
def test_dataset_split_reproduces_reference(tag):
    """synth.split_linkpred / split_kg vs train_test_split_edges_no_neg_adj_mask run from the reference's own
    prepare_dataset.py (tests/golden/split.npz): train / validation / test positives, the KG type slicing and
    negatives, and the IN candidate mask of :205-214.  (Non-KG negatives come from torch_geometric's
    negative_sampling upstream - a different sampler with the same contract, checked separately.)"""
    from gnndelete_amd.framework.synth import split_kg, split_linkpred
    fx = load_golden('split.npz')
    ei, n, seed = t(fx[f'{tag}::edge_index']), int(fx[f'{tag}::num_nodes']), int(fx[f'{tag}::seed'])
    if tag == 'kg':
        torch.manual_seed(seed)
        data, masks = split_kg(torch.arange(n), ei, t(fx[f'{tag}::edge_type']), n, 0.05, 0.05)
        for k, f in [('train_edge_type', 'train_type'), ('val_edge_type', 'val_type'), ('test_edge_type', 'test_type'),
                     ('val_neg_edge_index', 'val_neg'), ('test_neg_edge_index', 'test_neg')]:
            assert torch.equal(data[k], t(fx[f'{tag}::{f}'])), k
    else:
        thd = t(fx[f'{tag}::two_hop_degree']) if f'{tag}::two_hop_degree' in fx else None
        gen = torch.Generator().manual_seed(seed)
        data, masks = split_linkpred(torch.zeros(n, 1), ei[:, ei[0] < ei[1]], n, gen, 0.05, 0.05, two_hop_degree=thd)
        for stage in ('val', 'test'):
            neg, pos = data[f'{stage}_neg_edge_index'], data[f'{stage}_pos_edge_index']
            assert neg.shape == pos.shape and bool((neg[0] != neg[1]).all())
    assert torch.equal(data.train_pos_edge_index, t(fx[f'{tag}::train']))
    assert torch.equal(data.val_pos_edge_index, t(fx[f'{tag}::val']))
    assert torch.equal(data.test_pos_edge_index, t(fx[f'{tag}::test']))
    assert torch.equal(masks['in'], t(fx[f'{tag}::in_mask'])) and torch.equal(masks['out'], ~t(fx[f'{tag}::in_mask']))


def test_ogbl_biokg_reader_reproduces_reference_process_kg(tmp_path):
    """gnndelete_amd.framework.raw_readers: OGB's on-disk layout of ogbl-biokg (split/random/*.pt dicts with per-type
    local ids, raw/num-node-dict.csv.gz) -> the Data fields and Df candidate masks the reference's process_kg pickles
    (tests/golden/process_kg.npz, made by the reference's own function); prepare_dataset.py writes them."""
    import gzip
    import subprocess
    import sys
    from helpers import load_golden, t
    from test_oracle_golden import _ogb_split
    from gnndelete_amd.framework import raw_readers
    from gnndelete_amd.framework.data import Data
    fx = load_golden('process_kg.npz')
    split, nodes = _ogb_split(fx)
    root = tmp_path / 'ogbl_biokg'
    (root / 'raw').mkdir(parents=True)
    (root / 'split' / 'random').mkdir(parents=True)
    with gzip.open(root / 'raw' / 'num-node-dict.csv.gz', 'wt') as f:
        f.write(','.join(nodes) + '\n' + ','.join(str(v) for v in nodes.values()) + '\n')
    for k, d in split.items():
        torch.save(d, root / 'split' / 'random' / f'{k}.pt')
    data, df = raw_readers.load_raw_kg('ogbl-biokg', str(tmp_path))
    for k in ('x', 'edge_index', 'edge_type', 'train_pos_edge_index', 'train_edge_type', 'val_pos_edge_index', 'val_edge_type',
              'val_neg_edge_index', 'test_pos_edge_index', 'test_edge_type', 'test_neg_edge_index'):
        assert torch.equal(data[k], t(fx[f'out::{k}'])), k
    assert torch.equal(df['in'], t(fx['out::in_mask'])) and torch.equal(df['out'], t(fx['out::out_mask']))
    assert raw_readers.load_raw_kg('ogbl-biokg', str(tmp_path / 'nowhere')) is None
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run([sys.executable, os.path.join(repo, 'prepare_dataset.py'), '--dataset', 'ogbl-biokg', '--data_dir', str(tmp_path),
                    '--seeds', '42'], check=True, capture_output=True)
    saved = Data.load(str(tmp_path / 'ogbl-biokg' / 'd_42.pt'))
    assert torch.equal(saved.train_pos_edge_index, t(fx['out::train_pos_edge_index']))
    assert torch.equal(torch.load(tmp_path / 'ogbl-biokg' / 'df_42.pt')['in'], t(fx['out::in_mask']))


def test_rgcn_wave_plan_reproduces_the_typed_mean_aggregation():
    """TypedNodeCSR.wave_plan (pure torch, runs on the CPU): walking the unit plan the way gd_rgcn_wave_conv_f32 does - 16 slots x 4
    (source, weight) pairs per unit, unused pairs pointing one row past x, the segmented scan over the slots with the plan's
    same-node flags, only a node's LAST slot added to its accumulator row - gives the per-(node, relation) weighted sums of
    the node-major arrays; every typed edge sits in exactly one pair; the unit arrays end with the empty unit; the scan-step
    mask of a unit is the OR of its slots' flags.  A hub node (runs of many slots, across units) and an empty graph included."""
    import torch
    from gnndelete_amd.graph import TypedNodeCSR
    g = torch.Generator().manual_seed(0)
    n, m, R = 300, 20000, 7
    ei = torch.randint(0, n, (2, m), generator=g)
    et = torch.randint(0, R, (m,), generator=g)
    ei[1, :m // 5] = 5
    et[:m // 10] = 2
    tg = TypedNodeCSR(ei, et, n, R)
    for trans in (False, True):
        p = tg.wave_plan(trans)
        node_ptr, seg_ptr, seg_rel, col, w = tg.bwd if trans else tg.fwd
        assert p['n_units'] == int(p['tile_unit_ptr'][-1]) and p['unit_rel'].numel() == p['n_units'] + 1
        assert int((p['unit_edges'][..., 0] != n).sum()) == m
        assert int(p['unit_row'][-1].abs().sum()) == 0 and bool((p['unit_edges'][-1, :, :, 0] == n).all())
        x = torch.randn(n + 1, 8, dtype=torch.float64, generator=g)
        x[n] = 0
        runs = (seg_ptr[1:] - seg_ptr[:-1]).long()
        run_of_edge = torch.repeat_interleave(torch.arange(runs.numel()), runs)
        node_of_run = torch.repeat_interleave(torch.arange(n), (node_ptr[1:] - node_ptr[:-1]).long())
        want = torch.zeros(n, R, 8, dtype=torch.float64)
        want.index_put_((node_of_run[run_of_edge], seg_rel.long()[run_of_edge]), w.double()[:, None] * x[col.long()], accumulate=True)
        got = torch.zeros(n, R, 8, dtype=torch.float64)
        tup, T = p['tile_unit_ptr'], p['tile']
        for t in range(p['n_tiles']):
            for u in range(int(tup[t]), int(tup[t + 1])):
                word_u, ed, wd = int(p['unit_rel'][u]), p['unit_edges'][u], p['unit_row'][u].long()
                v = (x[ed[:, :, 0].long()] * ed[:, :, 1].contiguous().view(torch.float32).double()[..., None]).sum(1)
                steps = 0
                for b in range(4):
                    sh = 1 << b
                    sv = torch.zeros_like(v)
                    sv[sh:] = v[:-sh]
                    flag = (wd >> (8 + b)) & 1
                    steps |= int(flag.max()) << b
                    v = v + flag.double()[:, None] * sv
                assert word_u >> 16 == steps
                for q in range(16):
                    if (int(wd[q]) >> 12) & 1:
                        got[t * T + (int(wd[q]) & 255), word_u & 0xffff] += v[q]
        assert float((got - want).abs().max()) < 1e-12
    empty = TypedNodeCSR(torch.zeros(2, 0, dtype=torch.long), torch.zeros(0, dtype=torch.long), 100, 3).wave_plan(False)
    assert empty['n_units'] == 0 and bool((empty['unit_edges'][..., 0] == 100).all())


def test_bench_attaches_a_stage_profile_only_to_the_kernels_it_was_taken_with(tmp_path):
    """bench.py's hygiene (host logic, no GPU): the stage table of the replayed step is stamped with a hash of the kernel sources
    and the C header; load_stage_profile hands it out only for the same workload AND the same sources and says why not otherwise;
    the committed tables of this tree (GCN / GAT / GraphSAGE) carry the hash of this tree; the fabric ceiling of the SpMM's
    roofline entry is computed from the committed floor / probe records (not a literal) and only for the graph they were taken
    on; the algorithmic bytes of SURVEY 8(d); a traffic-based stage table for models without a hand-written one."""
    import json
    import os
    import bench
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sha = bench.kernel_source_hash()
    assert len(sha) == 16 and sha == bench.kernel_source_hash()
    # the tables of an EARLIER round's kernels are refused, whatever workload they name
    for name in ('r05_final_stages.json', 'r05_final_stages_gat.json', 'r05_final_stages_sage.json'):
        old_rec = json.load(open(os.path.join(root, 'profiles', name)))
        got, why = bench.load_stage_profile(os.path.join(root, 'profiles', name), old_rec['workload']['num_nodes'], old_rec['workload']['spmm_nnz'])
        assert (got == {} and 'stale' in why) or old_rec['csrc_sha'] == sha
    # this round's tables (what bench.py attaches by default) carry the hash of this tree
    current = [n_ for n_ in ('r06_final_stages.json', 'r06_final_stages_gat.json', 'r06_final_stages_sage.json', 'r06_final_stages_collab_nodecls_gat.json')
               if os.path.exists(os.path.join(root, 'profiles', n_))]
    for name in current or ['r05_final_stages.json']:
        path = os.path.join(root, 'profiles', name)
        rec = json.load(open(path))
        if not current:                      # (no table of this round committed yet: exercise the logic on a re-stamped copy)
            rec = dict(rec, csrc_sha=sha)
            path = str(tmp_path / ('restamped_' + name))
            open(path, 'w').write(json.dumps(rec))
        assert rec['csrc_sha'] == sha, f'{name} was taken with other kernel sources: rerun tools/experiments/r06_profile.sh'
        n, nnz = rec['workload']['num_nodes'], rec['workload']['spmm_nnz']
        got, why = bench.load_stage_profile(path, n, nnz)
        assert why is None and got['stages'] and all(v['in_step_us'] > 0 for v in got['stages'].values())
        got, why = bench.load_stage_profile(path, n + 1, nnz)
        assert got == {} and 'another workload' in why
        stale = dict(rec, csrc_sha='0' * 16)
        p2 = tmp_path / name
        p2.write_text(json.dumps(stale))
        got, why = bench.load_stage_profile(str(p2), n, nnz)
        assert got == {} and 'stale' in why and sha in why
        table = bench.stage_table_from_profile(rec)
        assert len(table) == len(rec['stages']) and all(e['in_step_us'] > 0 for e in table)
    got, why = bench.load_stage_profile(str(tmp_path / 'missing.json'), 1, 1)
    assert got == {} and 'no stage profile' in why
    fl = json.load(open(os.path.join(root, 'profiles', 'r02_spmm_traffic_floor.json')))
    c = bench.fabric_ceiling(fl['n'], fl['nnz'], fl['d'])
    assert 0.35 < c['frac'] < 0.5 and abs(c['frac'] - fl['algorithmic_bytes'] / fl['lru_bytes'] * c['fabric_tbs'] * 1e3 / bench.HBM_PEAK_GBS) < 1e-12
    assert bench.fabric_ceiling(fl['n'] + 1, fl['nnz'], fl['d']) is None
    assert bench.spmm_algorithmic_bytes(10, 50, 64) == 4 * 11 + 4 * 50 + 4 * 50 + 2 * 4 * 10 * 64


def test_cpu_oracle_job_in_a_child_process_reproduces_the_in_process_run():
    """tests/oracle_jobs.py: the CPU-oracle legs of the long GPU tests run as child processes (so that the GPU is not idle while
    they compute).  A child rebuilds its request from seeds: it must arrive at exactly the same weights / negatives (checksum)
    and at the oracle results of the same function run in this process (to the run-to-run noise of the oracle's threaded
    scatters: two runs in ONE process differ by ~5e-8 in the Del weights)."""
    import oracle_jobs as J
    from helpers import rel_l2
    J.start('selftest-small')
    kind, params = J.JOBS['selftest-small']
    here = J._run_linkpred(**params)
    there = J.result('selftest-small')
    assert here['checksum'] == there['checksum']
    for a, b in zip(here['logs'], there['logs']):
        assert abs(a['train_loss'] - b['train_loss']) <= 1e-6 * abs(a['train_loss'])
    assert rel_l2(there['w1'], here['w1']) < 1e-6 and rel_l2(there['w2'], here['w2']) < 1e-6


@pytest.mark.parametrize('gnn', ['gcn', 'gat', 'gin', 'sage'])
def test_padded_class_dimension_is_exact_in_the_oracle_arithmetic(gnn):
    """engine._padded_out_shadow (round 6): layer 2 of a node-classification model (out_dim = #classes = 4) widened to 64 output
    columns by ZERO columns.  Host-side check of the claim the engine relies on, in the oracle's own conv arithmetic (no HIP):
    the padded conv2's first 4 output columns are the unpadded conv2's, the other 60 are exactly zero; the padded W_D2 carries the
    caller's block top-left and zeros elsewhere; Del-2 on the padded output keeps the padding at zero; conv1 / deletion1 are shared."""
    from types import SimpleNamespace
    from gnndelete_amd.engine import _padded_out_shadow
    from gnndelete_amd.framework import models as M
    torch.manual_seed(3)
    n, f, h, o, pad = 60, 12, 128, 4, 64
    ei = torch.randint(0, n, (2, 300))
    ei = ei[:, ei[0] != ei[1]]
    m1, m2 = torch.rand(n) < 0.5, torch.rand(n) < 0.8
    cls = {'gcn': M.GCNDelete, 'gat': M.GATDelete, 'gin': M.GINDelete, 'sage': M.SAGEDelete}[gnn]
    model = cls(SimpleNamespace(in_dim=f, hidden_dim=h, out_dim=o), m1, m2)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.startswith('conv2') and 'bias' in name:
                p.copy_(torch.randn_like(p) * 0.1)
        model.deletion2.deletion_weight.copy_(torch.eye(o) * 0.7 + 0.05 * torch.randn(o, o))
    sh = _padded_out_shadow(model, pad)
    assert sh.conv1 is model.conv1 and sh.deletion1 is model.deletion1 and sh.deletion2.mask is model.deletion2.mask
    x = torch.randn(n, h).clamp(min=0)                       # relu(z1)

    def conv2(c):
        if gnn == 'gcn':
            return pyg.gcn_conv(x, ei, c.lin.weight, c.bias)
        if gnn == 'gat':
            return pyg.gat_conv(x, ei, c.lin_src.weight, c.att_src, c.att_dst, c.bias, c.negative_slope)
        if gnn == 'gin':
            return pyg.gin_conv(x, ei, c.nn.weight, c.nn.bias, c.eps)
        return pyg.sage_conv(x, ei, c.lin_l.weight, c.lin_l.bias, c.lin_r.weight)
    with torch.no_grad():
        want, got = conv2(model.conv2), conv2(sh.conv2)
        assert got.shape == (n, pad) and float(got[:, o:].abs().max()) == 0.0
        assert torch.allclose(got[:, :o], want, rtol=1e-6, atol=1e-7)
        wp = sh.deletion2.deletion_weight
        assert torch.equal(wp[:o, :o], model.deletion2.deletion_weight) and float(wp[o:].abs().max()) == 0.0 and float(wp[:, o:].abs().max()) == 0.0
        z2 = got.clone()
        z2[m2] = got[m2] @ wp
        assert float(z2[:, o:].abs().max()) == 0.0
        # the gradient of any loss on the first o columns is zero in the padding block: p2^T dz2 with p2 = 0 / dz2 = 0 there
        dz2 = torch.zeros(n, pad)
        dz2[:, :o] = torch.randn(n, o)
        gw = got[m2].t() @ dz2[m2]
        assert float(gw[o:].abs().max()) == 0.0 and float(gw[:, o:].abs().max()) == 0.0
