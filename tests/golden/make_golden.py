#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REAL reference code.

Runs only in the build container (needs /root/reference); the outputs (*.npz / *.json,
inputs AND expected outputs, a few hundred KB in total) are committed and are what
travels to the GPU box.  No reference source is copied: the reference is imported in
place, under stubs for the third-party modules this image lacks.

How the import works:
  * torch_geometric / torch_scatter / torch_sparse / wandb / ogb are replaced by stub
    modules; the conv classes the reference instantiates (GCNConv, GATConv, GINConv,
    RGCNConv) are oracle.pyg_semantics' restatements, so what these vectors pin is the
    code the reference OWNS (DeletionLayer, *Delete wiring, loss zoo, the trainer loop
    with every loss_type branch, Trainer.eval, parse_args, negative_sampling_kg, the
    delete_gnn.py preprocessing) - not PyG's arithmetic (pinned by closed-form KATs).
  * framework/__init__.py is bypassed with synthetic package objects because it
    imports modules that do not exist upstream (graph_editor, MIAttackTrainerNode).
  * PyG's negative_sampling draws from Python's `random`; it is stubbed to return the
    negatives stored in the fixture.

Usage:  python tests/golden/make_golden.py        (rewrites every fixture)
"""
import importlib
import json
import os
import pickle
import runpy
import sys
import tempfile
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, ROOT)

from oracle import pyg_semantics as pyg  # noqa: E402

STATE = {'neg': None, 'wandb': [], 'neg_gen': None, 'neg_log': [], 'batches': []}


# --------------------------------------------------------------------------- stubs
def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _neg_stub(edge_index=None, num_nodes=None, num_neg_samples=None, **kw):
    if STATE.get('neg_gen') is not None:
        # mini-batch loops draw fresh negatives per batch: drawn here from a seeded generator and RECORDED, so the
        # fixture holds every set the reference's loop consumed, in order
        k = int(edge_index.shape[1] if num_neg_samples is None else num_neg_samples)      # PyG default: one per edge
        neg = torch.randint(0, int(num_nodes), (2, k), generator=STATE['neg_gen'])
        STATE['neg_log'].append(neg.clone())
        return neg
    neg = STATE['neg']
    assert neg is not None and neg.shape[1] == int(num_neg_samples), (neg.shape, num_neg_samples)
    return neg.clone()


def saint_batch(data, node_idx):
    """What torch_geometric's GraphSAINTSampler hands the loop for one sampled node set [PyG-mem]: the induced
    subgraph in the adjacency's (row, col) order with relabelled endpoints, node-sized tensors sliced by the
    (sorted) node ids, edge-sized tensors by the kept edges, everything else passed through."""
    n, ei = int(data['num_nodes']), data['edge_index']
    e = ei.shape[1]
    member = torch.zeros(n, dtype=torch.bool)
    member[node_idx] = True
    order = torch.argsort(ei[0] * n + ei[1], stable=True)          # SparseTensor(row, col, value=arange(E)) order
    keep = order[(member[ei[0]] & member[ei[1]])[order]]
    relabel = torch.full((n,), -1, dtype=torch.long)
    relabel[node_idx] = torch.arange(node_idx.numel())
    b = Bag(num_nodes=int(node_idx.numel()), edge_index=relabel[ei[:, keep]])
    for k, v in data.items():
        if k in ('edge_index', 'num_nodes'):
            continue
        if torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == n:
            b[k] = v[node_idx]
        elif torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == e:
            b[k] = v[keep]
        else:
            b[k] = v
    return b


class _SaintStub:
    """GraphSAINTRandomWalkSampler with INJECTED node sets (STATE['batches']): the sampler's random stream cannot
    be reproduced without torch_sparse, the loop that consumes its batches can."""

    def __init__(self, data, batch_size=None, walk_length=None, num_steps=None, **kw):
        self.data = data
        assert num_steps == len(STATE['batches']), (num_steps, len(STATE['batches']))

    def __len__(self):
        return len(STATE['batches'])

    def __iter__(self):
        for nodes in STATE['batches']:
            yield saint_batch(self.data, nodes)


def _khop_stub(node_idx, num_hops, edge_index, relabel_nodes=False, num_nodes=None, **kw):
    subset, ei, mask = pyg.k_hop_subgraph(node_idx, num_hops, edge_index, num_nodes)
    return subset, ei, None, mask


def _to_undirected_stub(edge_index, edge_attr=None, num_nodes=None, reduce='add'):
    n = int(edge_index.max()) + 1 if num_nodes is None else num_nodes
    if edge_attr is None:
        return pyg.to_undirected(edge_index, [], n)[0]
    return pyg.to_undirected(edge_index, list(edge_attr), n)


def _is_undirected_stub(edge_index, *a, **kw):
    return pyg.is_undirected(edge_index, int(edge_index.max()) + 1)


def _seed_everything(seed):
    import random
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


class Bag(dict):
    """Attribute bag standing in for torch_geometric.data.Data (AttributeError on a
    missing key so hasattr(data, 'dtrain_mask') is False, base.py:238)."""
    __setattr__ = dict.__setitem__

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def to(self, *a, **k):
        return self

    def cpu(self):
        return self

    def __getstate__(self):
        return dict(self)

    def __setstate__(self, s):
        self.update(s)


class FakeDataset:
    def __init__(self, num_features):
        self.num_features = num_features

    def __repr__(self):
        return 'SynthDataset()'


class _MessagePassing(nn.Module):
    """Just enough of torch_geometric's MessagePassing for the reference's own RGATConv (rgat.py:24-351) to
    run: gather x_i / x_j along the edges (flow source -> target), call message(), sum per target, update()."""

    def __init__(self, aggr='add', node_dim=0, **kwargs):
        super().__init__()
        assert aggr == 'add' and node_dim == 0

    def propagate(self, edge_index, size=None, **kwargs):
        x = kwargs['x']
        src, dst = edge_index[0], edge_index[1]
        msg = self.message(x_i=x[dst], x_j=x[src], edge_type=kwargs.get('edge_type'), edge_attr=kwargs.get('edge_attr'),
                           index=dst, ptr=None, size_i=x.shape[0])
        out = torch.zeros((x.shape[0],) + tuple(msg.shape[1:]), dtype=msg.dtype)
        out.index_add_(0, dst, msg)
        return self.update(out)


def _pyg_softmax(src, index, ptr=None, num_nodes=None):
    n = int(num_nodes) if num_nodes is not None else int(index.max()) + 1
    flat = src.reshape(src.shape[0], -1)
    cols = [pyg.segment_softmax(flat[:, c], index, n) for c in range(flat.shape[1])]
    return torch.stack(cols, 1).reshape(src.shape)


def _glorot(t):
    if t is not None:
        pyg.glorot_(t.data)


def _fill(v):
    def f(t):
        if t is not None:
            t.data.fill_(v)
    return f


def install_stubs():
    _mod('torch_geometric')
    _mod('torch_geometric.nn', GCNConv=pyg.GCNConv, GATConv=pyg.GATConv, GINConv=pyg.GINConv,
         RGCNConv=pyg.RGCNConv, FastRGCNConv=pyg.RGCNConv)
    _mod('torch_geometric.nn.conv', MessagePassing=_MessagePassing)
    _mod('torch_geometric.nn.dense')
    _mod('torch_geometric.nn.dense.linear', Linear=nn.Linear)
    _mod('torch_geometric.nn.inits', glorot=_glorot, ones=_fill(1.0), zeros=_fill(0.0))
    _mod('torch_geometric.typing', Adj=object, OptTensor=object, Size=object, OptPairTensor=object)
    _mod('torch_geometric.utils', softmax=_pyg_softmax, negative_sampling=_neg_stub, k_hop_subgraph=_khop_stub,
         to_undirected=_to_undirected_stub, is_undirected=_is_undirected_stub, to_networkx=None)
    _mod('torch_geometric.loader', GraphSAINTRandomWalkSampler=_SaintStub)
    _mod('torch_geometric.data', DataLoader=None, Data=Bag)
    _mod('torch_geometric.seed', seed_everything=_seed_everything)
    _mod('torch_scatter', scatter_add=None)
    _mod('torch_sparse', SparseTensor=object)
    _mod('wandb', log=lambda d, *a, **k: STATE['wandb'].append(dict(d)), init=lambda *a, **k: None,
         watch=lambda *a, **k: None)
    _mod('ogb')
    _mod('ogb.graphproppred', Evaluator=None)
    _mod('ogb.linkproppred', PygLinkPropPredDataset=None)
    _mod('torch_geometric.transforms', NormalizeFeatures=None)
    _mod('torch_geometric.datasets', CitationFull=None, Coauthor=None, Flickr=None, RelLinkPredDataset=None,
         WordNet18=None, WordNet18RR=None)
    sys.modules['torch_geometric.utils'].train_test_split_edges = None
    _mod('train_mi', MLPAttacker=None)
    for name, path in [('framework', f'{REF}/framework'), ('framework.trainer', f'{REF}/framework/trainer')]:
        pkg = types.ModuleType(name)
        pkg.__path__ = [path]
        sys.modules[name] = pkg


def load_reference():
    install_stubs()
    D = importlib.import_module('framework.models.deletion')
    T = importlib.import_module('framework.trainer.gnndelete_nodeemb')
    TE = importlib.import_module('framework.trainer.gnndelete')
    A = importlib.import_module('framework.training_args')
    U = importlib.import_module('framework.utils')
    B = importlib.import_module('framework.trainer.base')
    return D, T, TE, A, U, B


# --------------------------------------------------------------------------- inputs
def np_(t):
    return t.detach().cpu().numpy().copy()


def synth_graph(n, m, f, seed, relations=0):
    """Small random simple graph: unique row<col edges, features, 90/5/5 split."""
    g = torch.Generator().manual_seed(seed)
    a = torch.randint(0, n, (4 * m,), generator=g)
    b = torch.randint(0, n, (4 * m,), generator=g)
    lo, hi = torch.minimum(a, b), torch.maximum(a, b)
    keep = lo != hi
    key = torch.unique(lo[keep] * n + hi[keep])
    key = key[torch.randperm(key.shape[0], generator=g)][:m]
    ei = torch.stack([key // n, key % n])
    nv = max(1, int(0.05 * ei.shape[1]))
    val, test, train = ei[:, :nv], ei[:, nv:2 * nv], ei[:, 2 * nv:]
    x = torch.randn(n, f, generator=g) * 0.5
    out = dict(x=x, num_nodes=n, train=train, val_pos=val, test_pos=test,
               val_neg=torch.randint(0, n, val.shape, generator=g),
               test_neg=torch.randint(0, n, test.shape, generator=g))
    if relations:
        out['train_type'] = torch.randint(0, relations, (train.shape[1],), generator=g)
    return out


def prepare_deletion(g, df_count, seed):
    """The delete_gnn.py:85-189 recipe on the synthetic graph (non-KG branch), written
    against the same stubs the reference main() gets - used for the trainer fixtures.
    (The real main() is exercised separately by golden_prep().)"""
    gen = torch.Generator().manual_seed(seed)
    E, n = g['train'], g['num_nodes']
    df_idx = torch.randperm(E.shape[1], generator=gen)[:df_count]
    df_mask = torch.zeros(E.shape[1], dtype=torch.bool)
    df_mask[df_idx] = True
    seeds = E[:, df_mask].flatten().unique()
    _, e2, m2 = pyg.k_hop_subgraph(seeds, 2, E, n)
    _, e1, _ = pyg.k_hop_subgraph(seeds, 1, E, n)
    s1 = torch.zeros(n, dtype=torch.bool)
    s2 = torch.zeros(n, dtype=torch.bool)
    s1[e1.flatten().unique()] = True
    s2[e2.flatten().unique()] = True
    und, (dfu, m2u) = pyg.to_undirected(E, [df_mask.int(), m2.int()], n)
    dfu, m2u = dfu.bool(), m2u.bool()
    neg = torch.randint(0, n, (2, int(dfu.sum())), generator=gen)
    d = Bag(x=g['x'], num_nodes=n, train_pos_edge_index=und, edge_index=und,
            val_pos_edge_index=g['val_pos'], val_neg_edge_index=g['val_neg'],
            test_pos_edge_index=g['test_pos'], test_neg_edge_index=g['test_neg'],
            df_mask=dfu, dr_mask=~dfu, sdf_mask=m2u, sdf_node_1hop_mask=s1, sdf_node_2hop_mask=s2,
            directed_df_edge_index=E[:, df_mask])
    return d, neg


def make_args(A, argv):
    old = sys.argv
    sys.argv = ['x'] + argv
    try:
        return A.parse_args()
    finally:
        sys.argv = old


def state_np(model):
    return {f'w::{k}': np_(v) for k, v in model.state_dict().items()}


def data_np(d, neg):
    out = {f'd::{k}': np_(v) for k, v in d.items() if torch.is_tensor(v)}
    out['d::num_nodes'] = np.int64(d['num_nodes'])
    out['neg'] = np_(neg)
    return out


# --------------------------------------------------------------------------- fixtures
def golden_del_layer(D):
    out = {}
    g = torch.Generator().manual_seed(1)
    for tag, n, d, frac in [('partial', 37, 16, 0.4), ('empty', 11, 8, 0.0), ('full', 9, 64, 1.0),
                            ('odd', 23, 12, 0.5)]:
        x = torch.randn(n, d, generator=g, requires_grad=True)
        mask = torch.rand(n, generator=g) < frac if 0 < frac < 1 else torch.full((n,), bool(frac))
        layer = D.DeletionLayer(d, mask)
        with torch.no_grad():
            layer.deletion_weight.copy_(torch.randn(d, d, generator=g) * 0.3)
        up = torch.randn(n, d, generator=g)
        y = layer(x)
        y.backward(up)
        out.update({f'{tag}::x': np_(x), f'{tag}::mask': np_(mask), f'{tag}::w': np_(layer.deletion_weight),
                    f'{tag}::up': np_(up), f'{tag}::y': np_(y), f'{tag}::gx': np_(x.grad),
                    f'{tag}::gw': np_(layer.deletion_weight.grad)})
    layer = D.DeletionLayer(4, None)          # no mask => identity, same object
    x = torch.randn(5, 4, generator=g)
    out['nomask::x'] = np_(x)
    out['nomask::y'] = np_(layer(x))
    out['init::w'] = np_(D.DeletionLayer(6, None).deletion_weight)
    np.savez_compressed(os.path.join(HERE, 'del_layer.npz'), **out)


def golden_losses(T):
    out = {}
    g = torch.Generator().manual_seed(2)
    a0 = torch.randn(29, 16, generator=g)
    b0 = torch.randn(29, 16, generator=g)
    out['a'], out['b'] = np_(a0), np_(b0)
    for name in ['mse_mean', 'mse_sum', 'kld_mean', 'kld_sum', 'cosine_mean', 'cosine_sum', 'linear_cka']:
        a = a0.clone().requires_grad_(True)
        v = T.get_loss_fct(name)(a, b0)
        v.backward()
        out[f'{name}::value'] = np_(v)
        out[f'{name}::grad'] = np_(a.grad)
    # rbf_cka: upstream's default sigma path needs `math`, which the module never imports (NameError) - the
    # explicit-sigma call is the one that can run
    a = a0.clone().requires_grad_(True)
    v = T.get_loss_fct('rbf_cka')(a, b0, sigma=2.0)
    v.backward()
    out['rbf_cka_sigma2::value'], out['rbf_cka_sigma2::grad'] = np_(v), np_(a.grad)
    np.savez_compressed(os.path.join(HERE, 'losses.npz'), **out)


GNN_CLASS = {'gcn': 'GCNDelete', 'gat': 'GATDelete', 'gin': 'GINDelete', 'rgcn': 'RGCNDelete'}


def build_ref_model(D, A, gnn, d, in_dim, seed, num_edge_type=None, hidden=32, out=16):
    args = make_args(A, ['--gnn', 'gcn', '--in_dim', str(in_dim), '--hidden_dim', str(hidden),
                         '--out_dim', str(out)])
    torch.manual_seed(seed)
    cls = getattr(D, GNN_CLASS[gnn])
    if gnn == 'rgcn':
        model = cls(args, d['num_nodes'], num_edge_type, d['sdf_node_1hop_mask'], d['sdf_node_2hop_mask'])
    else:
        model = cls(args, d['sdf_node_1hop_mask'], d['sdf_node_2hop_mask'])
    with torch.no_grad():                      # non-trivial biases / Del weights
        for n_, p in model.named_parameters():
            if n_.endswith('bias'):
                p.copy_(torch.randn_like(p) * 0.1)
            if 'deletion_weight' in n_:
                p.copy_(torch.eye(p.shape[0]) * 0.5 + torch.randn_like(p) * 0.05)
    return model, args


def golden_wiring(D, A):
    for gnn in ['gcn', 'gat', 'gin', 'rgcn']:
        R = 3 if gnn == 'rgcn' else 0
        g = synth_graph(60, 220, 12, seed=10 + len(gnn), relations=R)
        d, neg = prepare_deletion(g, 8, seed=5)
        out = {}
        if gnn == 'rgcn':
            # KG branch of delete_gnn.py:158-171: reverse edges get type + R, masks repeat(2)
            E = g['train']
            df = torch.zeros(E.shape[1], dtype=torch.bool)
            df[:8] = True
            ei = torch.cat([E, E.flip(0)], 1)
            et = torch.cat([g['train_type'], g['train_type'] + R])
            model, _ = build_ref_model(D, A, gnn, d, 12, seed=3, num_edge_type=R)
            x = torch.arange(60)
            z1, z2 = model(x, ei, et, return_all_emb=True)
            o1, o2 = model.get_original_embeddings(x, ei, et, return_all_emb=True)
            score = model.decode(z2, E, g['train_type'])
            out.update(edge_index=np_(ei), edge_type=np_(et), x=np_(x), dec_edge=np_(E),
                       dec_type=np_(g['train_type']), score=np_(score), num_edge_type=np.int64(R))
        else:
            model, _ = build_ref_model(D, A, gnn, d, 12, seed=3)
            ei = d['train_pos_edge_index'][:, d['sdf_mask']]
            z1, z2 = model(d['x'], ei, return_all_emb=True)
            o1, o2 = model.get_original_embeddings(d['x'], ei, return_all_emb=True)
            score = model.decode(z2, d['val_pos_edge_index'], d['val_neg_edge_index'])
            # per-call mask override path (used by the mini-batch trainers)
            alt1 = torch.rand(60, generator=torch.Generator().manual_seed(9)) < 0.3
            alt2 = torch.rand(60, generator=torch.Generator().manual_seed(8)) < 0.6
            a1, a2 = model(d['x'], ei, alt1, alt2, return_all_emb=True)
            out.update(edge_index=np_(ei), x=np_(d['x']), score=np_(score), alt1=np_(alt1), alt2=np_(alt2),
                       a1=np_(a1), a2=np_(a2), val_pos=np_(d['val_pos_edge_index']),
                       val_neg=np_(d['val_neg_edge_index']))
        out.update(state_np(model))
        out.update(mask1=np_(d['sdf_node_1hop_mask']), mask2=np_(d['sdf_node_2hop_mask']),
                   z1=np_(z1), z2=np_(z2), o1=np_(o1), o2=np_(o2))
        np.savez_compressed(os.path.join(HERE, f'wiring_{gnn}.npz'), **out)


def golden_trajectories(D, T, A):
    """The real train_fullbatch loop (gnndelete_nodeemb.py:108-349) for every loss_type."""
    cases = [('gat', 'both_layerwise'), ('gat', 'both_all'), ('gat', 'only2_layerwise'),
             ('gat', 'only2_all'), ('gat', 'only1'), ('gin', 'both_layerwise'),
             ('gcn', 'both_all'), ('gcn', 'only2_layerwise'), ('gcn', 'only1')]
    for gnn, loss_type in cases:
        g = synth_graph(80, 300, 10, seed=21)
        d, neg = prepare_deletion(g, 10, seed=6)
        model, _ = build_ref_model(D, A, gnn, d, 10, seed=4)
        with torch.no_grad():                 # start from the reference's own init
            model.deletion1.deletion_weight.fill_(1 / 1000)
            model.deletion2.deletion_weight.fill_(1 / 1000)
        init = state_np(model)
        model.to = lambda *a, **k: model      # trainer hard-codes .to('cuda')
        tmp = tempfile.mkdtemp()
        args = make_args(A, ['--gnn', gnn, '--unlearning_model', 'gnndelete_nodeemb', '--epochs', '6',
                             '--valid_freq', '3', '--loss_type', loss_type, '--checkpoint_dir', tmp,
                             '--dataset', 'Cora', '--lr', '0.01', '--alpha', '0.4'])
        if 'layerwise' in loss_type:
            opt = [torch.optim.Adam(model.deletion1.parameters(), lr=args.lr),
                   torch.optim.Adam(model.deletion2.parameters(), lr=args.lr)]
        else:
            opt = torch.optim.Adam([{'params': [p for n_, p in model.named_parameters() if 'del' in n_],
                                     'weight_decay': 0.0}], lr=args.lr)
        STATE['neg'], STATE['wandb'] = neg, []
        torch.manual_seed(77)                 # Trainer.eval draws 500 randperm subsets
        trainer = T.GNNDeleteNodeembTrainer(args)
        trainer.train_fullbatch(model, d, opt, args)
        steps = [w for w in STATE['wandb'] if 'Epoch' in w]
        vals = [w for w in STATE['wandb'] if 'val_loss' in w]
        out = dict(init)
        out.update(data_np(d, neg))
        out.update(train_loss=np.array([s['train_loss'] for s in steps]),
                   loss_r=np.array([s['loss_r'] for s in steps]),
                   loss_l=np.array([s['loss_l'] for s in steps]),
                   final_w1=np_(model.deletion1.deletion_weight), final_w2=np_(model.deletion2.deletion_weight),
                   val_dt_auc=np.array([v['val_dt_auc'] for v in vals]),
                   val_df_auc=np.array([v['val_df_auc'] for v in vals]),
                   val_loss=np.array([v['val_loss'] for v in vals]),
                   val_df_logit_mean=np.array([v['val_df_logit_mean'] for v in vals]),
                   lr=np.float64(args.lr), alpha=np.float64(args.alpha), epochs=np.int64(6),
                   eval_seed=np.int64(77))
        np.savez_compressed(os.path.join(HERE, f'traj_{gnn}_{loss_type}.npz'), **out)


def golden_edgeprob_trajectories(D, TE, A):
    """The real GNNDeleteTrainer.train_fullbatch loop (framework/trainer/gnndelete.py:138-309): N x N
    pair masks, per-epoch negatives (the stub returns the same injected set), sigmoid(z z^T) against
    logits_ori, 0.5 / 0.5 mix, single Adam with zero_grad after the step."""
    for gnn in ['gcn', 'gat']:
        g = synth_graph(90, 340, 10, seed=23)
        d, neg = prepare_deletion(g, 9, seed=8)
        model, _ = build_ref_model(D, A, gnn, d, 10, seed=5)
        with torch.no_grad():
            model.deletion1.deletion_weight.fill_(1 / 1000)
            model.deletion2.deletion_weight.fill_(1 / 1000)
            z_ori = model.get_original_embeddings(d['x'], d['train_pos_edge_index'][:, d['dr_mask']])
            logits_ori = z_ori @ z_ori.t()          # what Trainer.test stores as pred_proba.pt (base.py:288)
        init = state_np(model)
        model.to = lambda *a, **k: model
        tmp = tempfile.mkdtemp()
        args = make_args(A, ['--gnn', gnn, '--unlearning_model', 'gnndelete', '--epochs', '6', '--valid_freq', '3',
                             '--checkpoint_dir', tmp, '--dataset', 'Cora', '--lr', '0.01'])
        opt = torch.optim.Adam([{'params': [p for n_, p in model.named_parameters() if 'del' in n_],
                                 'weight_decay': 0.0}], lr=args.lr)
        STATE['neg'], STATE['wandb'] = neg, []
        torch.manual_seed(78)
        trainer = TE.GNNDeleteTrainer(args)
        trainer.train_fullbatch(model, d, opt, args, logits_ori=logits_ori)
        steps = [w for w in STATE['wandb'] if 'Epoch' in w]
        vals = [w for w in STATE['wandb'] if 'val_loss' in w]
        out = dict(init)
        out.update(data_np(d, neg))
        out.update(logits_ori=np_(logits_ori),
                   train_loss=np.array([s['train_loss'] for s in steps]),
                   loss_r=np.array([s['loss_r'] for s in steps]),
                   loss_l=np.array([s['loss_l'] for s in steps]),
                   final_w1=np_(model.deletion1.deletion_weight), final_w2=np_(model.deletion2.deletion_weight),
                   val_dt_auc=np.array([v['val_dt_auc'] for v in vals]),
                   val_df_auc=np.array([v['val_df_auc'] for v in vals]),
                   lr=np.float64(args.lr), epochs=np.int64(6), eval_seed=np.int64(78))
        np.savez_compressed(os.path.join(HERE, f'traj_edgeprob_{gnn}.npz'), **out)


def golden_original_training(B, A):
    """The real original-model loop Trainer.train_fullbatch (framework/trainer/base.py:75-142) on the
    reference's own GCN / GAT / GIN (framework/models/{gcn,gat,gin}.py): BCE-with-logits link prediction on
    the whole training graph with per-epoch negatives (the stub returns the same injected set), Adam."""
    models = {'gcn': importlib.import_module('framework.models.gcn').GCN,
              'gat': importlib.import_module('framework.models.gat').GAT,
              'gin': importlib.import_module('framework.models.gin').GIN}
    for gnn, cls in models.items():
        g = synth_graph(90, 340, 10, seed=29)
        E = g['train']
        und, _ = pyg.to_undirected(E, [torch.ones(E.shape[1], dtype=torch.int32)], g['num_nodes'])
        gen = torch.Generator().manual_seed(4)
        neg = torch.randint(0, g['num_nodes'], (2, und.shape[1]), generator=gen)
        d = Bag(x=g['x'], num_nodes=g['num_nodes'], train_pos_edge_index=und, edge_index=und,
                dtrain_mask=torch.ones(und.shape[1], dtype=torch.bool), dr_mask=torch.ones(und.shape[1], dtype=torch.bool),
                val_pos_edge_index=g['val_pos'], val_neg_edge_index=g['val_neg'],
                test_pos_edge_index=g['test_pos'], test_neg_edge_index=g['test_neg'])
        args = make_args(A, ['--gnn', gnn, '--unlearning_model', 'original', '--in_dim', '10', '--hidden_dim', '32',
                             '--out_dim', '16', '--dataset', 'Cora', '--checkpoint_dir', tempfile.mkdtemp(), '--lr', '0.01'])
        args.epochs, args.valid_freq = 6, 1            # parse_args forces 2000 / 500 for `original`
        torch.manual_seed(12)
        model = cls(args)
        init = state_np(model)
        opt = torch.optim.Adam(model.parameters(), lr=args.lr)
        STATE['neg'], STATE['wandb'] = neg, []
        torch.manual_seed(79)
        B.Trainer(args).train_fullbatch(model, d, opt, args)
        steps = [w for w in STATE['wandb'] if 'train_loss' in w]
        vals = [w for w in STATE['wandb'] if 'val_loss' in w]
        out = dict(init)
        out.update(data_np(d, neg))
        out.update({f'final::{k}': np_(v) for k, v in model.state_dict().items()})
        out.update(train_loss=np.array([s['train_loss'] for s in steps]),
                   val_loss=np.array([v['val_loss'] for v in vals]), val_dt_auc=np.array([v['val_dt_auc'] for v in vals]),
                   lr=np.float64(args.lr), epochs=np.int64(6), eval_seed=np.int64(79))
        np.savez_compressed(os.path.join(HERE, f'orig_{gnn}.npz'), **out)


def golden_nodecls_trajectory(D, T, A):
    """The real node-unlearning loop GNNDeleteNodeClassificationTrainer.train (gnndelete_nodeemb.py:498-657):
    layer-wise DEC + NI losses over data.edge_index, two Adams, accuracy / micro-F1 evaluation
    (NodeClassificationTrainer.eval, base.py:754-790).  GATDelete: upstream's GCNDelete crashes here (F5)."""
    g = synth_graph(100, 380, 10, seed=33)
    n = g['num_nodes']
    gen = torch.Generator().manual_seed(9)
    E = g['train']
    und, _ = pyg.to_undirected(E, [torch.ones(E.shape[1], dtype=torch.int32)], n)
    # node deletion as delete_node.py:66-110 sets it up: Df = every edge touching the deleted nodes
    df_nodes = torch.randperm(n, generator=gen)[:6]
    gone = torch.zeros(n, dtype=torch.bool)
    gone[df_nodes] = True
    df_mask = gone[und[0]] | gone[und[1]]
    df_edge = und[:, df_mask]
    seeds = df_edge.flatten().unique()
    _, e2, m2 = pyg.k_hop_subgraph(seeds, 2, und, n)
    _, e1, _ = pyg.k_hop_subgraph(seeds, 1, und, n)
    s1 = torch.zeros(n, dtype=torch.bool)
    s2 = torch.zeros(n, dtype=torch.bool)
    s1[e1.flatten().unique()] = True
    s2[e2.flatten().unique()] = True
    y = torch.randint(0, 4, (n,), generator=gen)
    perm = torch.randperm(n, generator=gen)
    tr, va, te = (torch.zeros(n, dtype=torch.bool) for _ in range(3))
    tr[perm[:60]] = True
    va[perm[60:80]] = True
    te[perm[80:]] = True
    neg = torch.randint(0, n, (2, int(df_mask.sum())), generator=gen)
    d = Bag(x=g['x'], num_nodes=n, edge_index=und, y=y, train_mask=tr, val_mask=va, test_mask=te,
            df_mask=df_mask, dr_mask=~df_mask, dtrain_mask=~df_mask, sdf_mask=m2, sdf_node_1hop_mask=s1,
            sdf_node_2hop_mask=s2, directed_df_edge_index=df_edge[:, df_edge[0] < df_edge[1]])
    args = make_args(A, ['--gnn', 'gat', '--unlearning_model', 'gnndelete_nodeemb', '--in_dim', '10', '--hidden_dim', '32',
                         '--out_dim', '4', '--epochs', '6', '--valid_freq', '3', '--dataset', 'DBLP', '--lr', '0.01',
                         '--alpha', '0.4', '--checkpoint_dir', tempfile.mkdtemp()])
    torch.manual_seed(6)
    model = D.GATDelete(args, s1, s2)
    with torch.no_grad():
        for n_, p in model.named_parameters():
            if n_.endswith('bias'):
                p.copy_(torch.randn_like(p) * 0.1)
    init = state_np(model)
    model.to = lambda *a, **k: model
    opt = [torch.optim.Adam(model.deletion1.parameters(), lr=args.lr),
           torch.optim.Adam(model.deletion2.parameters(), lr=args.lr)]
    STATE['neg'], STATE['wandb'] = neg, []
    T.GNNDeleteNodeClassificationTrainer(args).train(model, d, opt, args)
    steps = [w for w in STATE['wandb'] if 'Epoch' in w]
    vals = [w for w in STATE['wandb'] if 'val_loss' in w]
    out = dict(init)
    out.update(data_np(d, neg))
    out.update(train_loss=np.array([s['train_loss'] for s in steps]), loss_r=np.array([s['loss_r'] for s in steps]),
               loss_l=np.array([s['loss_l'] for s in steps]),
               final_w1=np_(model.deletion1.deletion_weight), final_w2=np_(model.deletion2.deletion_weight),
               val_loss=np.array([v['val_loss'] for v in vals]), val_dt_acc=np.array([v['val_dt_acc'] for v in vals]),
               val_dt_f1=np.array([v['val_dt_f1'] for v in vals]),
               lr=np.float64(args.lr), alpha=np.float64(args.alpha), epochs=np.int64(6))
    np.savez_compressed(os.path.join(HERE, 'traj_nodecls_gat.npz'), **out)


def golden_rgat(D, A):
    """The reference's own RGAT / RGATConv / RGATDelete (framework/models/rgat.py, deletion.py:165-193) under a
    minimal MessagePassing: dense relation weights (3 relation types) and the block-diagonal branch the
    reference takes above 20 types (rgat.py:361-363)."""
    rgat_mod = importlib.import_module('framework.models.rgat')
    for tag, R, hidden, out_dim in [('dense', 3, 32, 16), ('blocks', 21, 32, 16)]:
        g = synth_graph(60, 220, 12, seed=14 + R, relations=R)
        E = g['train']
        ei = torch.cat([E, E.flip(0)], 1)
        et = torch.cat([g['train_type'], g['train_type'] + R])
        args = make_args(A, ['--gnn', 'rgat', '--dataset', 'WordNet18', '--in_dim', '12', '--hidden_dim', str(hidden),
                             '--out_dim', str(out_dim)])
        m1 = torch.rand(60, generator=torch.Generator().manual_seed(3)) < 0.4
        m2 = torch.rand(60, generator=torch.Generator().manual_seed(4)) < 0.7
        torch.manual_seed(8)
        model = D.RGATDelete(args, 60, R, m1, m2)
        with torch.no_grad():
            for n_, p in model.named_parameters():
                if n_.endswith('bias'):
                    p.copy_(torch.randn_like(p) * 0.1)
                if 'deletion_weight' in n_:
                    p.copy_(torch.eye(p.shape[0]) * 0.5 + torch.randn_like(p) * 0.05)
                if n_.endswith('.l2'):
                    p.zero_()                       # upstream never initialises l2 (rgat.py:165): unused, keep it finite
        x = torch.arange(60)
        z1, z2 = model(x, ei, et, return_all_emb=True)
        o1, o2 = model.get_original_embeddings(x, ei, et, return_all_emb=True)
        score = model.decode(z2, E, g['train_type'])
        loss = (z2 ** 2).mean() + (z1 ** 2).mean()
        loss.backward()
        out = dict(state_np(model))
        out.update(edge_index=np_(ei), edge_type=np_(et), x=np_(x), dec_edge=np_(E), dec_type=np_(g['train_type']),
                   score=np_(score), num_edge_type=np.int64(R), mask1=np_(m1), mask2=np_(m2), z1=np_(z1), z2=np_(z2),
                   o1=np_(o1), o2=np_(o2), gw1=np_(model.deletion1.deletion_weight.grad),
                   gw2=np_(model.deletion2.deletion_weight.grad))
        np.savez_compressed(os.path.join(HERE, f'wiring_rgat_{tag}.npz'), **out)



def _node_sets(n, k, size, seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.randperm(n, generator=g)[:size].sort().values for _ in range(k)]


def golden_minibatch(D, T, A):
    """The real GraphSAINT mini-batch loop GNNDeleteNodeembTrainer.train_minibatch
    (gnndelete_nodeemb.py:352-495) on injected batches: original embeddings on ALL batch edges, Del forward
    on the batch's S_Df edges with per-batch masks, fresh negatives per batch (recorded), layer-wise update with
    zero_grad right after each step.  GATDelete: upstream's GCNDelete crashes in this loop too."""
    g = synth_graph(140, 620, 10, seed=61)
    d, _ = prepare_deletion(g, 24, seed=9)
    model, _ = build_ref_model(D, A, 'gat', d, 10, seed=7)
    with torch.no_grad():
        model.deletion1.deletion_weight.fill_(1 / 1000)
        model.deletion2.deletion_weight.fill_(1 / 1000)
    init = state_np(model)
    model.to = lambda *a, **k: model
    args = make_args(A, ['--gnn', 'gat', '--unlearning_model', 'gnndelete_nodeemb', '--epochs', '3', '--valid_freq', '3',
                         '--checkpoint_dir', tempfile.mkdtemp(), '--dataset', 'Cora', '--lr', '0.01', '--alpha', '0.4',
                         '--batch_size', '40', '--num_steps', '3'])
    opt = [torch.optim.Adam(model.deletion1.parameters(), lr=args.lr),
           torch.optim.Adam(model.deletion2.parameters(), lr=args.lr)]
    STATE['batches'] = _node_sets(140, 3, 90, seed=3)
    STATE['neg_gen'], STATE['neg_log'], STATE['wandb'] = torch.Generator().manual_seed(17), [], []
    torch.manual_seed(80)
    try:
        T.GNNDeleteNodeembTrainer(args).train_minibatch(model, d, opt, args)
    finally:
        STATE['neg_gen'] = None
    steps = [w for w in STATE['wandb'] if 'Epoch' in w]
    vals = [w for w in STATE['wandb'] if 'val_loss' in w]
    out = dict(init)
    d2 = Bag({k: v for k, v in d.items() if k not in ('node_id',) and not k.endswith('_non_df_mask')})
    out.update(data_np(d2, torch.zeros(2, 0, dtype=torch.long)))
    for i, b in enumerate(STATE['batches']):
        out[f'batch::{i}'] = np_(b)
    for i, ng in enumerate(STATE['neg_log']):
        out[f'negs::{i}'] = np_(ng)
    out.update(n_batches=np.int64(len(STATE['batches'])), n_negs=np.int64(len(STATE['neg_log'])),
               train_loss=np.array([s_['train_loss'] for s_ in steps]),
               train_loss_l=np.array([s_['train_loss_l'] for s_ in steps]),
               train_loss_r=np.array([s_['train_loss_r'] for s_ in steps]),
               final_w1=np_(model.deletion1.deletion_weight), final_w2=np_(model.deletion2.deletion_weight),
               val_dt_auc=np.array([v['val_dt_auc'] for v in vals]), val_df_auc=np.array([v['val_df_auc'] for v in vals]),
               lr=np.float64(args.lr), alpha=np.float64(args.alpha), epochs=np.int64(3), eval_seed=np.int64(80))
    np.savez_compressed(os.path.join(HERE, 'traj_minibatch_gat.npz'), **out)


def golden_edgeprob_minibatch(D, TE, A):
    """The edge-probability MINI-BATCH loop GNNDeleteTrainer.train_minibatch (framework/trainer/gnndelete.py:312-450) on
    injected GraphSAINT batches.  Upstream this loop is dead code: Trainer.get_embedding (base.py:52-63) reads
    data.dtrain_mask, which delete_gnn.py never sets (:124, :190 are commented out) -> AttributeError.  The fixture injects
    dtrain_mask = dr_mask (what the commented lines assigned) and runs the real loop: z_ori = the model's embedding of the
    whole graph on the Dr edges (computed once), per batch the Del forward on the batch's S_Df edges, fresh negatives
    (recorded), loss_e = MSE(df logits, negative logits), loss_l = MSE of the dot products of the batch's S_Df edges
    (row < col) against z_ori INDEXED WITH THE BATCH-LOCAL ids (an upstream quirk, kept), 0.5 / 0.5, single Adam.
    `logits_ori.to('cuda')` inside the loop (:383) is neutralised for this CPU run by a Tensor.to shim (environment,
    not reference code)."""
    _to = torch.Tensor.to

    def to_(self, *a, **k):
        if a and isinstance(a[0], str) and a[0] == 'cuda':
            return self
        return _to(self, *a, **k)
    upstream_error = None
    for gnn in ['gcn', 'gat']:
        g = synth_graph(140, 620, 10, seed=61)
        d, _ = prepare_deletion(g, 24, seed=9)
        if upstream_error is None:               # as delete_gnn.py hands the data over: no dtrain_mask -> the loop cannot start
            try:
                m0, _ = build_ref_model(D, A, gnn, d, 10, seed=7)
                a0 = make_args(A, ['--gnn', gnn, '--unlearning_model', 'gnndelete', '--epochs', '1', '--valid_freq', '1',
                                   '--checkpoint_dir', tempfile.mkdtemp(), '--dataset', 'Cora'])
                TE.GNNDeleteTrainer(a0).train_minibatch(m0, d, None, a0)
                upstream_error = 'no error'
            except Exception as e:               # noqa: BLE001
                upstream_error = f'{type(e).__name__}: {e}'
        d['dtrain_mask'] = d['dr_mask']
        model, _ = build_ref_model(D, A, gnn, d, 10, seed=7)
        with torch.no_grad():
            model.deletion1.deletion_weight.fill_(1 / 1000)
            model.deletion2.deletion_weight.fill_(1 / 1000)
        init = state_np(model)
        model.to = lambda *a, **k: model
        args = make_args(A, ['--gnn', gnn, '--unlearning_model', 'gnndelete', '--epochs', '3', '--valid_freq', '3',
                             '--checkpoint_dir', tempfile.mkdtemp(), '--dataset', 'Cora', '--lr', '0.01',
                             '--batch_size', '40', '--num_steps', '3'])
        opt = torch.optim.Adam([{'params': [p for n_, p in model.named_parameters() if 'del' in n_], 'weight_decay': 0.0}], lr=args.lr)
        STATE['batches'] = _node_sets(140, 3, 90, seed=3)
        STATE['neg_gen'], STATE['neg_log'], STATE['wandb'] = torch.Generator().manual_seed(17), [], []
        torch.manual_seed(81)
        torch.Tensor.to = to_
        try:
            TE.GNNDeleteTrainer(args).train_minibatch(model, d, opt, args)
        finally:
            STATE['neg_gen'] = None
            torch.Tensor.to = _to
        logs = [w for w in STATE['wandb'] if 'train_loss' in w]
        vals = [w for w in STATE['wandb'] if 'val_loss' in w]
        out = dict(init)
        d2 = Bag({k: v for k, v in d.items() if k not in ('node_id', 'edge_index')})
        out.update(data_np(d2, torch.zeros(2, 0, dtype=torch.long)))
        for i, b in enumerate(STATE['batches']):
            out[f'batch::{i}'] = np_(b)
        for i, ng in enumerate(STATE['neg_log']):
            out[f'negs::{i}'] = np_(ng)
        out.update(n_batches=np.int64(len(STATE['batches'])), n_negs=np.int64(len(STATE['neg_log'])),
                   log_train_loss=np.array([l_['train_loss'] for l_ in logs]),
                   log_train_loss_l=np.array([l_['train_loss_l'] for l_ in logs]),
                   log_train_loss_e=np.array([l_['train_loss_e'] for l_ in logs]),
                   final_w1=np_(model.deletion1.deletion_weight), final_w2=np_(model.deletion2.deletion_weight),
                   val_dt_auc=np.array([v['val_dt_auc'] for v in vals]), val_df_auc=np.array([v['val_df_auc'] for v in vals]),
                   lr=np.float64(args.lr), epochs=np.int64(3), eval_seed=np.int64(81))
        np.savez_compressed(os.path.join(HERE, f'traj_edgeprob_minibatch_{gnn}.npz'), **out)
    STATE['edgeprob_minibatch_upstream_error'] = upstream_error


def kg_request(n, m, R, seed, df_count):
    """delete_gnn.py:85-171 (KG branch) on a synthetic relational graph: directed training triples, Df by index,
    k-hop masks on the directed list, reverse edges with type + R appended, masks repeat(2)."""
    g = synth_graph(n, m, 4, seed=seed, relations=R)
    gen = torch.Generator().manual_seed(seed + 1)
    E, et = g['train'], g['train_type']
    df_idx = torch.randperm(E.shape[1], generator=gen)[:df_count]
    df_mask = torch.zeros(E.shape[1], dtype=torch.bool)
    df_mask[df_idx] = True
    seeds = E[:, df_mask].flatten().unique()
    _, e2, m2 = pyg.k_hop_subgraph(seeds, 2, E, n)
    _, e1, _ = pyg.k_hop_subgraph(seeds, 1, E, n)
    s1 = torch.zeros(n, dtype=torch.bool)
    s2 = torch.zeros(n, dtype=torch.bool)
    s1[e1.flatten().unique()] = True
    s2[e2.flatten().unique()] = True
    nv = g['val_pos'].shape[1]
    d = Bag(x=torch.arange(n), num_nodes=n, train_pos_edge_index=E, train_edge_type=et,
            edge_index=torch.cat([E, E.flip(0)], 1), edge_type=torch.cat([et, et + R]),
            df_mask=df_mask.repeat(2), dr_mask=(~df_mask).repeat(2), sdf_mask=m2.repeat(2),
            sdf_node_1hop_mask=s1, sdf_node_2hop_mask=s2,
            directed_df_edge_index=E[:, df_mask], directed_df_edge_type=et[df_mask],
            val_pos_edge_index=g['val_pos'], val_neg_edge_index=g['val_neg'],
            val_edge_type=torch.randint(0, R, (nv,), generator=gen),
            test_pos_edge_index=g['test_pos'], test_neg_edge_index=g['test_neg'],
            test_edge_type=torch.randint(0, R, (g['test_pos'].shape[1],), generator=gen))
    return d


def golden_kg(D, T, A, B):
    """KGGNNDeleteNodeembTrainer.train (gnndelete_nodeemb.py:659-846) on injected batches with RGCNDelete at
    R = 21 relation types (> 20, so RGCNConv takes its block-diagonal branch, rgcn.py:17-22), negatives from the
    reference's own negative_sampling_kg under a recorded seed; and KGTrainer.eval (base.py:495-567: DistMult
    scores without sigmoid for Dt, 500 fresh Dr subsets for Df) on the model it leaves behind."""
    R, n = 21, 150
    d = kg_request(n, 900, R, seed=71, df_count=40)
    args = make_args(A, ['--gnn', 'rgcn', '--unlearning_model', 'gnndelete_nodeemb', '--dataset', 'WordNet18',
                         '--in_dim', '32', '--hidden_dim', '32', '--out_dim', '16', '--checkpoint_dir', tempfile.mkdtemp(),
                         '--alpha', '0.4'])
    args.epochs, args.valid_freq, args.batch_size, args.num_steps, args.lr, args.num_edge_type = 2, 2, 30, 3, 0.01, R
    torch.manual_seed(15)
    model = D.RGCNDelete(args, n, R, d['sdf_node_1hop_mask'], d['sdf_node_2hop_mask'])
    with torch.no_grad():
        for n_, p in model.named_parameters():
            if n_.endswith('bias'):
                p.copy_(torch.randn_like(p) * 0.1)
    init = state_np(model)
    model.to = lambda *a, **k: model
    opt = [torch.optim.Adam(model.deletion1.parameters(), lr=args.lr),
           torch.optim.Adam(model.deletion2.parameters(), lr=args.lr)]
    STATE['batches'] = _node_sets(n, 3, 110, seed=5)
    STATE['wandb'] = []
    torch.manual_seed(81)                     # negative_sampling_kg and the 500 Dr subsets of eval draw from this stream
    trainer = T.KGGNNDeleteNodeembTrainer(args)
    trainer.train(model, d, opt, args)
    steps = [w for w in STATE['wandb'] if 'Epoch' in w]
    vals = [w for w in STATE['wandb'] if 'val_loss' in w]
    out = dict(init)
    d2 = Bag({k: v for k, v in d.items() if not k.endswith('_non_df_mask')})
    out.update(data_np(d2, torch.zeros(2, 0, dtype=torch.long)))
    for i, b in enumerate(STATE['batches']):
        out[f'batch::{i}'] = np_(b)
    out.update(n_batches=np.int64(3), num_edge_type=np.int64(R),
               train_loss=np.array([s_['train_loss'] for s_ in steps]), loss_r=np.array([s_['loss_r'] for s_ in steps]),
               loss_l=np.array([s_['loss_l'] for s_ in steps]),
               final_w1=np_(model.deletion1.deletion_weight), final_w2=np_(model.deletion2.deletion_weight),
               val_loss=np.array([v['val_loss'] for v in vals]), val_dt_auc=np.array([v['val_dt_auc'] for v in vals]),
               val_dt_aup=np.array([v['val_dt_aup'] for v in vals]), val_df_auc=np.array([v['val_df_auc'] for v in vals]),
               val_df_aup=np.array([v['val_df_aup'] for v in vals]),
               lr=np.float64(args.lr), alpha=np.float64(args.alpha), epochs=np.int64(2), seed=np.int64(81))
    np.savez_compressed(os.path.join(HERE, 'traj_kg_rgcn.npz'), **out)
    # ---- KGTrainer.eval on the trained model, own seed
    torch.manual_seed(82)
    loss, dt_auc, dt_aup, df_auc, df_aup, df_logit, _, log = B.KGTrainer.eval(trainer, model, d, 'test')
    ev = dict(state_np(model))
    ev.update(data_np(d2, torch.zeros(2, 0, dtype=torch.long)))
    ev.update(num_edge_type=np.int64(R), eval_seed=np.int64(82), test_loss=np.float64(loss), test_dt_auc=np.float64(dt_auc),
              test_dt_aup=np.float64(dt_aup), test_df_auc=np.float64(df_auc), test_df_aup=np.float64(df_aup),
              test_df_logit=np.array(df_logit))
    np.savez_compressed(os.path.join(HERE, 'eval_kg.npz'), **ev)


def golden_retrain(A):
    """RetrainTrainer.train_fullbatch (framework/trainer/retrain.py:39-131): BCE link prediction on Dr only with
    per-epoch negatives (recorded), Adam on every parameter, Trainer.eval for model selection; plus
    verification_error (framework/evaluation.py:63-81) between the retrained and a second model."""
    RT = importlib.import_module('framework.trainer.retrain')
    EV = importlib.import_module('framework.evaluation')
    GCN = importlib.import_module('framework.models.gcn').GCN
    g = synth_graph(90, 340, 10, seed=43)
    d, _ = prepare_deletion(g, 12, seed=10)
    args = make_args(A, ['--gnn', 'gcn', '--unlearning_model', 'retrain', '--in_dim', '10', '--hidden_dim', '32',
                         '--out_dim', '16', '--dataset', 'Cora', '--checkpoint_dir', tempfile.mkdtemp(), '--lr', '0.01'])
    args.epochs, args.valid_freq = 5, 5
    torch.manual_seed(13)
    model = GCN(args)
    other = GCN(args)
    init = state_np(model)
    model.to = lambda *a, **k: model
    opt = torch.optim.Adam(model.parameters(), lr=args.lr)
    STATE['neg_gen'], STATE['neg_log'], STATE['wandb'] = torch.Generator().manual_seed(19), [], []
    torch.manual_seed(83)
    try:
        RT.RetrainTrainer(args).train_fullbatch(model, d, opt, args)
    finally:
        STATE['neg_gen'] = None
    steps = [w for w in STATE['wandb'] if 'Epoch' in w]
    vals = [w for w in STATE['wandb'] if 'val_loss' in w]
    out = dict(init)
    out.update(data_np(d, torch.zeros(2, 0, dtype=torch.long)))
    out.update({f'final::{k}': np_(v) for k, v in model.state_dict().items()})
    out.update({f'other::{k}': np_(v) for k, v in other.state_dict().items()})
    for i, ng in enumerate(STATE['neg_log']):
        out[f'negs::{i}'] = np_(ng)
    out.update(n_negs=np.int64(len(STATE['neg_log'])), train_loss=np.array([s_['train_loss'] for s_ in steps]),
               val_loss=np.array([v['val_loss'] for v in vals]), val_dt_auc=np.array([v['val_dt_auc'] for v in vals]),
               val_df_auc=np.array([v['val_df_auc'] for v in vals]),
               ve=np.float64(EV.verification_error(model, other)),
               lr=np.float64(args.lr), epochs=np.int64(5), eval_seed=np.int64(83))
    np.savez_compressed(os.path.join(HERE, 'retrain_gcn.npz'), **out)


def golden_retrain_kg(A):
    """KGRetrainTrainer.train (framework/trainer/retrain.py:235-339) on the reference's own RGCN at 21 relation types
    (block-diagonal branch), injected GraphSAINT batches: message passing and positives on the batch's Dr edges only,
    forward-direction types decoded against head-shuffled negatives (global torch RNG, recorded seed), gradient norm
    clipped to 1, KGTrainer.eval for model selection."""
    RT = importlib.import_module('framework.trainer.retrain')
    RGCN = importlib.import_module('framework.models.rgcn').RGCN
    R_, n = 21, 150
    d = kg_request(n, 900, R_, seed=75, df_count=60)
    args = make_args(A, ['--gnn', 'rgcn', '--unlearning_model', 'retrain', '--dataset', 'WordNet18', '--in_dim', '32',
                         '--hidden_dim', '32', '--out_dim', '16', '--checkpoint_dir', tempfile.mkdtemp()])
    args.epochs, args.valid_freq, args.num_steps, args.lr, args.num_edge_type = 2, 2, 3, 0.01, R_
    torch.manual_seed(24)
    model = RGCN(args, n, R_)
    init = state_np(model)
    model.to = lambda *a, **k: model
    opt = torch.optim.Adam(model.parameters(), lr=args.lr)
    STATE['batches'] = _node_sets(n, 3, 110, seed=8)
    STATE['wandb'] = []
    torch.manual_seed(87)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')                  # (upstream ends with np.mean of an empty list)
        RT.KGRetrainTrainer(args).train(model, d, opt, args)
    steps = [w for w in STATE['wandb'] if 'step' in w]
    vals = [w for w in STATE['wandb'] if 'val_loss' in w]
    out = dict(init)
    d2 = Bag({k: v for k, v in d.items()})
    out.update(data_np(d2, torch.zeros(2, 0, dtype=torch.long)))
    out.update({f'final::{k}': np_(v) for k, v in model.state_dict().items()})
    for i, b in enumerate(STATE['batches']):
        out[f'batch::{i}'] = np_(b)
    out.update(n_batches=np.int64(3), num_edge_type=np.int64(R_), train_loss=np.array([s_['train_loss'] for s_ in steps]),
               val_loss=np.array([v['val_loss'] for v in vals]), val_dt_auc=np.array([v['val_dt_auc'] for v in vals]),
               val_dt_aup=np.array([v['val_dt_aup'] for v in vals]), val_df_auc=np.array([v['val_df_auc'] for v in vals]),
               lr=np.float64(args.lr), epochs=np.int64(2), seed=np.int64(87))
    np.savez_compressed(os.path.join(HERE, 'retrain_kg_rgcn.npz'), **out)


def golden_split():
    """train_test_split_edges_no_neg_adj_mask (prepare_dataset.py:31-136) and the IN / OUT candidate masks
    (:205-214) run from the reference's own module: plain, with the two-hop-degree ordering the ogbl-* datasets
    get (:186-189), and the KG branch (edge types sliced WITHOUT the permutation, as upstream does)."""
    old_cwd = os.getcwd()
    os.chdir(tempfile.mkdtemp())
    sys.path.insert(0, REF)
    try:
        P = importlib.import_module('prepare_dataset')
    finally:
        sys.path.remove(REF)
        os.chdir(old_cwd)
    out = {}
    for tag, n, m, kg, deg in [('plain', 80, 300, False, False), ('degree', 90, 360, False, True), ('kg', 70, 260, True, False)]:
        g = synth_graph(n, m, 3, seed=91 + n, relations=5 if kg else 0)
        gen = torch.Generator().manual_seed(n)
        E = torch.cat([g['train'], g['val_pos'], g['test_pos']], 1)
        if kg:
            ei = E[:, torch.randperm(E.shape[1], generator=gen)]
            et = torch.randint(0, 5, (ei.shape[1],), generator=gen)
        else:
            both = torch.cat([E, E.flip(0)], 1)
            ei = both[:, torch.randperm(both.shape[1], generator=gen)]
            et = None
        n_dir = int((ei[0] < ei[1]).sum()) if not kg else ei.shape[1]
        thd = torch.randint(10, 100, (n_dir,), generator=gen) if deg else None
        data = Bag(num_nodes=n, edge_index=ei.clone(), edge_attr=None, edge_type=et)
        STATE['neg_gen'], STATE['neg_log'] = torch.Generator().manual_seed(23), []
        torch.manual_seed(500 + n)
        try:
            res = P.train_test_split_edges_no_neg_adj_mask(data, test_ratio=0.05, two_hop_degree=thd, kg=kg)
        finally:
            STATE['neg_gen'] = None
        _, _, _, local = sys.modules['torch_geometric.utils'].k_hop_subgraph(
            res['test_pos_edge_index'].flatten().unique(), 2, res['train_pos_edge_index'], num_nodes=n)
        out.update({f'{tag}::edge_index': np_(ei), f'{tag}::num_nodes': np.int64(n), f'{tag}::seed': np.int64(500 + n),
                    f'{tag}::train': np_(res['train_pos_edge_index']), f'{tag}::val': np_(res['val_pos_edge_index']),
                    f'{tag}::test': np_(res['test_pos_edge_index']), f'{tag}::in_mask': np_(local)})
        if thd is not None:
            out[f'{tag}::two_hop_degree'] = np_(thd)
        if kg:
            out.update({f'{tag}::edge_type': np_(et), f'{tag}::train_type': np_(res['train_edge_type']),
                        f'{tag}::val_type': np_(res['val_edge_type']), f'{tag}::test_type': np_(res['test_edge_type']),
                        f'{tag}::val_neg': np_(res['val_neg_edge_index']), f'{tag}::test_neg': np_(res['test_neg_edge_index'])})
    np.savez_compressed(os.path.join(HERE, 'split.npz'), **out)


class FakeOgbKG:
    """Stands in for ogb.linkproppred.PygLinkPropPredDataset on a knowledge graph with typed entities: what
    process_kg (prepare_dataset.py:300-353) reads from it."""

    def __init__(self, root=None, name=None):
        self.split, self.num_nodes_dict = STATE['ogb_split'], STATE['ogb_num_nodes']

    def get_edge_split(self):
        return {k: dict(v) for k, v in self.split.items()}

    def __getitem__(self, i):
        return Bag(num_nodes_dict=self.num_nodes_dict, num_nodes=sum(self.num_nodes_dict.values()))

    def __repr__(self):
        return 'FakeOgbKG()'


def golden_process_kg():
    """process_kg's ogbl branch (prepare_dataset.py:300-399) run from the reference's own module on a small split in
    OGB's in-memory layout (per-type local entity ids, head_type / tail_type strings, 500 -> 4 corrupted tails):
    global ids, same-type relations kept once (head < tail), inverse triples with relation + 51, first corrupted
    tail as the negative, IN / OUT candidate masks from the 2-hop enclosing subgraph of the test triples."""
    old_cwd = os.getcwd()
    tmp = tempfile.mkdtemp()
    os.chdir(tmp)
    sys.path.insert(0, REF)
    try:
        P = importlib.import_module('prepare_dataset')
    finally:
        sys.path.remove(REF)
        os.chdir(old_cwd)
    g = torch.Generator().manual_seed(77)
    types = {'disease': 9, 'drug': 12, 'protein': 17}            # OGB's num_nodes_dict order (alphabetical)
    names = list(types)
    rel_types = [('drug', 'disease'), ('drug', 'protein'), ('protein', 'protein'), ('drug', 'drug'), ('disease', 'protein'),
                 ('protein', 'protein')]                          # relation id -> (head type, tail type)

    def triples(m, both_ways):
        rel = torch.randint(0, len(rel_types), (m,), generator=g)
        ht = [rel_types[r][0] for r in rel.tolist()]
        tt = [rel_types[r][1] for r in rel.tolist()]
        head = torch.tensor([int(torch.randint(0, types[a], (1,), generator=g)) for a in ht])
        tail = torch.tensor([int(torch.randint(0, types[b], (1,), generator=g)) for b in tt])
        if both_ways:                                            # same-type relations are stored in both directions
            same = torch.tensor([a == b for a, b in zip(ht, tt)])
            idx = same.nonzero().flatten().tolist()
            head, tail = torch.cat([head, tail[idx]]), torch.cat([tail, head[idx]])
            rel = torch.cat([rel, rel[idx]])
            ht, tt = ht + [tt[i] for i in idx], tt + [ht[i] for i in idx]
        return {'head': head.numpy(), 'relation': rel.numpy(), 'tail': tail.numpy(), 'head_type': ht, 'tail_type': tt}

    split = {'train': triples(260, True), 'valid': triples(20, False), 'test': triples(24, False)}
    for k in ('valid', 'test'):
        m = len(split[k]['head'])
        split[k]['head_neg'] = torch.randint(0, 9, (m, 4), generator=g)
        split[k]['tail_neg'] = torch.randint(0, 38, (m, 4), generator=g)
    STATE['ogb_split'], STATE['ogb_num_nodes'] = split, types
    P.PygLinkPropPredDataset = FakeOgbKG
    P.kg_datasets, P.seeds, P.data_dir = ['ogbl-fake'], [42], tmp
    os.makedirs(os.path.join(tmp, 'ogbl-fake'), exist_ok=True)
    P.process_kg()
    with open(os.path.join(tmp, 'ogbl-fake', 'd_42.pkl'), 'rb') as f:
        _, data = pickle.load(f)
    df = torch.load(os.path.join(tmp, 'ogbl-fake', 'df_42.pt'))
    out = {'types': np.array(names), 'type_count': np.array([types[t] for t in names])}
    for k, d in split.items():
        for name, v in d.items():
            out[f'in::{k}::{name}'] = np.array(v) if isinstance(v, list) else np_(v) if torch.is_tensor(v) else np.asarray(v)
    for k in ('x', 'edge_index', 'edge_type', 'train_pos_edge_index', 'train_edge_type', 'val_pos_edge_index', 'val_edge_type',
              'val_neg_edge_index', 'test_pos_edge_index', 'test_edge_type', 'test_neg_edge_index'):
        out[f'out::{k}'] = np_(data[k])
    out['out::in_mask'], out['out::out_mask'] = np_(df['in']), np_(df['out'])
    np.savez_compressed(os.path.join(HERE, 'process_kg.npz'), **out)


def golden_wide_trajectories(D, T, A):
    """train_fullbatch at the widths the fused HIP stages are built for (in 32 -> hidden 128 -> out 64), so that the
    reference-loop fixtures drive the MFMA row kernels and the fused loss / Del stages, not the generic fallbacks."""
    # one thread: at these widths the host BLAS splits its reductions by thread count and timing, and the fixture
    # would differ in the last bit from run to run
    threads = torch.get_num_threads()
    torch.set_num_threads(1)
    for gnn, loss_type in [('gcn', 'both_all'), ('gat', 'both_layerwise')]:
        g = synth_graph(160, 700, 32, seed=25)
        d, neg = prepare_deletion(g, 20, seed=11)
        model, _ = build_ref_model(D, A, gnn, d, 32, seed=14, hidden=128, out=64)
        with torch.no_grad():
            model.deletion1.deletion_weight.fill_(1 / 1000)
            model.deletion2.deletion_weight.fill_(1 / 1000)
        init = state_np(model)
        model.to = lambda *a, **k: model
        args = make_args(A, ['--gnn', gnn, '--unlearning_model', 'gnndelete_nodeemb', '--epochs', '5', '--valid_freq', '5',
                             '--loss_type', loss_type, '--checkpoint_dir', tempfile.mkdtemp(), '--dataset', 'Cora',
                             '--lr', '0.01', '--alpha', '0.4'])
        if 'layerwise' in loss_type:
            opt = [torch.optim.Adam(model.deletion1.parameters(), lr=args.lr),
                   torch.optim.Adam(model.deletion2.parameters(), lr=args.lr)]
        else:
            opt = torch.optim.Adam([{'params': [p for n_, p in model.named_parameters() if 'del' in n_],
                                     'weight_decay': 0.0}], lr=args.lr)
        STATE['neg'], STATE['wandb'] = neg, []
        torch.manual_seed(84)
        T.GNNDeleteNodeembTrainer(args).train_fullbatch(model, d, opt, args)
        steps = [w for w in STATE['wandb'] if 'Epoch' in w]
        vals = [w for w in STATE['wandb'] if 'val_loss' in w]
        out = dict(init)
        out.update(data_np(d, neg))
        out.update(train_loss=np.array([s_['train_loss'] for s_ in steps]), loss_r=np.array([s_['loss_r'] for s_ in steps]),
                   loss_l=np.array([s_['loss_l'] for s_ in steps]),
                   final_w1=np_(model.deletion1.deletion_weight), final_w2=np_(model.deletion2.deletion_weight),
                   val_dt_auc=np.array([v['val_dt_auc'] for v in vals]), val_df_auc=np.array([v['val_df_auc'] for v in vals]),
                   val_loss=np.array([v['val_loss'] for v in vals]),
                   val_df_logit_mean=np.array([v['val_df_logit_mean'] for v in vals]),
                   lr=np.float64(args.lr), alpha=np.float64(args.alpha), epochs=np.int64(5), eval_seed=np.int64(84))
        np.savez_compressed(os.path.join(HERE, f'traj_wide_{gnn}_{loss_type}.npz'), **out)
    torch.set_num_threads(threads)


def golden_original_minibatch(B, A):
    """The original-model MINI-BATCH loops on injected GraphSAINT batches: Trainer.train_minibatch
    (framework/trainer/base.py:144-227; the reference's own GCN, negatives = one per batch edge, recorded) and
    KGTrainer.train (base.py:394-493; the reference's own RGCN at 21 relation types, DistMult decoder on the
    forward-direction batch edges, negative_sampling_kg from a recorded seed, model selection on validation AUP)."""
    GCN = importlib.import_module('framework.models.gcn').GCN
    g = synth_graph(140, 620, 10, seed=63)
    E = g['train']
    und, _ = pyg.to_undirected(E, [torch.ones(E.shape[1], dtype=torch.int32)], g['num_nodes'])
    d = Bag(x=g['x'], num_nodes=g['num_nodes'], train_pos_edge_index=und, edge_index=und,
            dtrain_mask=torch.ones(und.shape[1], dtype=torch.bool), dr_mask=torch.ones(und.shape[1], dtype=torch.bool),
            val_pos_edge_index=g['val_pos'], val_neg_edge_index=g['val_neg'],
            test_pos_edge_index=g['test_pos'], test_neg_edge_index=g['test_neg'])
    args = make_args(A, ['--gnn', 'gcn', '--unlearning_model', 'original', '--in_dim', '10', '--hidden_dim', '32',
                         '--out_dim', '16', '--dataset', 'Cora', '--checkpoint_dir', tempfile.mkdtemp(), '--lr', '0.01'])
    args.epochs, args.valid_freq, args.batch_size, args.num_steps = 2, 2, 40, 3
    torch.manual_seed(21)
    model = GCN(args)
    init = state_np(model)
    opt = torch.optim.Adam(model.parameters(), lr=args.lr)
    STATE['batches'] = _node_sets(140, 3, 90, seed=4)
    STATE['neg_gen'], STATE['neg_log'], STATE['wandb'] = torch.Generator().manual_seed(29), [], []
    torch.manual_seed(85)
    try:
        B.Trainer(args).train_minibatch(model, d, opt, args)
    finally:
        STATE['neg_gen'] = None
    steps = [w for w in STATE['wandb'] if 'step' in w]
    vals = [w for w in STATE['wandb'] if 'val_loss' in w]
    out = dict(init)
    out.update(data_np(d, torch.zeros(2, 0, dtype=torch.long)))
    out.update({f'final::{k}': np_(v) for k, v in model.state_dict().items()})
    for i, b in enumerate(STATE['batches']):
        out[f'batch::{i}'] = np_(b)
    for i, ng in enumerate(STATE['neg_log']):
        out[f'negs::{i}'] = np_(ng)
    out.update(n_batches=np.int64(3), n_negs=np.int64(len(STATE['neg_log'])), train_loss=np.array([s_['train_loss'] for s_ in steps]),
               val_loss=np.array([v['val_loss'] for v in vals]), val_dt_auc=np.array([v['val_dt_auc'] for v in vals]),
               lr=np.float64(args.lr), epochs=np.int64(2), eval_seed=np.int64(85))
    np.savez_compressed(os.path.join(HERE, 'orig_minibatch_gcn.npz'), **out)

    # ---- KGTrainer.train
    RGCN = importlib.import_module('framework.models.rgcn').RGCN
    R_, n = 21, 150
    d = kg_request(n, 900, R_, seed=73, df_count=40)
    d['dr_mask'] = torch.ones_like(d['dr_mask'])                  # original training: nothing is deleted yet
    d['df_mask'] = torch.zeros_like(d['df_mask'])
    args = make_args(A, ['--gnn', 'rgcn', '--unlearning_model', 'original', '--dataset', 'WordNet18', '--in_dim', '32',
                         '--hidden_dim', '32', '--out_dim', '16', '--checkpoint_dir', tempfile.mkdtemp()])
    args.epochs, args.valid_freq, args.num_steps, args.lr, args.num_edge_type = 2, 2, 3, 0.01, R_
    torch.manual_seed(22)
    model = RGCN(args, n, R_)
    init = state_np(model)
    model.to = lambda *a, **k: model
    opt = torch.optim.Adam(model.parameters(), lr=args.lr)
    STATE['batches'] = _node_sets(n, 3, 110, seed=6)
    STATE['wandb'] = []
    torch.manual_seed(86)
    B.KGTrainer(args).train(model, d, opt, args)
    steps = [w for w in STATE['wandb'] if 'step' in w]
    vals = [w for w in STATE['wandb'] if 'val_loss' in w]
    out = dict(init)
    d2 = Bag({k: v for k, v in d.items()})
    out.update(data_np(d2, torch.zeros(2, 0, dtype=torch.long)))
    out.update({f'final::{k}': np_(v) for k, v in model.state_dict().items()})
    for i, b in enumerate(STATE['batches']):
        out[f'batch::{i}'] = np_(b)
    out.update(n_batches=np.int64(3), num_edge_type=np.int64(R_), train_loss=np.array([s_['train_loss'] for s_ in steps]),
               val_loss=np.array([v['val_loss'] for v in vals]), val_dt_auc=np.array([v['val_dt_auc'] for v in vals]),
               val_dt_aup=np.array([v['val_dt_aup'] for v in vals]), lr=np.float64(args.lr), epochs=np.int64(2), seed=np.int64(86))
    np.savez_compressed(os.path.join(HERE, 'orig_kg_rgcn.npz'), **out)


def golden_gcn_layerwise_crash(D, T, A):
    """record that upstream GCNDelete + both_layerwise raises."""
    g = synth_graph(40, 120, 6, seed=31)
    d, neg = prepare_deletion(g, 5, seed=7)
    model, _ = build_ref_model(D, A, 'gcn', d, 6, seed=4)
    model.to = lambda *a, **k: model
    tmp = tempfile.mkdtemp()
    args = make_args(A, ['--gnn', 'gcn', '--unlearning_model', 'gnndelete_nodeemb', '--epochs', '2',
                         '--checkpoint_dir', tmp, '--dataset', 'Cora'])
    opt = [torch.optim.Adam(model.deletion1.parameters(), lr=args.lr),
           torch.optim.Adam(model.deletion2.parameters(), lr=args.lr)]
    STATE['neg'] = neg
    try:
        T.GNNDeleteNodeembTrainer(args).train_fullbatch(model, d, opt, args)
        msg = 'no error'
    except RuntimeError as e:
        msg = str(e).splitlines()[0]
    return msg


def golden_parse_args(A):
    cases = {
        'default': [],
        'cora_nodeemb': ['--unlearning_model', 'gnndelete_nodeemb', '--gnn', 'gcn', '--dataset', 'Cora',
                         '--df', 'out', '--df_size', '0.5', '--epochs', '1500'],
        'dblp_nodeemb': ['--unlearning_model', 'gnndelete_nodeemb', '--gnn', 'gcn', '--dataset', 'DBLP',
                         '--df', 'out', '--df_size', '2.5', '--random_seed', '21'],
        'collab_gnndelete': ['--unlearning_model', 'gnndelete', '--gnn', 'gcn', '--dataset', 'ogbl-collab',
                             '--df', 'in', '--df_size', '5', '--epochs', '1500'],
        'biokg_rgcn': ['--unlearning_model', 'gnndelete_nodeemb', '--gnn', 'rgcn', '--dataset', 'ogbl-biokg',
                       '--df', 'in', '--df_size', '2.5'],
        'wn18_rgcn': ['--unlearning_model', 'gnndelete', '--gnn', 'rgcn', '--dataset', 'WordNet18'],
        'original_collab': ['--unlearning_model', 'original', '--gnn', 'gat', '--dataset', 'ogbl-collab'],
        'retrain_rgcn_biokg': ['--unlearning_model', 'retrain', '--gnn', 'rgcn', '--dataset', 'ogbl-biokg'],
        'grad_ascent': ['--unlearning_model', 'gradient_ascent', '--gnn', 'gin', '--dataset', 'PubMed'],
        'dtd': ['--unlearning_model', 'descent_to_delete'],
        'graph_editor': ['--unlearning_model', 'graph_editor'],
        'molhiv': ['--dataset', 'ogbg-molhiv'],
        'flags': ['--regen_feats', '--regen_links', '--eval_on_cpu', 'x', '--loss_fct', 'kld_mean',
                  '--loss_type', 'only1', '--alpha', '0.25', '--num_edge_type', '7', '--topk', '10'],
    }
    out = {k: {'argv': v, 'args': vars(make_args(A, v))} for k, v in cases.items()}
    with open(os.path.join(HERE, 'parse_args.json'), 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)


def golden_eval(D, T, A):
    """Trainer.eval / Trainer.test on a fixed model (base.py:229-375)."""
    g = synth_graph(70, 260, 9, seed=41)
    d, neg = prepare_deletion(g, 9, seed=8)
    model, _ = build_ref_model(D, A, 'gat', d, 9, seed=12)
    tmp = tempfile.mkdtemp()
    args = make_args(A, ['--gnn', 'gat', '--unlearning_model', 'gnndelete_nodeemb', '--checkpoint_dir', tmp,
                         '--dataset', 'Cora'])
    trainer = T.GNNDeleteNodeembTrainer(args)
    torch.manual_seed(123)
    loss, dt_auc, dt_aup, df_auc, df_aup, df_logit, _, log = trainer.eval(model, d, 'val')
    masks = torch.stack(trainer.df_pos_edge)
    torch.save({'model_state': model.state_dict()}, os.path.join(tmp, 'model_best.pt'))
    tl, tdt_auc, tdt_aup, tdf_auc, tdf_aup, tdf_logit, all_pair, tlog = trainer.test(model, d)
    out = dict(state_np(model))
    out.update(data_np(d, neg))
    out.update(val_loss=np.float64(loss), val_dt_auc=np.float64(dt_auc), val_dt_aup=np.float64(dt_aup),
               val_df_auc=np.float64(df_auc), val_df_aup=np.float64(df_aup), val_df_logit=np.array(df_logit),
               df_pos_masks=np_(masks), eval_seed=np.int64(123),
               test_loss=np.float64(tl), test_dt_auc=np.float64(tdt_auc), test_df_auc=np.float64(tdf_auc),
               test_auc_sum=np.float64(trainer.trainer_log['auc_sum']),
               test_auc_gap=np.float64(trainer.trainer_log['auc_gap']),
               test_all_pair=np_(all_pair))
    np.savez_compressed(os.path.join(HERE, 'eval.npz'), **out)


def golden_neg_kg(U):
    g = torch.Generator().manual_seed(3)
    ei = torch.randint(0, 50, (2, 90), generator=g)
    et = torch.randint(0, 5, (90,), generator=g)
    torch.manual_seed(2024)
    neg = U.negative_sampling_kg(ei, et)
    np.savez_compressed(os.path.join(HERE, 'neg_kg.npz'), edge_index=np_(ei), edge_type=np_(et),
                        neg=np_(neg), seed=np.int64(2024))


def golden_prep(D):
    """Run the reference's real delete_gnn.main() (delete_gnn.py:57-283) up to the
    trainer hand-off and capture the Data it built: Df selection, k-hop S_Df masks,
    symmetrisation, optimizer construction.  get_trainer returns a recorder."""
    captured = {}

    class Recorder:
        def __init__(self, args):
            self.args = args

        def train(self, model, data, optimizer, args, *rest):
            captured['data'] = data
            captured['opt'] = optimizer
            captured['args'] = args
            captured['model'] = model

        def test(self, *a, **k):
            return [{}]

        def save_log(self):
            pass

    fw = sys.modules['framework']
    for gnn, seed in [('gcn', 42), ('gat', 21)]:
        g = synth_graph(120, 500, 7, seed=51)
        E, n = g['train'], g['num_nodes']
        # IN / OUT candidate masks as prepare_dataset.py:205-214 defines them
        _, _, m2 = pyg.k_hop_subgraph(g['test_pos'].flatten().unique(), 2, E, n)
        cand = {'in': m2, 'out': ~m2}
        tmp = tempfile.mkdtemp()
        os.makedirs(os.path.join(tmp, 'data', 'Cora'))
        d0 = Bag(x=g['x'], num_nodes=n, train_pos_edge_index=E,
                 val_pos_edge_index=g['val_pos'], val_neg_edge_index=g['val_neg'],
                 test_pos_edge_index=g['test_pos'], test_neg_edge_index=g['test_neg'])
        with open(os.path.join(tmp, 'data', 'Cora', f'd_{seed}.pkl'), 'wb') as f:
            pickle.dump((FakeDataset(7), d0), f)
        torch.save(cand, os.path.join(tmp, 'data', 'Cora', f'df_{seed}.pt'))

        def get_model(args, m1=None, m2_=None, num_nodes=None, num_edge_type=None):
            return getattr(D, GNN_CLASS[args.gnn])(args, m1, m2_)
        fw.get_model, fw.get_trainer = get_model, Recorder
        ck = os.path.join(tmp, 'ck')
        op = os.path.join(ck, 'Cora', gnn, 'original', str(seed))
        os.makedirs(op)
        a0 = types.SimpleNamespace(in_dim=7, hidden_dim=128, out_dim=64)
        base = getattr(D, GNN_CLASS[gnn])(a0)
        torch.save({'model_state': base.state_dict()}, os.path.join(op, 'model_best.pt'))
        for df, size, lt in [('out', 2.5, 'both_layerwise'), ('in', 100, 'both_all')]:
            argv = ['--unlearning_model', 'gnndelete_nodeemb', '--gnn', gnn, '--dataset', 'Cora', '--df', df,
                    '--df_size', str(size), '--random_seed', str(seed), '--data_dir', os.path.join(tmp, 'data'),
                    '--checkpoint_dir', ck, '--loss_type', lt]
            old_argv, old_cwd = sys.argv, os.getcwd()
            sys.argv = ['delete_gnn.py'] + argv
            os.chdir(tmp)
            sys.path.insert(0, REF)
            try:
                runpy.run_path(os.path.join(REF, 'delete_gnn.py'), run_name='ref_delete_gnn')['main']()
            finally:
                sys.argv = old_argv
                os.chdir(old_cwd)
                sys.path.remove(REF)
                torch.autograd.set_detect_anomaly(False)
            d = captured['data']
            out = {'in::train': np_(E), 'in::cand': np_(cand[df]), 'in::num_nodes': np.int64(n),
                   'in::df_size': np.float64(size), 'in::seed': np.int64(seed)}
            for k in ['train_pos_edge_index', 'df_mask', 'dr_mask', 'sdf_mask', 'sdf_node_1hop_mask',
                      'sdf_node_2hop_mask', 'directed_df_edge_index']:
                out[f'out::{k}'] = np_(d[k])
            out['out::n_opt'] = np.int64(len(captured['opt']) if isinstance(captured['opt'], list) else 1)
            out['out::ckpt_dir'] = np.array(os.path.relpath(captured['args'].checkpoint_dir, ck))
            out['out::in_dim'] = np.int64(captured['args'].in_dim)
            np.savez_compressed(os.path.join(HERE, f'prep_{gnn}_{df}.npz'), **out)


def write_manifest(crash):
    path = os.path.join(HERE, 'MANIFEST.json')
    if crash is None:                                            # keep the recorded upstream error text
        with open(path) as f:
            crash = json.load(f).get('gcn_both_layerwise_upstream_error')
    prev = {}
    if os.path.exists(path):
        with open(path) as f:
            prev = json.load(f)
    with open(path, 'w') as f:
        json.dump({'generated_by': 'tests/golden/make_golden.py', 'reference': REF, 'torch': torch.__version__,
                   'gcn_both_layerwise_upstream_error': crash,
                   # GNNDeleteTrainer.train_minibatch as delete_gnn.py calls it (no data.dtrain_mask); the
                   # traj_edgeprob_minibatch_* fixtures were recorded with dtrain_mask = dr_mask injected
                   'edgeprob_minibatch_upstream_error': STATE.get('edgeprob_minibatch_upstream_error') or prev.get('edgeprob_minibatch_upstream_error'),
                   'files': sorted(x for x in os.listdir(HERE) if x.endswith(('.npz', '.json')))}, f, indent=1)


def main():
    torch.set_num_threads(4)
    D, T, TE, A, U, B = load_reference()
    if sys.argv[1:] == ['edgeprob']:            # add these fixtures without rewriting the others
        golden_edgeprob_trajectories(D, TE, A)
        return
    if sys.argv[1:] == ['original']:
        golden_original_training(B, A)
        return
    if sys.argv[1:] == ['nodecls']:
        golden_nodecls_trajectory(D, T, A)
        return
    if sys.argv[1:] == ['rgat']:
        golden_rgat(D, A)
        return
    if sys.argv[1:] == ['round2']:              # the fixtures added in round 2 (+ the manifest)
        golden_minibatch(D, T, A)
        golden_kg(D, T, A, B)
        golden_retrain(A)
        golden_split()
        golden_wide_trajectories(D, T, A)
        golden_original_minibatch(B, A)
        write_manifest(None)
        return
    if sys.argv[1:] == ['wide']:
        golden_wide_trajectories(D, T, A)
        return
    if sys.argv[1:] == ['retrain_kg']:          # round 3
        golden_retrain_kg(A)
        write_manifest(None)
        return
    if sys.argv[1:] == ['process_kg']:
        golden_process_kg()
        write_manifest(None)
        return
    if sys.argv[1:] == ['orig_minibatch']:
        golden_original_minibatch(B, A)
        write_manifest(None)
        return
    if sys.argv[1:] == ['losses']:              # round 6 (+ rbf_cka with an explicit sigma)
        golden_losses(T)
        write_manifest(None)
        return
    if sys.argv[1:] == ['edgeprob_minibatch']:  # round 4
        golden_edgeprob_minibatch(D, TE, A)
        write_manifest(None)
        return
    golden_del_layer(D)
    golden_losses(T)
    golden_wiring(D, A)
    golden_trajectories(D, T, A)
    golden_edgeprob_trajectories(D, TE, A)
    golden_original_training(B, A)
    golden_nodecls_trajectory(D, T, A)
    golden_rgat(D, A)
    golden_minibatch(D, T, A)
    golden_edgeprob_minibatch(D, TE, A)
    golden_kg(D, T, A, B)
    golden_retrain(A)
    golden_retrain_kg(A)
    golden_split()
    golden_wide_trajectories(D, T, A)
    golden_original_minibatch(B, A)
    crash = golden_gcn_layerwise_crash(D, T, A)
    golden_parse_args(A)
    golden_eval(D, T, A)
    golden_neg_kg(U)
    golden_prep(D)
    write_manifest(crash)
    print('golden vectors written to', HERE)


if __name__ == '__main__':
    main()
