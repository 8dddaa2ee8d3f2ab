"""Readers for the raw files the reference's datasets are distributed as, used by prepare_dataset.py when such
files are present under <data_dir>/<name>/raw/ (nothing can be downloaded in this environment, so these are
exercised with small files written in the same layouts: tests/test_host_utils.py).

  CitationFull (Cora, Cora_ML, CiteSeer, DBLP, PubMed - prepare_dataset.py:141-142): one `<name>.npz` with a
  CSR adjacency (`adj_data/adj_indices/adj_indptr/adj_shape`), a CSR attribute matrix (`attr_*`) and `labels`;
  torch_geometric reads it as: attributes binarised (x > 0 -> 1), adjacency -> COO, self loops removed, made
  undirected; the reference then row-normalises the features (T.NormalizeFeatures).

  ogbl-collab (prepare_dataset.py:147-148): `edge.csv.gz` (one `src,dst` pair per line, no header) and
  `node-feat.csv.gz` (one comma-separated feature row per node); OGB adds the inverse edges.

Both return (x float32 [N, F], unique undirected edges as `row < col` int64 [2, M], y or None).

  ogbl-biokg (prepare_dataset.py:300-353): OGB's pre-made split `split/random/{train,valid,test}.pt` - torch-saved
  dicts of `head / relation / tail` arrays with per-type local entity ids, `head_type / tail_type` string lists and,
  for valid / test, `head_neg / tail_neg` [n, 500] - plus `raw/num-node-dict.csv.gz` (one column per entity type).
  -> the Data fields the reference pickles and the IN / OUT candidate masks (process_ogbl_kg)."""
import os

import numpy as np
import torch


def _unique_row_lt_col(row, col, n):
    row, col = torch.as_tensor(row, dtype=torch.long), torch.as_tensor(col, dtype=torch.long)
    keep = row != col
    lo, hi = torch.minimum(row[keep], col[keep]), torch.maximum(row[keep], col[keep])
    key = torch.unique(lo * n + hi)
    return torch.stack([key // n, key % n])


def read_citation_full(path):
    import scipy.sparse as sp
    with np.load(path, allow_pickle=True) as f:
        x = sp.csr_matrix((f['attr_data'], f['attr_indices'], f['attr_indptr']), tuple(f['attr_shape'])).todense()
        adj = sp.csr_matrix((f['adj_data'], f['adj_indices'], f['adj_indptr']), tuple(f['adj_shape'])).tocoo()
        y = torch.from_numpy(np.asarray(f['labels'])).long() if 'labels' in f.files else None
    x = torch.from_numpy(np.asarray(x)).float()
    x[x > 0] = 1
    x = x / x.sum(1, keepdim=True).clamp(min=1)                       # T.NormalizeFeatures
    n = x.shape[0]
    return x, _unique_row_lt_col(adj.row, adj.col, n), y


def read_ogbl_collab(raw_dir):
    import pandas as pd
    edges = pd.read_csv(os.path.join(raw_dir, 'edge.csv.gz'), header=None).values.T
    x = torch.from_numpy(pd.read_csv(os.path.join(raw_dir, 'node-feat.csv.gz'), header=None).values).float()
    return x, _unique_row_lt_col(edges[0], edges[1], x.shape[0]), None


def process_ogbl_kg(split_edge, num_nodes_dict, rev_offset=51):
    """process_kg's ogbl branch (prepare_dataset.py:300-399) on OGB's in-memory split: global entity id = local id + the
    running offset of its type (dict order); negatives = (head, first corrupted tail as stored); training triples
    between two types as they are, inside one type once (head < tail) and appended behind the others; inverse triples
    with relation + 51 (upstream's constant); Df candidates inside / outside the 2-hop enclosing subgraph of the test
    triples.  Vectorised (4.8 M triples in seconds, upstream walks them in Python).  -> (Data, {'in', 'out'})."""
    from .data import Data
    from .graph_utils import k_hop_subgraph
    names = list(num_nodes_dict)
    offset = np.concatenate([[0], np.cumsum([int(num_nodes_dict[k]) for k in names])])
    code = {k: i for i, k in enumerate(names)}
    n_entity = int(offset[-1])

    def to_global(d):
        ht = np.fromiter((code[t] for t in d['head_type']), dtype=np.int64, count=len(d['head_type']))
        tt = np.fromiter((code[t] for t in d['tail_type']), dtype=np.int64, count=len(d['tail_type']))
        head = torch.from_numpy(np.asarray(d['head'], dtype=np.int64) + offset[ht])
        tail = torch.from_numpy(np.asarray(d['tail'], dtype=np.int64) + offset[tt])
        return head, tail, torch.from_numpy(ht), torch.from_numpy(tt)

    fields = {}
    for name, key in (('val', 'valid'), ('test', 'test')):
        d = split_edge[key]
        head, tail, _, _ = to_global(d)
        fields[f'{name}_pos_edge_index'] = torch.stack([head, tail])
        fields[f'{name}_edge_type'] = torch.as_tensor(np.asarray(d['relation'])).long()
        fields[f'{name}_neg_edge_index'] = torch.stack([head, torch.as_tensor(np.asarray(d['tail_neg']))[:, 0].long()])
    d = split_edge['train']
    head, tail, ht, tt = to_global(d)
    rel = torch.as_tensor(np.asarray(d['relation'])).long()
    directed = ht != tt
    once = (~directed) & (head < tail)
    order = torch.cat([directed.nonzero().flatten(), once.nonzero().flatten()])
    train = torch.stack([head[order], tail[order]])
    train_type = rel[order]
    data = Data(x=torch.arange(n_entity), num_nodes=n_entity, num_features=0, train_pos_edge_index=train,
                train_edge_type=train_type, edge_index=torch.cat([train, train.flip(0)], 1),
                edge_type=torch.cat([train_type, train_type + rev_offset]), **fields)
    _, _, _, local = k_hop_subgraph(fields['test_pos_edge_index'].flatten().unique(), 2, train, num_nodes=n_entity)
    return data, {'in': local, 'out': ~local}


def read_ogbl_biokg(root):
    import pandas as pd
    counts = pd.read_csv(os.path.join(root, 'raw', 'num-node-dict.csv.gz'))
    num_nodes_dict = {k: int(counts[k][0]) for k in counts.columns}
    split = {k: torch.load(os.path.join(root, 'split', 'random', f'{k}.pt'), weights_only=False) for k in ('train', 'valid', 'test')}
    return process_ogbl_kg(split, num_nodes_dict)


RAW_FILES = {'Cora': 'cora.npz', 'Cora_ML': 'cora_ml.npz', 'CiteSeer': 'citeseer.npz', 'DBLP': 'dblp.npz',
             'PubMed': 'pubmed.npz'}


def load_raw(name, data_dir):
    """(x, edges, y) for a reference dataset name whose raw files exist under data_dir, else None."""
    if name in RAW_FILES:
        path = os.path.join(data_dir, name, 'raw', RAW_FILES[name])
        return read_citation_full(path) if os.path.exists(path) else None
    if name == 'ogbl-collab':
        raw = os.path.join(data_dir, 'ogbl_collab', 'raw')
        return read_ogbl_collab(raw) if os.path.exists(os.path.join(raw, 'edge.csv.gz')) else None
    return None


def load_raw_kg(name, data_dir):
    """(Data, Df candidate masks) for a knowledge graph whose OGB files exist under data_dir, else None."""
    if name == 'ogbl-biokg':
        root = os.path.join(data_dir, 'ogbl_biokg')
        return read_ogbl_biokg(root) if os.path.exists(os.path.join(root, 'split', 'random', 'train.pt')) else None
    return None
