"""Readers for the raw files the reference's datasets are distributed as, used by prepare_dataset.py when such
files are present under <data_dir>/<name>/raw/ (nothing can be downloaded in this environment, so these are
exercised with small files written in the same layouts: tests/test_host_utils.py).

  CitationFull (Cora, Cora_ML, CiteSeer, DBLP, PubMed - prepare_dataset.py:141-142): one `<name>.npz` with a
  CSR adjacency (`adj_data/adj_indices/adj_indptr/adj_shape`), a CSR attribute matrix (`attr_*`) and `labels`;
  torch_geometric reads it as: attributes binarised (x > 0 -> 1), adjacency -> COO, self loops removed, made
  undirected; the reference then row-normalises the features (T.NormalizeFeatures).

  ogbl-collab (prepare_dataset.py:147-148): `edge.csv.gz` (one `src,dst` pair per line, no header) and
  `node-feat.csv.gz` (one comma-separated feature row per node); OGB adds the inverse edges.

Both return (x float32 [N, F], unique undirected edges as `row < col` int64 [2, M], y or None)."""
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
