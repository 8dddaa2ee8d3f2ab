#!/usr/bin/env python3
"""Write the datasets the CLIs read: <data_dir>/<name>/d_<seed>.pt (the Data dict: x, num_nodes,
num_features, train_pos_edge_index as directed row<col edges, val/test pos/neg edges) and
df_<seed>.pt ({'in': mask, 'out': mask} over the train edges) - the content of the reference's
d_<seed>.pkl / df_<seed>.pt (prepare_dataset.py:186-264), as plain tensors instead of pickled
torch_geometric objects.  No real dataset can be downloaded here, so `--dataset` normally names one of
the seeded synthetic stand-ins of gnndelete_amd.framework.synth (same shapes as the originals); a
reference dataset name (Cora, DBLP, PubMed, ogbl-collab, ...) is accepted when its raw files already sit
under <data_dir>/<name>/raw/ (gnndelete_amd/framework/raw_readers.py), and goes through the same split.

  python prepare_dataset.py --dataset synth-dblp --seeds 42 21 13 87 100"""
import argparse
import os

import torch

from gnndelete_amd.framework.raw_readers import RAW_FILES, load_raw, load_raw_kg
from gnndelete_amd.framework.synth import KG_SHAPES, SHAPES, make_kg_dataset, make_linkpred_dataset, split_linkpred


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--dataset', default='synth-dblp',
                   choices=sorted(SHAPES) + sorted(KG_SHAPES) + sorted(RAW_FILES) + ['ogbl-collab', 'ogbl-biokg'])
    p.add_argument('--data_dir', default='./data')
    p.add_argument('--seeds', type=int, nargs='+', default=[42, 21, 13, 87, 100])
    a = p.parse_args()
    out = os.path.join(a.data_dir, a.dataset)
    os.makedirs(out, exist_ok=True)
    raw = raw_kg = None
    if a.dataset == 'ogbl-biokg':
        raw_kg = load_raw_kg(a.dataset, a.data_dir)              # OGB ships the split: the same for every seed (as upstream)
    if a.dataset not in SHAPES and a.dataset not in KG_SHAPES and raw_kg is None:
        raw = load_raw(a.dataset, a.data_dir)
        if raw is None:
            raise SystemExit(f"no raw files for '{a.dataset}' under {a.data_dir} (nothing can be downloaded here); "
                             f"use one of the synthetic stand-ins: {sorted(SHAPES) + sorted(KG_SHAPES)}")
    for seed in a.seeds:
        if raw_kg is not None:
            data, df = raw_kg
        elif raw is not None:
            x, edges, _ = raw
            data, df = split_linkpred(x, edges, x.shape[0], torch.Generator().manual_seed(seed))
        else:
            make = make_kg_dataset if a.dataset in KG_SHAPES else make_linkpred_dataset
            data, df = make(a.dataset, seed=seed)
        data.save(os.path.join(out, f'd_{seed}.pt'))
        torch.save(df, os.path.join(out, f'df_{seed}.pt'))
        print(seed, data, {k: int(v.sum()) for k, v in df.items()})


if __name__ == '__main__':
    main()
