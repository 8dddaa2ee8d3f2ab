#!/usr/bin/env python3
"""CPU-only planner STUDY for the row-partitioned step on bench.py's request: halo rows / bytes per rank
at P = 2 / 4 / 8 for (a) the engine's partition - equal contiguous blocks of the locality order - and two variants that are
NOT built into the engine: (b) the top-k most-referenced source rows replicated on every rank (their layer inputs are
recomputed locally: x is replicated, so x W1^T and the aggregation of a hub's own in-neighbours need no exchange) and taken out
of the halo lists, (c) block limits moved to community boundaries of the label-propagation order and balanced by in-edges
instead of by rows.  Forward halo = distinct source rows of the own target rows that another rank owns (64-float rows of t2);
backward halo = the same on the transposed graph restricted to the S1 rows.  Writes profiles/r04_halo_plan_synth_collab.json."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def halo(rows_t, cols_s, owner, world, mask_rows=None, skip=None):
    """rows_t: target of each edge, cols_s: source; owner[node] = rank.  -> per-rank distinct remote source rows."""
    keep = owner[rows_t] != owner[cols_s]
    if mask_rows is not None:
        keep &= mask_rows[rows_t]
    if skip is not None:
        keep &= ~skip[cols_s]
    key = torch.unique(owner[rows_t][keep] * owner.numel() + cols_s[keep])
    return torch.bincount(key // owner.numel(), minlength=world)


def main():
    from gnndelete_amd.collectives import row_blocks
    from gnndelete_amd.framework.data import prepare_edge_deletion, resolve_df_size
    from gnndelete_amd.framework.synth import make_linkpred_dataset
    from gnndelete_amd.framework.utils import seed_everything
    from gnndelete_amd.reorder import locality_order
    data, df_masks = make_linkpred_dataset('synth-collab', seed=42)
    seed_everything(42)
    prepare_edge_deletion(data, df_masks['in'], resolve_df_size(5.0, data.train_pos_edge_index.shape[1]))
    n = int(data.num_nodes)
    E = data.train_pos_edge_index[:, data.sdf_mask]
    perm, inv = locality_order(E, n)
    src, dst = inv[E[0]], inv[E[1]]
    loops = torch.arange(n)
    src, dst = torch.cat([src, loops]), torch.cat([dst, loops])
    key = torch.unique(dst * n + src)
    dst, src = key // n, key % n
    m1 = data.sdf_node_1hop_mask[perm]
    indeg = torch.bincount(dst, minlength=n)
    refs = torch.bincount(src, minlength=n)                     # how often a row is gathered
    out = {'workload': 'synth-collab GCN 5% IN (bench.py)', 'num_nodes': n, 'row_bytes_d64': 256,
           'note': 'STUDY: variants (b) and (c) are not built into dist_engine.py; (a) is what the engine does', 'per_world': {}}
    for world in (2, 4, 8):
        chunk, _ = row_blocks(n, world)
        own_a = torch.clamp(torch.arange(n) // chunk, max=world - 1)
        res = {}

        def report(tag, owner, skip=None, extra=None):
            f = halo(dst, src, owner, world, None, skip)
            b = halo(src, dst, owner, world, m1, skip)          # transposed graph: targets = sources, S1 rows only
            rows = torch.bincount(owner, minlength=world)
            e = {'own_rows_min_max': [int(rows.min()), int(rows.max())],
                 'recv_rows_forward_max': int(f.max()), 'recv_rows_backward_max': int(b.max()),
                 'recv_MB_per_rank_max': round(int((f + b).max()) * 256 / 1e6, 2),
                 'recv_MB_all_ranks': round(int((f + b).sum()) * 256 / 1e6, 2)}
            if extra:
                e.update(extra)
            res[tag] = e
        report('a_equal_blocks_of_the_locality_order', own_a)
        for k in (1024, 4096, 16384):
            hubs = torch.zeros(n, dtype=torch.bool)
            hubs[torch.topk(refs, k).indices] = True
            report(f'b_top_{k}_gathered_rows_replicated', own_a, skip=hubs,
                   extra={'replicated_rows': k, 'replicated_rows_in_edges': int(indeg[hubs].sum()),
                          'share_of_all_gathers': round(float(refs[hubs].sum()) / float(refs.sum()), 3)})
        # (c) limits balanced by in-edges (the aggregation's work) instead of rows, contiguous in the locality order
        cost = torch.cumsum((indeg + 24).double(), 0)
        cuts = torch.searchsorted(cost, cost[-1] * torch.arange(1, world, dtype=torch.float64) / world)
        own_c = torch.bucketize(torch.arange(n), cuts, right=True)
        report('c_blocks_balanced_by_in_edges', own_c)
        out['per_world'][world] = res
    path = os.path.join(ROOT, 'profiles', 'r04_halo_plan_synth_collab.json')
    with open(path, 'w') as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
