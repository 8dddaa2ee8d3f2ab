#!/usr/bin/env python3
"""CPU-only planner report for the row-partitioned step (gnndelete_amd/dist_engine.py) on bench.py's request:
rows and bytes every rank receives in the two halo exchanges per step, per pair and in total, against the dense
all-gather of the same matrices, for P = 2 / 4 / 8 ranks.  xGMI is point-to-point (one link per pair of GPUs), so the
per-PAIR maximum is what bounds an exchange."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    from gnndelete_amd.collectives import halo_plan, row_blocks
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

    def csr(rows, cols):
        order = torch.argsort(rows * n + cols)
        rp = torch.zeros(n + 1, dtype=torch.long)
        rp[1:] = torch.cumsum(torch.bincount(rows, minlength=n), 0)
        return rp, cols[order]
    rp, col = csr(dst, src)
    rpt, colt = csr(src, dst)
    out = {'workload': 'synth-collab GCN 5% IN (bench.py)', 'num_nodes': n, 'row_bytes_d64': 256, 'per_world': {}}
    for world in (2, 4, 8):
        chunk, _ = row_blocks(n, world)
        f = halo_plan(rp, col, n, 0, world, chunk)
        b = halo_plan(rpt, colt, n, 0, world, chunk, m1)
        pf, pb = torch.tensor(f.pair_counts), torch.tensor(b.pair_counts)
        recv = (pf + pb).sum(1)                         # rows received per rank per step (both exchanges)
        pair = pf + pf.t() + pb + pb.t()                # rows crossing a pair's link per step, both directions
        dense = 2 * (n - chunk)
        need1 = [int(pf[q].sum()) for q in range(world)]
        out['per_world'][world] = {
            'rows_per_rank': chunk,
            'recv_rows_per_rank_max': int(recv.max()), 'recv_MB_per_rank_max': round(int(recv.max()) * 256 / 1e6, 2),
            'pair_rows_max': int(pair.max()), 'pair_MB_max': round(int(pair.max()) * 256 / 1e6, 2),
            'allgather_rows_per_rank': dense, 'allgather_MB_per_rank': round(dense * 256 / 1e6, 2),
            'halo_over_allgather': round(float(recv.max()) / dense, 3),
            'layer1_rows_recomputed_max': max(need1), 'layer1_recompute_over_own': round(max(need1) / chunk, 2),
            'allreduce_bytes': 4 * (128 * 128 + 64 * 64 + 4)}
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
