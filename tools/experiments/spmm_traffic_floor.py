#!/usr/bin/env python3
"""Lower bound on the L2-to-fabric traffic of the layer-1 SpMM of bench.py's request (CPU only, exact).

The SpMM hands each of the 8 XCDs one contiguous range of work items (csrc/spmm.hip).  The eight L2s are private
(4 MiB each, not coherent with each other), so a source row x_j has to cross the fabric at least once for EVERY XCD
that gathers it, however large that XCD's L2 were.  This script counts exactly that on the bench graph in the
engine's own node order:

  floor_inf   = 8 L2s of unlimited size: sum over XCDs of the distinct x rows it gathers (+ indices, + y once)
  lru         = the same with a 4 MiB fully associative LRU per XCD over the kernel's visiting order (1024 waves
                of an XCD sweep consecutive items together), x rows and y rows both allocating
  any_order   = what no row order / XCD assignment can avoid: edges of the random (non-community) part of the
                generator, each needing its source row in the gathering XCD, counted by distinct (XCD, source) pairs
                under the BEST case that every community edge is free

  --layouts   (round 5) the same two figures for a ROW-RANGE x COLUMN-SLICE assignment of the XCDs:
              "RxC" = R contiguous row ranges, each worked on by C XCDs that own d / C feature columns of every row
              (8x1 = what the kernel does; 4x2 and 2x4 trade wider row ranges - fewer distinct (XCD, source) pairs, rows of
              d / C floats so the 4 MiB hold C x as many of them - against reading the index arrays C times and gathering
              shorter pieces: 4 d / C bytes, a 128-byte line at d = 128 / C = 4 and at d = 64 / C = 2)

Prints JSON; DESIGN.md section 7 and profiles/r02_spmm_bound.md quote it next to the PMC counters of the kernel;
profiles/r05_spmm_2d_floor.json holds the --layouts runs."""
import argparse
import json
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--workload', default='synth-collab')
    p.add_argument('--df', default='in')
    p.add_argument('--df_size', type=float, default=5.0)
    p.add_argument('--seed', type=int, default=42)
    p.add_argument('--d', type=int, default=128)
    p.add_argument('--l2_mib', type=float, default=4.0)
    p.add_argument('--no_lru', action='store_true')
    p.add_argument('--layouts', default='', help='comma list of RxC (R x C = 8), e.g. 8x1,4x2,2x4')
    args = p.parse_args()

    from gnndelete_amd.framework.data import prepare_edge_deletion, resolve_df_size
    from gnndelete_amd.framework.synth import make_linkpred_dataset
    from gnndelete_amd.framework.utils import seed_everything
    from gnndelete_amd.reorder import locality_order

    data, df_masks = make_linkpred_dataset(args.workload, seed=args.seed)
    seed_everything(args.seed)
    size = resolve_df_size(args.df_size, data.train_pos_edge_index.shape[1])
    prepare_edge_deletion(data, df_masks[args.df], size)
    n = int(data.num_nodes)
    E = data.train_pos_edge_index[:, data.sdf_mask]
    perm, inv = locality_order(E, n)
    src, dst = inv[E[0]].numpy(), inv[E[1]].numpy()
    keep = src != dst
    src, dst = src[keep], dst[keep]
    # GCN structure: one self loop per node; CSR over targets sorted by (target, source)
    loops = np.arange(n)
    src = np.concatenate([src, loops])
    dst = np.concatenate([dst, loops])
    key = dst.astype(np.int64) * n + src
    key = np.unique(key)
    dst, src = (key // n).astype(np.int64), (key % n).astype(np.int64)
    nnz = key.size
    rowptr = np.zeros(n + 1, np.int64)
    np.add.at(rowptr, dst + 1, 1)
    rowptr = np.cumsum(rowptr)
    deg = np.diff(rowptr)
    # work items: rows cut into pieces of <= 64 edges
    pieces = np.maximum((deg + 63) // 64, 1)
    item_row = np.repeat(np.arange(n), pieces)
    first = np.cumsum(pieces) - pieces
    k = np.arange(item_row.size) - first[item_row]
    it_start = rowptr[item_row] + k * 64
    it_end = np.minimum(rowptr[item_row + 1], it_start + 64)
    n_items = item_row.size
    per = (n_items + 7) // 8
    row_bytes = 4 * args.d
    out = {'workload': args.workload, 'n': n, 'nnz': int(nnz), 'd': args.d, 'n_items': int(n_items),
           'algorithmic_bytes': int(4 * (n + 1) + 8 * nnz + 8 * n * args.d)}
    edge_item = np.repeat(np.arange(n_items), it_end - it_start)          # item of every CSR entry, in CSR order
    edge_xcd = edge_item // per
    pair = np.unique(edge_xcd.astype(np.int64) * n + src)
    distinct = np.bincount((pair // n).astype(np.int64), minlength=8)
    gathers = np.bincount(edge_xcd, minlength=8)
    out['gathers_per_xcd'] = gathers.tolist()
    out['distinct_source_rows_per_xcd'] = distinct.tolist()
    x_floor = int(distinct.sum()) * row_bytes
    fixed = 4 * (n + 1) + 8 * nnz + 4 * n * args.d + 16 * n_items          # indices, y once, item records
    out['floor_inf_l2_bytes'] = x_floor + fixed
    out['floor_inf_l2_over_algorithmic'] = round((x_floor + fixed) / out['algorithmic_bytes'], 3)
    # how local is the order: gathers whose source lies within +-W rows of the target
    for w in (64, 128, 512, 4096):
        out[f'frac_gathers_within_{w}_rows'] = round(float(np.mean(np.abs(src - dst) <= w)), 4)

    if not args.no_lru:
        cap_rows = int(args.l2_mib * 1048576 // row_bytes)
        misses_x = 0
        for xcd in range(8):
            i0, i1 = xcd * per, min(n_items, (xcd + 1) * per)
            cache = OrderedDict()
            # the XCD's 1024 resident waves visit items i0 + t*1024 + w: in time order that is simply item order
            for i in range(i0, i1):
                for c in src[it_start[i]:it_end[i]]:
                    if c in cache:
                        cache.move_to_end(c)
                    else:
                        misses_x += 1
                        cache[c] = None
                        if len(cache) > cap_rows:
                            cache.popitem(last=False)
                # the output row allocates in the same L2 (write-back cache); tagged so it cannot alias an x row
                yk = -1 - int(item_row[i])
                cache[yk] = None
                if len(cache) > cap_rows:
                    cache.popitem(last=False)
        out['lru_l2_mib'] = args.l2_mib
        out['lru_x_miss_rows'] = int(misses_x)
        out['lru_bytes'] = int(misses_x) * row_bytes + fixed
        out['lru_over_algorithmic'] = round(out['lru_bytes'] / out['algorithmic_bytes'], 3)
    if args.layouts:
        out['layouts'] = {}
        for lay in args.layouts.split(','):
            R, C = (int(v) for v in lay.split('x'))
            assert R * C == 8 and args.d % (4 * C) == 0
            slice_bytes = row_bytes // C
            per_r = (n_items + R - 1) // R
            edge_rr = edge_item // per_r
            pair = np.unique(edge_rr.astype(np.int64) * n + src)
            distinct_r = np.bincount((pair // n).astype(np.int64), minlength=R)
            # every one of the C XCDs of a row range fetches its slice of each distinct source row; indices and item records are
            # read by all C of them; y is written once (in slices)
            fixed_c = C * (4 * (n + 1) + 8 * nnz + 16 * n_items) + 4 * n * args.d
            rec = {'row_ranges': R, 'column_slices': C, 'gathered_piece_bytes': slice_bytes,
                   'distinct_source_rows_per_range': distinct_r.tolist(),
                   'floor_inf_l2_bytes': int(distinct_r.sum()) * slice_bytes * C + fixed_c}
            rec['floor_inf_l2_over_algorithmic'] = round(rec['floor_inf_l2_bytes'] / out['algorithmic_bytes'], 3)
            if not args.no_lru:
                cap = int(args.l2_mib * 1048576 // slice_bytes)
                miss = 0
                for rr in range(R):                         # the C XCDs of a range see the same access stream: simulate one
                    i0, i1 = rr * per_r, min(n_items, (rr + 1) * per_r)
                    cache = OrderedDict()
                    for i in range(i0, i1):
                        for c in src[it_start[i]:it_end[i]]:
                            if c in cache:
                                cache.move_to_end(c)
                            else:
                                miss += 1
                                cache[c] = None
                                if len(cache) > cap:
                                    cache.popitem(last=False)
                        cache[-1 - int(item_row[i])] = None
                        if len(cache) > cap:
                            cache.popitem(last=False)
                rec['lru_x_miss_pieces'] = int(miss) * C
                rec['lru_bytes'] = int(miss) * slice_bytes * C + fixed_c
                rec['lru_over_algorithmic'] = round(rec['lru_bytes'] / out['algorithmic_bytes'], 3)
            out['layouts'][lay] = rec
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
