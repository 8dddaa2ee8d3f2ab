"""Worker run under torch.distributed (world_size >= 2) by tests/test_dist_*.py.

mode=cpu : gloo on CPU tensors - the collectives, the row-partition planner, and a dense-torch
           emulation of the partitioned step (same algebra as gnndelete_amd.dist_engine, no HIP)
           checked against single-process autograd.
mode=gpu : every rank on cuda:0 over gloo (one-GPU box) - the real PartitionedNodeembEngine
           against the single-process NodeembEngine on the same request."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def small_request(seed=3, n=300, m=1400, f=12, h=32, o=16):
    from gnndelete_amd.framework.data import prepare_edge_deletion
    from gnndelete_amd.framework.graph_utils import negative_sampling
    from gnndelete_amd.framework.synth import make_linkpred_dataset
    data, dfm = make_linkpred_dataset(None, seed=seed, shape=(n, f, m, 'dense'))
    torch.manual_seed(seed)
    prepare_edge_deletion(data, dfm['in'], 40)
    gen = torch.Generator().manual_seed(seed)
    neg = negative_sampling(data.train_pos_edge_index, data.num_nodes, int(data.df_mask.sum()), generator=gen)
    keep = torch.ones(data.num_nodes, dtype=torch.bool)
    keep[data.directed_df_edge_index.flatten().unique()] = False
    return data, neg, data.sdf_node_1hop_mask & keep, data.sdf_node_2hop_mask & keep, (f, h, o)


def cpu_checks(rank, world):
    from gnndelete_amd.collectives import all_gather_rows, all_reduce_sum, row_blocks
    from gnndelete_amd.engine import _LayerTerms
    from gnndelete_amd.graph import SplitPlan
    from oracle import gnndelete_ref as R
    from oracle import pyg_semantics as pyg

    # ---- collectives
    chunk, n_pad = row_blocks(10, world)
    full = torch.zeros(n_pad, 3)
    full[rank * chunk:(rank + 1) * chunk] = rank + 1
    all_gather_rows(full, rank, world, chunk)
    for r in range(world):
        assert torch.all(full[r * chunk:(r + 1) * chunk] == r + 1)
    buf = torch.full((5,), float(rank + 1))
    all_reduce_sum(buf, world)
    assert torch.all(buf == sum(range(1, world + 1)))

    # ---- planner: the ranks' work items tile the global plan exactly
    data, neg, ni1, ni2, (f, h, o) = small_request()
    n = data.num_nodes
    E = data.train_pos_edge_index
    e_sdf = E[:, data.sdf_mask]
    ei, w = pyg.gcn_norm(e_sdf, n)
    order = torch.argsort(ei[1] * n + ei[0])
    rowptr = torch.zeros(n + 1, dtype=torch.long)
    rowptr[1:] = torch.cumsum(torch.bincount(ei[1], minlength=n), 0)
    chunk, n_pad = row_blocks(n, world)
    lo, hi = min(n, rank * chunk), min(n, (rank + 1) * chunk)
    mine = SplitPlan(rowptr.int(), chunk=8, row_range=(lo, hi))
    whole = SplitPlan(rowptr.int(), chunk=8)
    sel = (whole.items[:, 0] >= lo) & (whole.items[:, 0] < hi)
    assert torch.equal(mine.items[:, :3], whole.items[sel][:, :3])
    counts = torch.tensor([mine.n_items, mine.n_slots], dtype=torch.float32)
    all_reduce_sum(counts, world)
    assert counts.tolist() == [whole.n_items, whole.n_slots]

    # ---- sparse (halo) exchange delivers exactly the rows the local SpMM gathers
    from gnndelete_amd.collectives import exchange_rows, halo_lists
    col = ei[0][order]
    send_rows, in_splits, recv_rows, out_splits = halo_lists(rowptr, col, n, rank, world, chunk)
    truth = torch.arange(n_pad, dtype=torch.float32)[:, None].repeat(1, 3)          # row r holds value r
    mine_full = torch.full((n_pad, 3), -1.0)
    mine_full[lo:hi] = truth[lo:hi]
    recv = torch.empty(sum(out_splits), 3)
    exchange_rows(mine_full.index_select(0, send_rows), recv, in_splits, out_splits, world)
    mine_full.index_copy_(0, recv_rows, recv)
    gathered = torch.unique(col[int(rowptr[lo]):int(rowptr[hi])])
    assert torch.equal(mine_full[gathered], truth[gathered])
    assert recv_rows.numel() < n - (hi - lo) or world == 1

    # ---- dense emulation of the partitioned step vs single-process autograd
    torch.manual_seed(0)
    model = R.TwoLayerDelete('gcn', f, h, o, data.sdf_node_1hop_mask, data.sdf_node_2hop_mask)
    with torch.no_grad():
        model.deletion1.deletion_weight.copy_(torch.eye(h) * 0.5 + 0.05 * torch.randn(h, h))
        model.deletion2.deletion_weight.copy_(torch.eye(o) * 0.5 + 0.05 * torch.randn(o, o))
    with torch.no_grad():
        z1o, z2o = model.get_original_embeddings(data.x, E[:, data.dr_mask], return_all_emb=True)
    alpha = 0.4
    pos = E[:, data.df_mask]
    z1, z2 = model(data.x, e_sdf, return_all_emb=True)
    r1, r2, l1, l2 = R.nodeemb_terms(z1, z2, z1o, z2o, pos, neg, ni1, ni2, R.LOSSES['mse_mean'])
    (alpha * (r1 + r2) + (1 - alpha) * (l1 + l2)).backward()
    want1, want2 = model.deletion1.deletion_weight.grad, model.deletion2.deletion_weight.grad

    A = torch.zeros(n, n).index_put_((ei[1], ei[0]), w, accumulate=True)          # [target, source]
    W1, b1 = model.conv1.lin.weight.detach(), model.conv1.bias.detach()
    W2, b2 = model.conv2.lin.weight.detach(), model.conv2.bias.detach()
    D1, D2 = model.deletion1.deletion_weight.detach(), model.deletion2.deletion_weight.detach()
    own = torch.zeros(n, dtype=torch.bool)
    own[lo:hi] = True
    s1, s2 = data.sdf_node_1hop_mask & own, data.sdf_node_2hop_mask & own
    t1 = data.x @ W1.t()                                         # replicated
    p1 = A[lo:hi] @ t1 + b1
    z1p = torch.zeros(n_pad, h)
    z1p[lo:hi] = p1
    xs1 = z1p[:n][s1].clone()
    z1p[:n][s1] = xs1 @ D1
    t2 = torch.zeros(n_pad, o)
    t2[lo:hi] = z1p[lo:hi].clamp(min=0) @ W2.t()
    all_gather_rows(t2, rank, world, chunk)                      # exchange 1
    z2p = torch.zeros(n_pad, o)
    z2p[lo:hi] = A[lo:hi] @ t2[:n] + b2
    xs2 = z2p[:n][s2].clone()
    z2p[:n][s2] = xs2 @ D2
    tm1 = _LayerTerms(pos, neg, ni1, z1o, alpha, 1 - alpha, 'mean', (lo, hi))
    tm2 = _LayerTerms(pos, neg, ni2, z2o, alpha, 1 - alpha, 'mean', (lo, hi))

    def loss_grad(tm, z):
        dz = torch.zeros_like(z)
        rows = tm.row_idx.long()
        dz[rows] = tm.coef[:, None] * (z[rows] - tm.tm)
        return dz
    dz1, dz2 = loss_grad(tm1, z1p), loss_grad(tm2, z2p)
    gC = xs2.t() @ dz2[:n][s2]
    dz2[:n][s2] = dz2[:n][s2] @ D2.t()
    all_gather_rows(dz2, rank, world, chunk)                     # exchange 2
    dt2 = A.t()[lo:hi] @ dz2[:n]
    dh = torch.zeros(n, h)
    dh[lo:hi] = dt2 @ W2
    gA = xs1.t() @ dz1[:n][s1]
    gB = xs1.t() @ (dh * (z1p[:n] > 0))[s1]
    pack = torch.cat([gA.flatten(), gB.flatten(), gC.flatten()])
    all_reduce_sum(pack, world)                                  # exchange 3
    g1 = (pack[:h * h] + pack[h * h:2 * h * h]).view(h, h)
    g2 = pack[2 * h * h:].view(o, o)
    assert torch.allclose(g1, want1, rtol=1e-3, atol=1e-7), float((g1 - want1).abs().max())
    assert torch.allclose(g2, want2, rtol=1e-3, atol=1e-7), float((g2 - want2).abs().max())


def gpu_checks(rank, world):
    from types import SimpleNamespace
    from gnndelete_amd.dist_engine import PartitionedNodeembEngine
    from gnndelete_amd.engine import NodeembEngine
    from gnndelete_amd.framework.models import GCNDelete, GINDelete
    dev = torch.device('cuda', 0)
    data, neg, ni1, ni2, (f, h, o) = small_request(n=6000, m=30000, f=32, h=128, o=64)
    for cls, lt in [(GCNDelete, 'both_layerwise'), (GINDelete, 'both_all'), (GCNDelete, 'only2_all')]:
        results = []
        for partitioned in (False, True):
            torch.manual_seed(11)
            m = cls(SimpleNamespace(in_dim=f, hidden_dim=h, out_dim=o), data.sdf_node_1hop_mask,
                    data.sdf_node_2hop_mask).to(dev)
            x, E = data.x.to(dev), data.train_pos_edge_index.to(dev)
            e_sdf = E[:, data.sdf_mask.to(dev)].contiguous()
            with torch.no_grad():
                z1o, z2o = m.get_original_embeddings(x, E[:, data.dr_mask.to(dev)].contiguous(), return_all_emb=True)
            args = (m, x, e_sdf, z1o, z2o, E[:, data.df_mask.to(dev)], neg.to(dev), ni1, ni2)
            if partitioned:
                eng = PartitionedNodeembEngine(*args, rank, world, loss_type=lt, alpha=0.5, lr=1e-2,
                                               exchange='allgather' if lt == 'only2_all' else 'halo')
            else:
                eng = NodeembEngine(*args, loss_type=lt, alpha=0.5, lr=1e-2)
            for _ in range(6):
                eng.step()
            torch.cuda.synchronize()
            results.append((m.deletion1.deletion_weight.detach().cpu(), m.deletion2.deletion_weight.detach().cpu(),
                            eng.loss_history()))
        (a1, a2, ah), (b1, b2, bh) = results
        err = max(float((a1 - b1).norm() / a1.norm()), float((a2 - b2).norm() / a2.norm()))
        assert err < 1e-4, (cls.__name__, lt, err)
        assert torch.allclose(ah, bh, rtol=1e-4), (cls.__name__, lt)
        if rank == 0:
            print(f'{cls.__name__} {lt}: partitioned == single (rel err {err:.2e})', flush=True)


def main():
    mode = sys.argv[1]
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    try:
        (cpu_checks if mode == 'cpu' else gpu_checks)(rank, world)
        dist.barrier()
        if rank == 0:
            print('DIST_OK', flush=True)
    finally:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
