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
    from gnndelete_amd.collectives import exchange_rows, halo_plan
    col = ei[0][order]
    plan = halo_plan(rowptr, col, n, rank, world, chunk)
    truth = torch.arange(n_pad, dtype=torch.float32)[:, None].repeat(1, 3)          # row r holds value r
    mine_full = torch.full((n_pad, 3), -1.0)
    mine_full[lo:hi] = truth[lo:hi]
    recv = torch.empty(max(1, plan.n_recv), 3)
    exchange_rows(mine_full.index_select(0, plan.send_rows), recv, plan, world)
    mine_full.index_copy_(0, plan.recv_rows, recv[:plan.n_recv])
    gathered = torch.unique(col[int(rowptr[lo]):int(rowptr[hi])])
    assert torch.equal(mine_full[gathered], truth[gathered])
    assert plan.n_recv < n - (hi - lo) or world == 1
    # both ends of every pair agree on the counts, and a row-masked plan only asks for what the masked rows gather
    pc = torch.tensor(plan.pair_counts)
    assert plan.in_splits == pc[:, rank].tolist() and plan.out_splits == pc[rank].tolist()
    some = torch.zeros(n, dtype=torch.bool)
    some[::3] = True
    sub = halo_plan(rowptr, col, n, rank, world, chunk, some)
    deg = rowptr[1:] - rowptr[:-1]
    tgt = torch.repeat_interleave(torch.arange(n), deg)
    mine_edges = (tgt >= lo) & (tgt < hi) & some[tgt]
    want = torch.unique(col[mine_edges])
    want = want[(want < lo) | (want >= hi)]
    assert torch.equal(torch.sort(sub.recv_rows).values, want)

    # ---- dense emulation of the partitioned step (dist_engine's segments A-D with the halo exchanges) vs
    # single-process autograd, for the GCN and the GraphSAGE aggregation
    for kind in ('gcn', 'sage'):
        emulate_partitioned_step(kind, rank, world, data, neg, ni1, ni2, f, h, o)


def emulate_partitioned_step(kind, rank, world, data, neg, ni1, ni2, f, h, o):
    from gnndelete_amd.collectives import all_reduce_sum, exchange_rows, halo_plan, row_blocks
    from gnndelete_amd.engine import _LayerTerms
    from oracle import gnndelete_ref as R
    from oracle import pyg_semantics as pyg
    n = data.num_nodes
    E = data.train_pos_edge_index
    e_sdf = E[:, data.sdf_mask]
    chunk, _ = row_blocks(n, world)
    lo, hi = min(n, rank * chunk), min(n, (rank + 1) * chunk)
    torch.manual_seed(0)
    model = R.TwoLayerDelete(kind, f, h, o, data.sdf_node_1hop_mask, data.sdf_node_2hop_mask)
    with torch.no_grad():
        model.deletion1.deletion_weight.copy_(torch.eye(h) * 0.5 + 0.05 * torch.randn(h, h))
        model.deletion2.deletion_weight.copy_(torch.eye(o) * 0.5 + 0.05 * torch.randn(o, o))
        z1o, z2o = model.get_original_embeddings(data.x, E[:, data.dr_mask], return_all_emb=True)
    alpha = 0.4
    pos = E[:, data.df_mask]
    z1, z2 = model(data.x, e_sdf, return_all_emb=True)
    r1, r2, l1, l2 = R.nodeemb_terms(z1, z2, z1o, z2o, pos, neg, ni1, ni2, R.LOSSES['mse_mean'])
    (alpha * (r1 + r2) + (1 - alpha) * (l1 + l2)).backward()
    want1, want2 = model.deletion1.deletion_weight.grad, model.deletion2.deletion_weight.grad

    if kind == 'gcn':
        ei, w = pyg.gcn_norm(e_sdf, n)
        W1, b1, W2, b2 = (model.conv1.lin.weight.detach(), model.conv1.bias.detach(), model.conv2.lin.weight.detach(),
                          model.conv2.bias.detach())
        R1 = R2 = None
    else:
        ei = e_sdf
        w = 1.0 / torch.bincount(ei[1], minlength=n).clamp(min=1).float()[ei[1]]
        W1, b1, W2, b2 = (model.conv1.lin_l.weight.detach(), model.conv1.lin_l.bias.detach(),
                          model.conv2.lin_l.weight.detach(), model.conv2.lin_l.bias.detach())
        R1, R2 = model.conv1.lin_r.weight.detach(), model.conv2.lin_r.weight.detach()
    A = torch.zeros(n, n).index_put_((ei[1], ei[0]), w, accumulate=True)          # [target, source]
    order = torch.argsort(ei[1] * n + ei[0])
    rowptr = torch.zeros(n + 1, dtype=torch.long)
    rowptr[1:] = torch.cumsum(torch.bincount(ei[1], minlength=n), 0)
    col = ei[0][order]
    order_t = torch.argsort(ei[0] * n + ei[1])
    rowptr_t = torch.zeros(n + 1, dtype=torch.long)
    rowptr_t[1:] = torch.cumsum(torch.bincount(ei[0], minlength=n), 0)
    col_t = ei[1][order_t]
    D1, D2 = model.deletion1.deletion_weight.detach(), model.deletion2.deletion_weight.detach()
    own = torch.zeros(n, dtype=torch.bool)
    own[lo:hi] = True
    m1 = data.sdf_node_1hop_mask
    s1, s2 = m1 & own, data.sdf_node_2hop_mask & own
    halo_f = halo_plan(rowptr, col, n, rank, world, chunk)
    halo_b = halo_plan(rowptr_t, col_t, n, rank, world, chunk, m1)
    ownr = torch.arange(lo, hi)
    need1 = torch.unique(torch.cat([ownr, halo_f.recv_rows]))
    bad = float('nan')
    # segment A: only the rows in need1 are ever formed / read
    t1 = torch.full((n, h), bad)
    t1[need1] = data.x[need1] @ W1.t()
    pre1 = torch.full((n, h), bad)
    pre1[lo:hi] = A[lo:hi][:, need1] @ t1[need1] + b1 + (data.x[lo:hi] @ R1.t() if R1 is not None else 0)
    z1p = pre1.clone()
    z1p[s1] = pre1[s1] @ D1
    t2 = torch.full((n, o), bad)
    t2[lo:hi] = z1p[lo:hi].clamp(min=0) @ W2.t()
    t2r = z1p[lo:hi].clamp(min=0) @ R2.t() if R2 is not None else 0
    send = t2.index_select(0, halo_f.send_rows)
    assert not torch.isnan(send).any()
    recv = torch.empty(max(1, halo_f.n_recv), o)
    exchange_rows(send, recv, halo_f, world)                      # all-to-all 1
    t2.index_copy_(0, halo_f.recv_rows, recv[:halo_f.n_recv])
    # segment B
    avail = need1
    p2 = torch.full((n, o), bad)
    p2[lo:hi] = A[lo:hi][:, avail] @ t2[avail] + b2 + t2r
    z2p = p2.clone()
    z2p[s2] = p2[s2] @ D2
    tm1 = _LayerTerms(pos, neg, ni1, z1o, alpha, 1 - alpha, 'mean', (lo, hi))
    tm2 = _LayerTerms(pos, neg, ni2, z2o, alpha, 1 - alpha, 'mean', (lo, hi))

    def loss_grad(tm, z):
        dz = torch.zeros(n, z.shape[1])
        rows = tm.row_idx.long()
        dz[rows] = tm.coef[:, None] * (z[rows] - tm.tm)
        return dz
    dz1, dz2 = loss_grad(tm1, z1p), loss_grad(tm2, z2p)
    g2 = p2[s2].t() @ dz2[s2]
    dp2 = torch.full((n, o), bad)
    dp2[lo:hi] = 0.0
    dp2[s2] = dz2[s2] @ D2.t()
    send = dp2.index_select(0, halo_b.send_rows)
    assert not torch.isnan(send).any()
    recv = torch.empty(max(1, halo_b.n_recv), o)
    exchange_rows(send, recv, halo_b, world)                      # all-to-all 2
    dp2.index_copy_(0, halo_b.recv_rows, recv[:halo_b.n_recv])
    # segment C: the transposed aggregation of the own S1 rows reads own + received rows only
    avail_b = torch.unique(torch.cat([ownr, halo_b.recv_rows]))
    dt2 = A.t()[s1][:, avail_b] @ dp2[avail_b]
    dh = dt2 @ W2 + (dp2[s1] @ R2 if R2 is not None else 0)
    g1 = pre1[s1].t() @ (dz1[s1] + dh * (z1p[s1] > 0))
    pack = torch.cat([g1.flatten(), g2.flatten()])
    all_reduce_sum(pack, world)                                   # all-reduce
    g1, g2 = pack[:h * h].view(h, h), pack[h * h:].view(o, o)
    assert torch.allclose(g1, want1, rtol=1e-3, atol=1e-7), (kind, float((g1 - want1).abs().max()))
    assert torch.allclose(g2, want2, rtol=1e-3, atol=1e-7), (kind, float((g2 - want2).abs().max()))


def gpu_checks(rank, world, rccl=False, direct=False):
    from types import SimpleNamespace
    from gnndelete_amd.dist_engine import PartitionedNodeembEngine
    from gnndelete_amd.engine import NodeembEngine
    from gnndelete_amd.framework.models import GATDelete, GCNDelete, GINDelete, SAGEDelete
    dev = torch.device('cuda', 0)
    data, neg, ni1, ni2, (f, h, o) = small_request(n=6000, m=30000, f=32, h=128, o=64)
    cases = [(GCNDelete, 'both_layerwise'), (GINDelete, 'both_all'), (GCNDelete, 'only2_all'), (SAGEDelete, 'both_layerwise'),
             (SAGEDelete, 'both_all'), (GCNDelete, 'only1'), (GINDelete, 'only2_layerwise'), (GATDelete, 'both_layerwise'),
             (GATDelete, 'both_all')]
    if world > 2:             # three ranks time-share the box's one GPU (10 x the two-rank run): ONE backbone, synchronous and
        cases = [(GCNDelete, 'both_layerwise')]        # overlapped (two ranks run every backbone and loss type)
    group = None
    if rccl:                  # a world of one over RCCL: the data-path communicator the GPU node uses
        cases = [(GCNDelete, 'both_layerwise'), (SAGEDelete, 'both_all'), (GATDelete, 'both_layerwise')]
        if direct:
            from gnndelete_amd.collectives import DirectComm, HaloPlan, all_reduce_sum, exchange_rows
            group = DirectComm(rank, world, dev)
            buf = torch.arange(20480 + 4, dtype=torch.float32, device=dev)
            all_reduce_sum(buf, world, group)
            send = torch.randn(37, 64, device=dev)
            recv = torch.zeros(40, 64, device=dev)
            exchange_rows(send, recv, HaloPlan(None, [37], None, [37], [[37]]), world, group)
            torch.cuda.synchronize()
            assert torch.equal(buf.cpu(), torch.arange(20480 + 4, dtype=torch.float32)) and torch.equal(recv[:37], send)
            print('direct collectives ok', flush=True)
        else:
            group = dist.new_group(backend='nccl', device_id=dev)
            assert dist.get_backend(group) == 'nccl'
    for ci, (cls, lt) in enumerate(cases):
        results = []
        # single GPU | partitioned, synchronous exchanges | exchanges overlapped (at three ranks for the first case only: two
        # ranks run every case both ways, and the suite has a time limit)
        variants = (False, True, 'overlap') if (world == 2 or ci == 0) else (False, True)
        for partitioned in variants:
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
                                               use_graph=(lt != 'only1'), group=group, overlap=(partitioned == 'overlap'))
                assert eng._async == (partitioned == 'overlap')
                rep = eng.halo_report()
                assert rep['recv_bytes_per_step'] < rep['allgather_bytes_per_step'] or world == 1
            else:
                eng = NodeembEngine(*args, loss_type=lt, alpha=0.5, lr=1e-2)
            for _ in range(6):
                eng.step()
            torch.cuda.synchronize()
            results.append((m.deletion1.deletion_weight.detach().cpu(), m.deletion2.deletion_weight.detach().cpu(),
                            eng.loss_history()))
        (a1, a2, ah), (b1, b2, bh) = results[:2]
        if len(results) == 3:
            c1, c2, ch = results[2]
            same_or_rounding(f'{cls.__name__} {lt}', ((b1, c1), (b2, c2), (bh, ch)))
        err = max(float((a1 - b1).norm() / a1.norm()), float((a2 - b2).norm() / a2.norm()))
        assert err < 1e-4, (cls.__name__, lt, err)
        assert torch.allclose(ah, bh, rtol=1e-4, equal_nan=True), (cls.__name__, lt, ah, bh)
        if rank == 0:
            print(f'{cls.__name__} {lt}: partitioned == single (rel err {err:.2e})', flush=True)


def same_or_rounding(tag, pairs):
    """The overlapped program runs the same kernels on the same operands in another order: identical Del weights AND loss logs,
    bit for bit.  (Round 4: the logs of two engines could differ by one ulp - the constants of the log were summed with
    index_add_'s atomics at set-up; they are summed in a fixed order now, engine._LayerTerms.)"""
    for a, b in pairs:
        assert torch.equal(a.nan_to_num(), b.nan_to_num()), tag


def rgcn_checks(rank, world):
    """R-GCN (block-diagonal relation weights, 50 relation types): the row-partitioned engine with typed halos against the
    single-GPU fused engine on the same knowledge-graph request."""
    from types import SimpleNamespace
    import bench
    from gnndelete_amd.dist_engine import PartitionedNodeembEngine
    dev = torch.device('cuda', 0)
    for lt in ('both_layerwise', 'both_all', 'only2_all'):
        results = []
        for partitioned in (False, True, 'overlap'):
            args = SimpleNamespace(gnn='rgcn', workload='synth-kg-small', seed=42, df='in', df_size=2.5, loss_type=lt, no_graph=False)
            data, model, neg, ni1, ni2 = bench.build_kg_request(args)
            os.environ['GD_DIST_OVERLAP'] = '1' if partitioned == 'overlap' else '0'       # (the engine's default switch)
            eng = bench.make_kg_engine(args, data, model, neg, ni1, ni2, dev, rank, world, None, partition=bool(partitioned))
            assert isinstance(eng, PartitionedNodeembEngine) == bool(partitioned)
            assert not partitioned or eng._async == (partitioned == 'overlap')
            for _ in range(4):
                eng.step()
            torch.cuda.synchronize()
            results.append((model.deletion1.deletion_weight.detach().cpu(), model.deletion2.deletion_weight.detach().cpu(),
                            eng.loss_history()))
        (a1, a2, ah), (b1, b2, bh), (c1, c2, ch) = results
        same_or_rounding(f'RGCNDelete {lt}', ((b1, c1), (b2, c2), (bh, ch)))
        err = max(float((a1 - b1).norm() / a1.norm()), float((a2 - b2).norm() / a2.norm()))
        assert err < 1e-4, (lt, err)
        assert torch.allclose(ah, bh, rtol=1e-4, equal_nan=True), (lt, ah, bh)
        if rank == 0:
            print(f'RGCNDelete {lt}: partitioned == single (rel err {err:.2e})', flush=True)


def rgcn_cpu_checks(rank, world):
    """The TYPED row partition at any world size, on CPU (the R-GCN partition had only ever seen two
    ranks).  Everything PartitionedNodeembEngine builds for mode 'rgcn' - this rank's TypedNodeCSR(row_range) (in-edges of the
    own target rows forward, out-edges of the own source rows with the GLOBAL mean weights backward) and the two halo lists
    derived from the untyped union graph - drives a dense emulation of conv2's forward on the own target rows and of its input
    gradient on the own source rows, reading ONLY own + received rows (everything else is NaN); both against the oracle's
    RGCNConv with autograd on the whole graph.  The ranks' typed edge sets must tile the global ones exactly."""
    from gnndelete_amd.collectives import all_reduce_sum, exchange_rows, halo_plan, row_blocks
    from gnndelete_amd.graph import TypedNodeCSR, build_csr
    from oracle import pyg_semantics as pyg
    torch.manual_seed(5)
    n, r, e, d_in, d_out, nb = 331, 10, 2600, 16, 8, 2          # (n not a multiple of 8: the last rank's block is short)
    src, dst = torch.randint(0, n, (e,)), torch.randint(0, n, (e,))
    keep = src != dst
    ei = torch.unique(torch.stack([src[keep], dst[keep], torch.randint(0, r, (int(keep.sum()),))]), dim=1)
    ei, et = ei[:2].contiguous(), ei[2].contiguous()
    x = torch.randn(n, d_in, dtype=torch.float64, requires_grad=True)
    weight = torch.randn(r, nb, d_in // nb, d_out // nb, dtype=torch.float64) * 0.3
    root, bias = torch.randn(d_in, d_out, dtype=torch.float64) * 0.3, torch.randn(d_out, dtype=torch.float64)
    want = pyg.rgcn_conv(x, ei, et, weight, root, bias, num_blocks=nb)
    dy = torch.randn(n, d_out, dtype=torch.float64)
    want_dx, = torch.autograd.grad(want, x, dy)
    chunk, _ = row_blocks(n, world)
    lo, hi = min(n, rank * chunk), min(n, (rank + 1) * chunk)
    whole = TypedNodeCSR(ei, et, n, r)
    mine = TypedNodeCSR(ei, et, n, r, row_range=(lo, hi))
    # ---- tiling: this rank's forward / backward edges are exactly the whole graph's edges into / out of its rows
    cnt = torch.tensor([mine.fwd[3].numel(), mine.bwd[3].numel()], dtype=torch.float32)
    all_reduce_sum(cnt, world)
    assert cnt.tolist() == [whole.fwd[3].numel(), whole.bwd[3].numel()], cnt
    g = build_csr(ei, n, 'sum')                                   # the untyped union graph: structure of the halo lists
    halo_f = halo_plan(g.rowptr, g.col, n, rank, world, chunk)
    halo_b = halo_plan(g.rowptr_t, g.col_t, n, rank, world, chunk)
    wblk = torch.zeros(r, d_in, d_out, dtype=torch.float64)        # block-diagonal relation weights, dense
    bi, bo = d_in // nb, d_out // nb
    for b in range(nb):
        wblk[:, b * bi:(b + 1) * bi, b * bo:(b + 1) * bo] = weight[:, b]

    def runs(side):
        node_ptr, seg_ptr, seg_rel, col, w = side
        seg_len = (seg_ptr[1:] - seg_ptr[:-1]).long()
        node_of_run = torch.repeat_interleave(torch.arange(n), (node_ptr[1:] - node_ptr[:-1]).long())
        return (torch.repeat_interleave(node_of_run, seg_len), torch.repeat_interleave(seg_rel.long(), seg_len), col.long(), w.double())

    def exchange(buf, plan):
        send = buf.index_select(0, plan.send_rows).float()
        assert not torch.isnan(send).any()
        recv = torch.empty(max(1, plan.n_recv), buf.shape[1])
        exchange_rows(send, recv, plan, world)
        return recv[:plan.n_recv].double()
    # ---- forward on the own target rows: gathers own + received source rows
    xd = x.detach()
    xin = torch.full((n, d_in), float('nan'), dtype=torch.float64)
    xin[lo:hi] = xd[lo:hi]
    xin.index_copy_(0, halo_f.recv_rows, exchange(xin, halo_f))
    tgt, rel, col, w = runs(mine.fwd)
    assert bool(((tgt >= lo) & (tgt < hi)).all())
    msg = torch.einsum('ei,eio->eo', xin[col] * w[:, None], wblk[rel])
    out = (xd[lo:hi] @ root + bias).index_add_(0, tgt - lo, msg)
    assert not torch.isnan(out).any()
    assert torch.allclose(out.float(), want.detach()[lo:hi].float(), rtol=2e-5, atol=2e-6), float((out - want.detach()[lo:hi]).abs().max())
    # ---- input gradient on the own source rows: gathers own + received rows of dy (transposed graph, global mean weights)
    dyin = torch.full((n, d_out), float('nan'), dtype=torch.float64)
    dyin[lo:hi] = dy[lo:hi]
    dyin.index_copy_(0, halo_b.recv_rows, exchange(dyin, halo_b))
    s_node, rel, col, w = runs(mine.bwd)                          # node = the SOURCE row, col = the target whose dy is gathered
    assert bool(((s_node >= lo) & (s_node < hi)).all())
    gmsg = torch.einsum('eo,eio->ei', dyin[col] * w[:, None], wblk[rel])
    dx = (dy[lo:hi] @ root.t()).index_add_(0, s_node - lo, gmsg)
    assert not torch.isnan(dx).any()
    assert torch.allclose(dx.float(), want_dx[lo:hi].float(), rtol=2e-5, atol=2e-6), float((dx - want_dx[lo:hi]).abs().max())
    if rank == 0:
        print(f'typed partition x{world}: forward and input gradient of the own rows == whole-graph oracle', flush=True)


def main():
    mode = sys.argv[1]
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    try:
        if mode in ('rccl1', 'direct1'):
            assert world == 1 and os.environ.get('GD_FORCE_COLLECTIVES') == '1'
            gpu_checks(rank, world, rccl=True, direct=mode == 'direct1')
        elif mode == 'rgcn':
            rgcn_checks(rank, world)
        elif mode == 'rgcn_cpu':
            rgcn_cpu_checks(rank, world)
        else:
            (cpu_checks if mode == 'cpu' else gpu_checks)(rank, world)
        dist.barrier()
        if rank == 0:
            print('DIST_OK', flush=True)
    finally:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
