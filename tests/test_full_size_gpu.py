"""Parity at BASELINE.json's FULL sizes (north_star: affected-node embeddings within 1e-4 rel-L2, AUC within
+-0.002 on identical seeds): the HIP engine and the CPU oracle run the same few Del-training iterations of the same
request from the same state with the same negatives.  What these tests see that the fixture-sized ones cannot: hub
rows split over several work items, all 8 XCD ranges, the label-propagation node order (engine: n > 4096), the
wide-K first layer of the bag-of-words graphs (F = 1,639), the typed conv kernel at 51 relation types / 9.5 M
typed edges with block-diagonal weights.

  config 1  synth-cora    GCN        0.5 % OUT   (F = 8,710: the K-tiled first-layer product streams W1 through LDS)
  config 2  synth-dblp    GCN        2.5 % OUT   (delete_gnn.py --dataset DBLP --gnn gcn --df out --df_size 2.5)
  config 3  synth-collab  GraphSAGE  5 % IN      (BASELINE names GraphSAGE; no reference model: oracle = own restatement)
  config 5' synth-collab  GAT        5 % IN
  config 4  synth-biokg   R-GCN      forward + Del-weight gradients at R = 51 (num_blocks = 4); the fused engine for 10 iterations
  config 5  synth-collab  GAT        node deletion (delete_node.py), 10 Del epochs through the node-classification trainer
The GCN / collab case of config 3 is bench.py's own `post_delete_auc` leg."""
from types import SimpleNamespace

import pytest
import torch

from helpers import oracle_runner, rel_l2

pytestmark = pytest.mark.gpu

ITERS = {'gcn': 20, 'sage': 10, 'gat': 10}     # Del iterations before the 1e-4 embedding bar of north_star is asserted (GCN = the metric's model)


def _request(workload, gnn, df, df_size, seed=42):
    import bench
    args = SimpleNamespace(workload=workload, gnn=gnn, df=df, df_size=df_size, seed=seed)
    return bench.build_request(args, torch.device('cuda'))


def _auc(z, pos, neg):
    from gnndelete_amd.framework.metrics import batched_roc_auc
    ei = torch.cat([pos, neg], 1).to(z.device)
    score = (z[ei[0]] * z[ei[1]]).sum(-1).sigmoid()
    label = torch.cat([torch.ones(pos.shape[1]), torch.zeros(neg.shape[1])]).to(z.device)
    return float(batched_roc_auc(score, label)[0])


def _assert_del_weights_within_fp32_spread(tag, hip_w, w64, ens, iters):
    """The Del WEIGHTS are a looser observable than the embeddings north_star bounds: Adam's first updates are
    lr * m / sqrt(v) ~ +-lr per entry whatever the gradient's size, so the fp32 summation-order noise of a 180k-row
    weight-gradient reduction shows up undamped in the weight, while the embeddings see it scaled by lr.  So they are held
    to what fp32 arithmetic itself can deliver: HIP's distance to the fp64 oracle's weights <= 2 x the largest distance of
    an fp32 ENSEMBLE (the same oracle in fp32 with several scatter orders; for the CPU-sized cases also the CPU oracle) to them
    (+ 5e-5: the ensemble's own spread from run to run is a factor of four at this horizon - 3.5e-6 ... 1.4e-5 for W_D1 of
    GCN at collab size - while HIP sits at 1.5e-5 every time; 5e-5 is 20 x below the 1e-3 this assertion replaced)."""
    for k, name in enumerate(('W_D1', 'W_D2')):
        d_ens = [rel_l2(e[k], w64[k]) for e in ens]
        d_hip = rel_l2(hip_w[k], w64[k])
        print(f'[{tag}] {name} after {iters} iterations, rel-L2 to the fp64 oracle: fp32 ensemble '
              + ' '.join(f'{v:.2e}' for v in d_ens) + f' / HIP {d_hip:.2e}')
        assert d_hip <= max(2.0 * max(d_ens), 5e-5), (name, d_hip, d_ens)


@pytest.mark.parametrize('workload,gnn,df,df_size', [('synth-cora', 'gcn', 'out', 0.5), ('synth-dblp', 'gcn', 'out', 2.5),
                                                     ('synth-collab', 'gcn', 'in', 5.0),       # the metric's own model
                                                     ('synth-collab', 'sage', 'in', 5.0), ('synth-collab', 'gat', 'in', 5.0)])
def test_full_size_training_parity(workload, gnn, df, df_size):
    import oracle_jobs
    from gnndelete_amd.engine import NodeembEngine
    iters = ITERS[gnn]
    from oracle import gnndelete_ref as R
    # the CPU oracle's iterations at collab size (30 s of host time for GCN) run in a child process from the same seeded request
    # (tests/oracle_jobs.py; started at session start when the whole suite runs) while this process uses the GPU
    job = f'full-collab-{gnn}' if workload == 'synth-collab' else None
    if job:
        assert oracle_jobs.JOBS[job][1] == dict(workload=workload, gnn=gnn, df=df, df_size=df_size, iters=iters)
        oracle_jobs.start(job)
    data, model, neg, ni1, ni2 = _request(workload, gnn, df, df_size)
    state = {k: v.clone() for k, v in model.state_dict().items()}
    torch.set_num_threads(min(32, torch.get_num_threads()))
    ref = R.TwoLayerDelete(gnn, data.x.shape[1], 128, 64, data.sdf_node_1hop_mask, data.sdf_node_2hop_mask)
    ref.load_state_dict(state, strict=False)
    E = data.train_pos_edge_index
    e_dr, e_sdf, pos = E[:, data.dr_mask], E[:, data.sdf_mask], E[:, data.df_mask]
    with torch.no_grad():
        z1o, z2o = ref.get_original_embeddings(data.x, e_dr, return_all_emb=True)
    if not job:
        targets = dict(z1_ori=z1o, z2_ori=z2o, pos_edge=pos, neg_edge=neg, ni_mask1=ni1, ni_mask2=ni2)
        opt = R.make_optimizer(ref, 'both_layerwise', 1e-3)
        logs = [R.nodeemb_epoch(ref, lambda: ref(data.x, e_sdf, return_all_emb=True), targets, opt, 'both_layerwise', 0.5,
                                R.LOSSES['mse_mean']) for _ in range(iters)]
    dev = torch.device('cuda')
    hip = model.to(dev)
    eng = NodeembEngine(hip, data.x.to(dev), e_sdf.to(dev).contiguous(), z1o.to(dev), z2o.to(dev), pos.to(dev), neg.to(dev),
                        ni1, ni2, loss_type='both_layerwise', alpha=0.5, lr=1e-3)
    assert eng.perm is not None, 'full-size requests run in the locality order'
    assert eng.graph.plan.n_split > 0 or workload == 'synth-cora', 'hub rows are split over several work items at this size'
    if workload == 'synth-collab':
        # the collab-sized step runs the one-pass Del-1 form (round 5); its compact partial counts must not shrink the buffers the
        # two-launch kernels write when bench.py times them on this engine (a 512-block loss-partial write into a 256-block buffer
        # ran over the tail launch's check-in counter: a hang under the profiler, a memory fault in the GAT line)
        from gnndelete_amd import _lib
        assert eng._fuse_del1 and eng._fuse_wg2
        assert eng._chain1 == (gnn in ('gcn', 'gat', 'gin')), 'GCN / GIN / GAT form the previous input gradient inside the Del-1 pass'
        assert eng._lp1.numel() >= 2 * _lib.lib().gd_rows_gemm_wgrad_blocks(eng.s1)
        assert eng._lp2.numel() >= 2 * _lib.lib().gd_rows_gemm_wgrad_blocks(eng.s2)
    for _ in range(iters):
        eng.step()
    hist = eng.loss_history()
    wts = lambda m_: (m_.deletion1.deletion_weight.detach().double().cpu(), m_.deletion2.deletion_weight.detach().double().cpu())
    if workload == 'synth-collab':
        # (round 6) the same iterations restricted to the rows the request can influence - `--affected_rows_only`, now CHAINED like
        # the full step and with the weight-stationary row GEMMs on (index list + selector) operands: the same sums over the same
        # affected rows, so the Del weights agree with the full step's far inside the parity bound
        import copy
        hip_r = copy.deepcopy(hip)
        hip_r.load_state_dict(state)
        eng_r = NodeembEngine(hip_r, data.x.to(dev), e_sdf.to(dev).contiguous(), z1o.to(dev), z2o.to(dev), pos.to(dev), neg.to(dev),
                              ni1, ni2, loss_type='both_layerwise', alpha=0.5, lr=1e-3, affected_rows_only=True)
        assert eng_r._rows_only and eng_r.s2 < eng_r.n and eng_r._chain1 == eng._chain1
        for _ in range(iters):
            eng_r.step()
        assert torch.allclose(eng_r.loss_history(), hist, rtol=1e-5, atol=0)
        for a_, b_ in zip(wts(hip_r), wts(hip)):
            assert rel_l2(a_, b_) < 2e-5, rel_l2(a_, b_)
        del eng_r, hip_r
    if job:                                # the CPU oracle's result: its log, and its final Del weights into `ref`
        res = oracle_jobs.result(job)
        assert res['checksum'] == oracle_jobs.checksum(state, neg), 'the child built another request than this process'
        logs = res['logs']
        with torch.no_grad():
            ref.deletion1.deletion_weight.copy_(res['w1'])
            ref.deletion2.deletion_weight.copy_(res['w2'])
    for i, log in enumerate(logs):
        assert abs(float(hist[i, 0]) - log['train_loss']) <= 1e-4 * abs(log['train_loss']), (i, float(hist[i, 0]), log)
    def run_oracle(dtype, perm):           # one oracle at a time (their autograd tapes at collab size are tens of GB)
        import gc
        step, snap, _ = oracle_runner(gnn, data, state, neg, ni1, ni2, dtype, dev, perm=perm)
        for _ in range(iters):
            step()
        w = snap()[:2]
        del step, snap
        gc.collect()
        torch.cuda.empty_cache()
        return w
    w64 = run_oracle(torch.float64, None)
    ens = [wts(ref)] + [run_oracle(torch.float32, p) for p in (None, 1, 2)]
    _assert_del_weights_within_fp32_spread(f'{workload} {gnn}', wts(hip), w64, ens, iters)
    with torch.no_grad():
        r1, r2 = ref(data.x, e_dr, return_all_emb=True)
        h1, h2 = hip(data.x.to(dev), e_dr.to(dev).contiguous(), return_all_emb=True)
    m1, m2 = data.sdf_node_1hop_mask, data.sdf_node_2hop_mask
    assert rel_l2(h1.cpu()[m1], r1[m1]) < 1e-4
    assert rel_l2(h2.cpu()[m2], r2[m2]) < 1e-4
    a_hip = _auc(h2, data.test_pos_edge_index, data.test_neg_edge_index)
    a_cpu = _auc(r2, data.test_pos_edge_index, data.test_neg_edge_index)
    assert abs(a_hip - a_cpu) < 2e-3


def test_full_size_rgcn_forward_and_del_gradients():
    """Config 4 shape: N = 93,773 entities, 51 relation types (102 with the reverse direction, delete_gnn.py:158-164)
    -> RGCNConv's block-diagonal branch (rgcn.py:17-22), ~8.6 M typed edges, embedding 128 -> 128 -> 64.  One
    full-graph forward with Del operators on random S_Df-like masks and the gradients of a quadratic loss w.r.t.
    both Del weights, HIP (typed conv kernel, csrc/rgcn.hip) vs the CPU oracle."""
    from gnndelete_amd.framework.models import RGCNDelete
    from gnndelete_amd.framework.synth import make_kg_dataset
    from oracle import gnndelete_ref as R
    data, _ = make_kg_dataset('synth-biokg', seed=42)
    n, nr = data.num_nodes, 51
    E, et = data.train_pos_edge_index, data.train_edge_type
    ei, ety = torch.cat([E, E.flip(0)], 1), torch.cat([et, et + nr])
    g = torch.Generator().manual_seed(5)
    m1, m2 = torch.rand(n, generator=g) < 0.3, torch.rand(n, generator=g) < 0.6
    torch.manual_seed(11)
    hip = RGCNDelete(SimpleNamespace(in_dim=128, hidden_dim=128, out_dim=64), n, nr, m1, m2)
    with torch.no_grad():
        for name, p in hip.named_parameters():
            if 'deletion_weight' in name:
                p.copy_(torch.eye(p.shape[0]) * 0.5 + torch.randn_like(p) * 0.05)
    assert hip.conv1.num_blocks == 4
    ref = R.TwoLayerDelete('rgcn', 128, 128, 64, m1, m2, num_nodes=n, num_edge_type=nr)
    res = ref.load_state_dict(hip.state_dict(), strict=False)
    assert not res.missing_keys and not res.unexpected_keys, res
    torch.set_num_threads(min(32, torch.get_num_threads()))

    def loss_of(z1, z2, a, b):
        return (z1[a] ** 2).mean() + (z2[b] ** 2).mean()
    # (inputs of Del-1 and dL/dz1 of both implementations, for the gate-masked comparison of the W_D1 gradient below)
    kept = {}
    ref.deletion1.register_forward_hook(lambda mod, inp, out: kept.__setitem__('p_ref', inp[0].detach()))
    hip.deletion1.register_forward_hook(lambda mod, inp, out: kept.__setitem__('p_hip', inp[0].detach()))
    r1, r2 = ref(data.x, ei, ety, return_all_emb=True)
    r1.retain_grad()
    loss_of(r1, r2, m1, m2).backward()
    hip = hip.cuda()
    h1, h2 = hip(data.x.cuda(), ei.cuda(), ety.cuda(), return_all_emb=True)
    h1.retain_grad()
    loss_of(h1, h2, m1.cuda(), m2.cuda()).backward()
    assert rel_l2(h1.detach().cpu()[m1], r1.detach()[m1]) < 1e-4
    assert rel_l2(h2.detach().cpu()[m2], r2.detach()[m2]) < 1e-4
    assert rel_l2(hip.deletion2.deletion_weight.grad.cpu(), ref.deletion2.deletion_weight.grad) < 1e-4
    # W_D1's gradient passes the ReLU between the layers: of the 12 M entries of z1 a handful lie within fp32 rounding
    # of zero, and two correct implementations gate such an entry differently (tools/experiments/diag_rgcn_row.py: ONE
    # row of dL/dz1 off by its gated term, every other row equal to 3e-9).  So the comparison is
    # GATE-MASKED instead of loosened: the rows of z1 whose sign pattern differs between the two
    # implementations - a handful, counted - are taken out of BOTH gradients (dW_D1 = sum over the Del-1 rows of
    # p1[i]^T dL/dz1[i]); what is left must agree to the bound every other quantity here is held to.
    hz, rz = h1.detach().cpu(), r1.detach()
    flip = ((hz > 0) != (rz > 0)).any(1) & m1
    n_flip = int(flip.sum())
    print(f'[synth-biokg rgcn forward] Del-1 rows whose ReLU sign pattern differs between HIP and the CPU oracle: {n_flip} of {int(m1.sum())}')
    assert n_flip <= 16, n_flip
    g_hip = hip.deletion1.deletion_weight.grad.double().cpu() - kept['p_hip'].double().cpu()[flip].T @ h1.grad.double().cpu()[flip]
    g_ref = ref.deletion1.deletion_weight.grad.double() - kept['p_ref'].double()[flip].T @ r1.grad.double()[flip]
    assert rel_l2(g_hip, g_ref) < 1e-4, rel_l2(g_hip, g_ref)
    # DistMult scores of the validation triples on the unlearned embeddings
    s_hip = hip.decode(h2, data.val_pos_edge_index.cuda(), data.val_edge_type.cuda())
    s_ref = ref.decode(r2, data.val_pos_edge_index, data.val_edge_type)
    assert rel_l2(s_hip.detach().cpu(), s_ref.detach()) < 1e-4


def test_full_size_rgcn_fused_engine_matches_oracle():
    """Config 4 through the path bench.py --gnn rgcn and delete_gnn.py --fullgraph run: NodeembEngine(mode 'rgcn') itself
    - the fused, hipGraph-captured R-GCN Del step whose three typed launches are the wave-private kernel
    (csrc/rgcn_wave.hip: one wave per (64-node tile, diagonal block), the reference's num_blocks = 4, rgcn.py:17-22) - on
    the whole synth-biokg request (93,773 entities, 102 relation types, ~8.4 M typed Dr edges, 2.5 % IN triple deletion;
    gnndelete_nodeemb.py:745-800) for TEN iterations from the same state with the same head-shuffled negatives.
    The oracle runs as torch ops on the GPU (the CPU oracle needs ~20 s per R-GCN iteration): an fp64 run is the yardstick
    - per-iteration losses within 1e-4 - and the Del weights and the affected-node embeddings (north_star: 1e-4 rel-L2) are
    held to the spread of an fp32 ensemble of the same oracle (three scatter orders) around it."""
    import gc
    import bench
    from gnndelete_amd import ops
    args = SimpleNamespace(workload='synth-biokg', gnn='rgcn', df='in', df_size=2.5, seed=42, loss_type='both_layerwise',
                           no_graph=False, cpu_baseline_iters=2)
    dev = torch.device('cuda')
    data, model, neg, ni1, ni2 = bench.build_kg_request(args)
    state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    eng = bench.make_kg_engine(args, data, model, neg, ni1, ni2, dev)
    assert eng._mode == 'rgcn' and eng._graph is None
    assert ops.rgcn_wave_form(128, 128, 4) and ops.rgcn_wave_form(128, 64, 4) and ops.rgcn_wave_form(64, 128, 4), \
        'the three typed launches of this request run on the wave-private kernel'
    iters = 10
    for _ in range(iters):
        eng.step()
    assert eng._graph is not None, 'the step is replayed from a hipGraph'
    hist = eng.loss_history()
    assert bool(torch.isfinite(hist).all()) and hist.shape[0] == iters
    ei, et = data.edge_index[:, data.dr_mask], data.edge_type[data.dr_mask]
    with torch.no_grad():
        h1, h2 = model(data.x.to(dev), ei.to(dev).contiguous(), et.to(dev).contiguous(), return_all_emb=True)
    h1, h2 = h1[ni1.to(dev)].double().cpu(), h2[ni2.to(dev)].double().cpu()
    hip_w = (model.deletion1.deletion_weight.detach().double().cpu(), model.deletion2.deletion_weight.detach().double().cpu())

    def run_oracle(dtype, perm):           # one oracle at a time: 102 relations x [N, 128] on the tape of each
        step, snap, _ = oracle_runner('rgcn', data, state, neg, ni1, ni2, dtype, dev, perm=perm, edges=data.edge_index,
                                      edge_type=data.edge_type, pos=data.kg_dec_edge, num_edge_type=data.kg_num_edge_type,
                                      del_masks=(ni1, ni2), train_mask=data.dr_mask)
        logs = [step() for _ in range(iters)]
        out = snap()[:4]
        del step, snap
        gc.collect()
        torch.cuda.empty_cache()
        return logs, out
    # the epoch as delete_gnn.py --fullgraph runs it by default (conv1's output computed once, conv2's input gradient only on the
    # Del-1 rows): the same ten iterations from the same state, held to the same bounds below
    model.load_state_dict(state)
    eng_t = bench.make_kg_engine(args, data, model, neg, ni1, ni2, dev, cache_layer1=True, affected_rows_only=True)
    assert eng_t.typed_s1 is not None and eng_t.cache_layer1
    for _ in range(iters):
        eng_t.step()
    hist_t = eng_t.loss_history()
    trainer_w = (model.deletion1.deletion_weight.detach().double().cpu(), model.deletion2.deletion_weight.detach().double().cpu())
    logs64, (w1, w2, r1, r2) = run_oracle(torch.float64, None)
    for i, log in enumerate(logs64):
        assert abs(float(hist[i, 0]) - log['train_loss']) <= 1e-4 * abs(log['train_loss']), (i, float(hist[i, 0]), log)
        assert abs(float(hist_t[i, 0]) - log['train_loss']) <= 1e-4 * abs(log['train_loss']), ('trainer defaults', i, float(hist_t[i, 0]), log)
    ens = [run_oracle(torch.float32, p)[1] for p in (None, 1, 2)]
    _assert_del_weights_within_fp32_spread('synth-biokg rgcn', hip_w, (w1, w2), [e[:2] for e in ens], iters)
    _assert_del_weights_within_fp32_spread('synth-biokg rgcn, trainer defaults', trainer_w, (w1, w2), [e[:2] for e in ens], iters)
    # affected-node embeddings: north_star's 1e-4 wherever fp32 arithmetic delivers it - after ten both_layerwise iterations of
    # THIS request a correct fp32 implementation is itself ~2e-4 from the fp64 run in z1 (the ReLU between the layers gates the
    # layer-2 gradient with [z1 > 0]), so the bound is the larger of 1e-4 and twice the ensemble's distance
    for name, h, r, k in (('z1[S1]', h1, r1, 2), ('z2[S2]', h2, r2, 3)):
        d_ens, d_hip = [rel_l2(e[k], r) for e in ens], rel_l2(h, r)
        print(f'[synth-biokg rgcn] {name} after {iters} iterations, rel-L2 to the fp64 oracle: fp32 ensemble '
              + ' '.join(f'{v:.2e}' for v in d_ens) + f' / HIP {d_hip:.2e}')
        #  north_star's 1e-4 is asserted AS IS whenever the fp32 ensemble itself is inside it; only where correct fp32
        # implementations are themselves outside (said so loudly below) is HIP held to twice their distance instead
        if max(d_ens) <= 1e-4:
            assert d_hip <= 1e-4, (name, d_hip, d_ens)
        else:
            import warnings
            warnings.warn(f'[synth-biokg rgcn] {name}: the fp32 ENSEMBLE is {max(d_ens):.2e} from the fp64 oracle after {iters} iterations - '
                          f'outside north_star\'s 1e-4 bound on its own; HIP ({d_hip:.2e}) is held to twice that distance')
            assert d_hip <= 2.0 * max(d_ens), (name, d_hip, d_ens)


def test_full_size_node_deletion_gat_matches_oracle(tmp_path, monkeypatch):
    """BASELINE config 5 at the size it names: delete_node.py's request (5 % of the NODES deleted with every edge touching
    them, S_Df on the undirected edge_index, delete_node.py:77-142) on the ogbl-collab-shaped node-classification stand-in
    (235,868 nodes, 4 classes), GAT.  The HIP node-classification trainer (GNNDeleteNodeClassificationTrainer ->
    fused engine; out_dim = 4: the engine pads layer 2 to 64 zero columns, engine._padded_out_shadow) against the oracle's
    restatement of the same loop (gnndelete_nodeemb.py:498-657) from the same state with the same negatives: per-epoch
    losses, Del weights, affected-node embeddings, test accuracy.  The epoch time of the HIP path is printed."""
    import time
    import oracle_jobs
    from gnndelete_amd.framework.data import Data
    from gnndelete_amd.framework.trainer import gnndelete_nodeemb as TN
    from oracle import gnndelete_ref as R
    epochs, lr, alpha = 10, 1e-2, 0.5
    # ---- oracle (CPU): its ten epochs run in a child process from the same seeded request (tests/oracle_jobs.py) while this
    # process drives the GPU
    assert oracle_jobs.JOBS['full-nodecls-gat'][1] == dict(epochs=epochs, lr=lr, alpha=alpha)
    oracle_jobs.start('full-nodecls-gat')
    data, hip, state, neg = oracle_jobs.nodecls_request()
    n, E = data.num_nodes, data.edge_index
    s1, s2 = data.sdf_node_1hop_mask, data.sdf_node_2hop_mask
    torch.set_num_threads(min(32, torch.get_num_threads()))
    ref = R.TwoLayerDelete('gat', data.x.shape[1], 128, data.num_classes, s1, s2)
    ref.load_state_dict(state, strict=False)
    d = {k: v for k, v in data.items()}
    d['train_pos_edge_index'] = E
    # ---- HIP trainer
    monkeypatch.setattr(TN, 'negative_sampling', lambda *a, **k: neg.cuda())
    args = SimpleNamespace(unlearning_model='gnndelete_nodeemb', dataset='synth-collab', checkpoint_dir=str(tmp_path),
                           eval_on_cpu=False, epochs=epochs, valid_freq=epochs, lr=lr, alpha=alpha, loss_fct='mse_mean',
                           loss_type='both_layerwise', gnn='gat')
    opt = [torch.optim.Adam(hip.deletion1.parameters(), lr=lr), torch.optim.Adam(hip.deletion2.parameters(), lr=lr)]
    tr = TN.GNNDeleteNodeClassificationTrainer(args)
    t0 = time.time()
    tr.train(hip, Data(d), opt, args)
    torch.cuda.synchronize()
    print(f'config 5 at collab size: {epochs} Del epochs + validation + checkpoints {time.time() - t0:.2f} s; '
          f'train_time per epoch {tr.trainer_log["log"][0].get("train_time", float("nan")) * 1e3:.2f} ms')
    hist = torch.tensor(tr.trainer_log['loss_history'])
    # Del weights: HIP's distance to an fp64 run of the same loop against the spread of an fp32 ensemble (the CPU oracle above +
    # the same oracle as torch ops on the GPU with three scatter orders), as in test_full_size_training_parity
    import gc
    dev = torch.device('cuda')
    ni1, ni2 = R.non_df_masks(n, data.directed_df_edge_index, s1, s2)
    wts = lambda m_: (m_.deletion1.deletion_weight.detach().double().cpu(), m_.deletion2.deletion_weight.detach().double().cpu())

    def run_oracle(dtype, perm):
        step, snap, _ = oracle_runner('gat', data, state, neg, ni1, ni2, dtype, dev, lr=lr, alpha=alpha, perm=perm, edges=E,
                                      out=data.num_classes)
        ls = [step() for _ in range(epochs)]
        w = snap()[:2]
        del step, snap
        gc.collect()
        torch.cuda.empty_cache()
        return ls, w
    logs64, w64 = run_oracle(torch.float64, None)
    for i, log in enumerate(logs64):
        assert abs(float(hist[i, 0]) - log['train_loss']) <= 1e-4 * abs(log['train_loss']), (i, float(hist[i, 0]), log)
    gpu_members = [run_oracle(torch.float32, p)[1] for p in (None, 1, 2)]
    res = oracle_jobs.result('full-nodecls-gat')                    # the CPU oracle's epochs: log + final Del weights into `ref`
    assert res['checksum'] == oracle_jobs.checksum(state, neg), 'the child built another request than this process'
    logs = res['logs']
    with torch.no_grad():
        ref.deletion1.deletion_weight.copy_(res['w1'])
        ref.deletion2.deletion_weight.copy_(res['w2'])
    for i, log in enumerate(logs):
        assert abs(float(hist[i, 0]) - log['train_loss']) <= 1e-4 * abs(log['train_loss']), (i, float(hist[i, 0]), log)
    ens = [wts(ref)] + gpu_members
    _assert_del_weights_within_fp32_spread('synth-collab node deletion gat', wts(hip), w64, ens, epochs)
    e_dr = E[:, data.dr_mask]
    with torch.no_grad():
        r1, r2 = ref(data.x, e_dr, return_all_emb=True)
        h1, h2 = hip(data.x.to(dev), e_dr.to(dev).contiguous(), return_all_emb=True)
    assert rel_l2(h1.cpu()[s1], r1[s1]) < 1e-4 and rel_l2(h2.cpu()[s2], r2[s2]) < 1e-4
    acc_ref = float((r2.argmax(1)[data.test_mask] == data.y[data.test_mask]).float().mean())
    acc_hip = float((h2.cpu().argmax(1)[data.test_mask] == data.y[data.test_mask]).float().mean())
    assert abs(acc_ref - acc_hip) <= 2e-3, (acc_ref, acc_hip)
