"""HIP-backed models (gnndelete_amd.framework.models) vs the golden vectors generated from the
reference and vs the CPU oracle: embeddings, decoder scores, Del-weight gradients, and whole
training trajectories for every loss_type.  Tolerance 1e-4 rel-L2 (north_star), met with margin."""
import numpy as np
import pytest
import torch

from helpers import hip_model, load_golden, oracle_model, rel_l2, split_fixture, t

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.mark.parametrize('gnn', ['gcn', 'gat', 'gin'])
def test_delete_models_match_reference_golden(gnn):
    fx = load_golden(f'wiring_{gnn}.npz')
    state, _, rest = split_fixture(fx)
    m = hip_model(gnn, state, t(rest['mask1']), t(rest['mask2']))
    x, ei = t(rest['x']).cuda(), t(rest['edge_index']).cuda()
    z1, z2 = m(x, ei, return_all_emb=True)
    o1, o2 = m.get_original_embeddings(x, ei, return_all_emb=True)
    for got, key in [(z1, 'z1'), (z2, 'z2'), (o1, 'o1'), (o2, 'o2')]:
        assert rel_l2(got.detach().cpu(), rest[key]) < TOL, key
    a1, a2 = m(x, ei, mask_1hop=t(rest['alt1']), mask_2hop=t(rest['alt2']), return_all_emb=True)
    assert rel_l2(a1.detach().cpu(), rest['a1']) < TOL and rel_l2(a2.detach().cpu(), rest['a2']) < TOL
    s = m.decode(z2, t(rest['val_pos']).cuda(), t(rest['val_neg']).cuda())
    assert rel_l2(s.detach().cpu(), rest['score']) < TOL
    assert m(x, ei).shape == z2.shape


def test_rgcn_delete_matches_reference_golden():
    fx = load_golden('wiring_rgcn.npz')
    state, _, rest = split_fixture(fx)
    r = int(rest['num_edge_type'])
    m = hip_model('rgcn', state, t(rest['mask1']), t(rest['mask2']), num_nodes=state['node_emb.weight'].shape[0],
                  num_edge_type=r)
    x, ei, et = t(rest['x']).cuda(), t(rest['edge_index']).cuda(), t(rest['edge_type']).cuda()
    z1, z2 = m(x, ei, et, return_all_emb=True)
    o1, o2 = m.get_original_embeddings(x, ei, et, return_all_emb=True)
    for got, key in [(z1, 'z1'), (z2, 'z2'), (o1, 'o1'), (o2, 'o2')]:
        assert rel_l2(got.detach().cpu(), rest[key]) < TOL, key
    s = m.decode(z2, t(rest['dec_edge']).cuda(), t(rest['dec_type']).cuda())
    assert rel_l2(s.detach().cpu(), rest['score']) < TOL


@pytest.mark.parametrize('tag', ['dense', 'blocks'])
def test_rgat_delete_matches_reference_golden(tag):
    """RGATConv on the HIP side (attention logits from two [N, R] tables, typed conv kernel with alpha as edge
    weights under no_grad, per-relation autograd loop otherwise) against the reference's own RGAT code."""
    fx = load_golden(f'wiring_rgat_{tag}.npz')
    state, _, rest = split_fixture(fx)
    r = int(rest['num_edge_type'])
    m = hip_model('rgat', state, t(rest['mask1']), t(rest['mask2']), num_nodes=state['node_emb.weight'].shape[0],
                  num_edge_type=r)
    x, ei, et = t(rest['x']).cuda(), t(rest['edge_index']).cuda(), t(rest['edge_type']).cuda()
    z1, z2 = m(x, ei, et, return_all_emb=True)
    o1, o2 = m.get_original_embeddings(x, ei, et, return_all_emb=True)
    for got, key in [(z1, 'z1'), (z2, 'z2'), (o1, 'o1'), (o2, 'o2')]:
        assert rel_l2(got.detach().cpu(), rest[key]) < TOL, key
    s = m.decode(z2, t(rest['dec_edge']).cuda(), t(rest['dec_type']).cuda())
    assert rel_l2(s.detach().cpu(), rest['score']) < TOL
    ((z2 ** 2).mean() + (z1 ** 2).mean()).backward()
    assert rel_l2(m.deletion1.deletion_weight.grad.cpu(), rest['gw1']) < TOL
    assert rel_l2(m.deletion2.deletion_weight.grad.cpu(), rest['gw2']) < TOL
    with torch.no_grad():                                   # evaluation path: fused typed kernel
        e1, e2 = m(x, ei, et, return_all_emb=True)
    assert rel_l2(e1.cpu(), rest['z1']) < TOL and rel_l2(e2.cpu(), rest['z2']) < TOL


@pytest.mark.parametrize('gnn', ['gcn', 'gat', 'gin'])
def test_del_weight_gradients_match_oracle(gnn):
    fx = load_golden(f'wiring_{gnn}.npz')
    state, _, rest = split_fixture(fx)
    mo = oracle_model(gnn, state, t(rest['mask1']), t(rest['mask2']))
    mh = hip_model(gnn, state, t(rest['mask1']), t(rest['mask2']))
    x, ei = t(rest['x']), t(rest['edge_index'])
    g = torch.Generator().manual_seed(0)
    u1, u2 = torch.randn(x.shape[0], 32, generator=g), torch.randn(x.shape[0], 16, generator=g)
    z1, z2 = mo(x, ei, return_all_emb=True)
    ((z1 * u1).sum() + (z2 * u2).sum()).backward()
    h1, h2 = mh(x.cuda(), ei.cuda(), return_all_emb=True)
    ((h1 * u1.cuda()).sum() + (h2 * u2.cuda()).sum()).backward()
    for name in ['deletion1', 'deletion2']:
        got = getattr(mh, name).deletion_weight.grad.cpu()
        want = getattr(mo, name).deletion_weight.grad
        assert rel_l2(got, want) < TOL, name
    assert all(p.grad is None for n_, p in mh.named_parameters() if n_.startswith('conv1'))


TRAJ = [('gat', 'both_layerwise'), ('gat', 'both_all'), ('gat', 'only2_layerwise'), ('gat', 'only2_all'),
        ('gat', 'only1'), ('gin', 'both_layerwise'), ('gcn', 'both_all'), ('gcn', 'only2_layerwise'),
        ('gcn', 'only1')]


@pytest.mark.parametrize('gnn,loss_type', TRAJ)
def test_training_trajectory_matches_reference_loop(gnn, loss_type):
    """The reference's real train_fullbatch loop (golden) reproduced with the HIP model under the
    same update rule (torch Adam on the Del weights, autograd through the HIP ops)."""
    from oracle import gnndelete_ref as R
    fx = load_golden(f'traj_{gnn}_{loss_type}.npz')
    state, data, rest = split_fixture(fx)
    m = hip_model(gnn, state, data['sdf_node_1hop_mask'], data['sdf_node_2hop_mask'])
    m.relational = False
    dev = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in data.items()}
    E = dev['train_pos_edge_index']
    e_dr, e_sdf = E[:, dev['dr_mask']], E[:, dev['sdf_mask']]
    ni1, ni2 = R.non_df_masks(data['x'].shape[0], data['directed_df_edge_index'], data['sdf_node_1hop_mask'],
                              data['sdf_node_2hop_mask'])
    with torch.no_grad():
        z1o, z2o = m.get_original_embeddings(dev['x'], e_dr, return_all_emb=True)
    targets = dict(z1_ori=z1o, z2_ori=z2o, pos_edge=E[:, dev['df_mask']], neg_edge=t(rest['neg']).cuda(),
                   ni_mask1=ni1.cuda(), ni_mask2=ni2.cuda())
    opt = R.make_optimizer(m, loss_type, float(rest['lr']))
    logs = [R.nodeemb_epoch(m, lambda: m(dev['x'], e_sdf, return_all_emb=True), targets, opt, loss_type,
                            float(rest['alpha']), R.LOSSES['mse_mean']) for _ in range(int(rest['epochs']))]
    for key in ['train_loss', 'loss_r', 'loss_l']:
        np.testing.assert_allclose(np.array([l[key] for l in logs]), rest[key], rtol=1e-4, atol=1e-8, err_msg=key)
    assert rel_l2(m.deletion1.deletion_weight.detach().cpu(), rest['final_w1']) < TOL
    assert rel_l2(m.deletion2.deletion_weight.detach().cpu(), rest['final_w2']) < TOL


@pytest.mark.parametrize('gnn,loss_fct', [('gcn', 'kld_mean'), ('gat', 'cosine_sum'), ('gin', 'cosine_mean'), ('gcn', 'kld_sum')])
def test_non_mse_losses_train_like_the_oracle(gnn, loss_fct):
    """--loss_fct kld_* / cosine_* (gnndelete_nodeemb.py:18-28, :196-210): the HIP model with the HIP row-pair loss kernel
    (gd_rowpair_loss_f32 behind framework.trainer.gnndelete_nodeemb.get_loss_fct) against the oracle model with the
    reference's torch expressions, five epochs of the both_all update from the same state."""
    from oracle import gnndelete_ref as R
    from gnndelete_amd.framework.trainer.gnndelete_nodeemb import get_loss_fct
    fx = load_golden(f'traj_{gnn}_both_layerwise.npz' if gnn != 'gcn' else 'traj_gcn_both_all.npz')
    state, data, rest = split_fixture(fx)
    ni1, ni2 = R.non_df_masks(data['x'].shape[0], data['directed_df_edge_index'], data['sdf_node_1hop_mask'],
                              data['sdf_node_2hop_mask'])
    E = data['train_pos_edge_index']
    results = []
    for hip in (False, True):
        make = hip_model if hip else oracle_model
        m = make(gnn, state, data['sdf_node_1hop_mask'], data['sdf_node_2hop_mask'])
        m.relational = False
        to = (lambda v: v.cuda()) if hip else (lambda v: v)
        x, e_dr, e_sdf = to(data['x']), to(E[:, data['dr_mask']]), to(E[:, data['sdf_mask']])
        with torch.no_grad():
            z1o, z2o = m.get_original_embeddings(x, e_dr, return_all_emb=True)
        targets = dict(z1_ori=z1o, z2_ori=z2o, pos_edge=to(E[:, data['df_mask']]), neg_edge=to(t(rest['neg'])),
                       ni_mask1=to(ni1), ni_mask2=to(ni2))
        opt = R.make_optimizer(m, 'both_all', float(rest['lr']))
        fct = get_loss_fct(loss_fct) if hip else R.LOSSES[loss_fct]
        logs = [R.nodeemb_epoch(m, lambda: m(x, e_sdf, return_all_emb=True), targets, opt, 'both_all', 0.5, fct) for _ in range(5)]
        results.append((np.array([l['train_loss'] for l in logs]), m.deletion1.deletion_weight.detach().cpu(),
                        m.deletion2.deletion_weight.detach().cpu()))
    (lo, a1, a2), (lh, b1, b2) = results
    np.testing.assert_allclose(lh, lo, rtol=1e-4, atol=1e-7)
    assert rel_l2(b1, a1) < TOL and rel_l2(b2, a2) < TOL


def test_sage_extension_matches_oracle_forward_and_gradients():
    """GraphSAGE (mean) - named by BASELINE.json config 3, absent from the reference: the HIP model
    against the oracle restatement (itself pinned by a dense KAT only)."""
    from types import SimpleNamespace
    from gnndelete_amd.framework import get_model
    from oracle import gnndelete_ref as R
    from helpers import random_graph
    torch.manual_seed(1)
    n, f = 400, 24
    ei = random_graph(n, 2500, seed=5, isolate=4)
    x = torch.randn(n, f)
    m1, m2 = torch.rand(n) < 0.4, torch.rand(n) < 0.7
    args = SimpleNamespace(unlearning_model='gnndelete_nodeemb', gnn='sage', in_dim=f, hidden_dim=128, out_dim=64)
    hip = get_model(args, m1, m2)
    with torch.no_grad():
        hip.deletion1.deletion_weight.copy_(torch.eye(128) * 0.5 + 0.02 * torch.randn(128, 128))
        hip.deletion2.deletion_weight.copy_(torch.eye(64) * 0.5 + 0.02 * torch.randn(64, 64))
    ref = R.TwoLayerDelete('sage', f, 128, 64, m1, m2)
    ref.load_state_dict(hip.state_dict())
    hip = hip.cuda()
    z1, z2 = hip(x.cuda(), ei.cuda(), return_all_emb=True)
    r1, r2 = ref(x, ei, return_all_emb=True)
    assert rel_l2(z1.detach().cpu(), r1.detach()) < TOL and rel_l2(z2.detach().cpu(), r2.detach()) < TOL
    (z1.pow(2).mean() + z2.pow(2).mean()).backward()
    (r1.pow(2).mean() + r2.pow(2).mean()).backward()
    assert rel_l2(hip.deletion1.deletion_weight.grad.cpu(), ref.deletion1.deletion_weight.grad) < TOL
    assert rel_l2(hip.deletion2.deletion_weight.grad.cpu(), ref.deletion2.deletion_weight.grad) < TOL
    o1, o2 = hip.get_original_embeddings(x.cuda(), ei.cuda(), return_all_emb=True)
    q1, q2 = ref.get_original_embeddings(x, ei, return_all_emb=True)
    assert rel_l2(o2.detach().cpu(), q2.detach()) < TOL


def test_typed_csr_cache_sees_in_place_edits():
    """The R-GCN / R-GAT convs cache their relation-typed CSR per (edge_index, edge_type); editing either tensor IN
    PLACE must rebuild it (keyed on storage address + torch's version counter, not on object identity alone)."""
    from gnndelete_amd.nn import RGCNConv
    from oracle import pyg_semantics as pyg
    torch.manual_seed(0)
    n, r = 50, 3
    conv = RGCNConv(8, 8, r).cuda()
    x = torch.randn(n, 8).cuda()
    ei = torch.randint(0, n, (2, 200)).cuda()
    et = torch.randint(0, r, (200,)).cuda()

    def ref():
        return pyg.rgcn_conv(x.cpu(), ei.cpu(), et.cpu(), conv.weight.detach().cpu(), conv.root.detach().cpu(),
                             conv.bias.detach().cpu())
    with torch.no_grad():
        assert rel_l2(conv(x, ei, et).cpu(), ref()) < TOL
        ei[0, :50] = torch.randint(0, n, (50,)).cuda()          # same tensor object, new content
        assert rel_l2(conv(x, ei, et).cpu(), ref()) < TOL
        et[:50] = (et[:50] + 1) % r
        assert rel_l2(conv(x, ei, et).cpu(), ref()) < TOL
