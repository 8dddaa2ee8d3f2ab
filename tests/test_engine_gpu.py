"""Fused Del-training step (gnndelete_amd.engine.NodeembEngine, explicit backward, hipGraph
replay) vs the trajectories of the reference's real train_fullbatch loop (golden vectors) for
every loss_type, plus graph-vs-eager bit equality.  Tolerance: 1e-4 rel (north_star)."""
import numpy as np
import pytest
import torch

from helpers import hip_model, load_golden, rel_l2, split_fixture, t

pytestmark = pytest.mark.gpu

TRAJ = [('gat', 'both_layerwise'), ('gat', 'both_all'), ('gat', 'only2_layerwise'), ('gat', 'only2_all'),
        ('gat', 'only1'), ('gin', 'both_layerwise'), ('gcn', 'both_all'), ('gcn', 'only2_layerwise'),
        ('gcn', 'only1')]


def make_engine(gnn, loss_type, use_graph, fx=None):
    from gnndelete_amd.engine import NodeembEngine
    from oracle import gnndelete_ref as R
    fx = fx or load_golden(f'traj_{gnn}_{loss_type}.npz')
    state, data, rest = split_fixture(fx)
    m = hip_model(gnn, state, data['sdf_node_1hop_mask'], data['sdf_node_2hop_mask'])
    dev = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in data.items()}
    E = dev['train_pos_edge_index']
    ni1, ni2 = R.non_df_masks(data['x'].shape[0], data['directed_df_edge_index'], data['sdf_node_1hop_mask'],
                              data['sdf_node_2hop_mask'])
    with torch.no_grad():
        z1o, z2o = m.get_original_embeddings(dev['x'], E[:, dev['dr_mask']], return_all_emb=True)
    eng = NodeembEngine(m, dev['x'], E[:, dev['sdf_mask']].contiguous(), z1o, z2o, E[:, dev['df_mask']],
                        t(rest['neg']).cuda(), ni1, ni2, loss_type=loss_type, alpha=float(rest['alpha']),
                        lr=float(rest['lr']), use_graph=use_graph)
    return eng, m, rest


@pytest.mark.parametrize('use_graph', [False, True])
@pytest.mark.parametrize('gnn,loss_type', TRAJ)
def test_engine_reproduces_reference_trajectory(gnn, loss_type, use_graph):
    eng, m, rest = make_engine(gnn, loss_type, use_graph)
    for _ in range(int(rest['epochs'])):
        eng.step()
    hist = eng.loss_history().numpy()
    for col, key in enumerate(['train_loss', 'loss_r', 'loss_l']):
        np.testing.assert_allclose(hist[:, col], rest[key], rtol=1e-4, atol=1e-8, err_msg=key)
    assert rel_l2(m.deletion1.deletion_weight.detach().cpu(), rest['final_w1']) < 1e-4
    assert rel_l2(m.deletion2.deletion_weight.detach().cpu(), rest['final_w2']) < 1e-4


@pytest.mark.parametrize('gnn,loss_type', [('gcn', 'both_layerwise'), ('gat', 'both_layerwise')])
def test_graph_replay_is_bit_identical_to_eager(gnn, loss_type):
    fx = load_golden(f'traj_gat_{loss_type}.npz') if gnn == 'gat' else load_golden('traj_gcn_both_all.npz')
    a, ma, rest = make_engine(gnn, loss_type, False, fx)
    b, mb, _ = make_engine(gnn, loss_type, True, fx)
    for _ in range(5):
        a.step()
        b.step()
    assert torch.equal(ma.deletion1.deletion_weight, mb.deletion1.deletion_weight)
    assert torch.equal(ma.deletion2.deletion_weight, mb.deletion2.deletion_weight)
    assert torch.equal(a.loss_history(), b.loss_history())


def test_gcn_both_layerwise_runs_where_upstream_crashes():
    """Upstream GCNDelete + both_layerwise raises (MANIFEST.json records the error); with the
    backbone frozen it is well defined and must agree with the oracle under the same rule."""
    from oracle import gnndelete_ref as R
    from helpers import oracle_model
    fx = load_golden('traj_gcn_both_all.npz')
    eng, m, rest = make_engine('gcn', 'both_layerwise', True, fx)
    state, data, _ = split_fixture(fx)
    mo = oracle_model('gcn', state, data['sdf_node_1hop_mask'], data['sdf_node_2hop_mask'])
    logs, _ = R.nodeemb_fullbatch(mo, data, 6, 'both_layerwise', float(rest['alpha']), 'mse_mean',
                                  float(rest['lr']), neg_edge=t(rest['neg']))
    for _ in range(6):
        eng.step()
    hist = eng.loss_history().numpy()
    np.testing.assert_allclose(hist[:, 0], [l['train_loss'] for l in logs], rtol=1e-4)
    assert rel_l2(m.deletion1.deletion_weight.detach().cpu(), mo.deletion1.deletion_weight.detach()) < 1e-4
    assert rel_l2(m.deletion2.deletion_weight.detach().cpu(), mo.deletion2.deletion_weight.detach()) < 1e-4
