"""Debug: the R-GCN engine with / without the tail launch on the unfused stages (GD_TAIL_FUSED_ONLY)."""
import os, sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from types import SimpleNamespace
from test_engine_gpu import _kg_request
from gnndelete_amd.engine import NodeembEngine
from gnndelete_amd.framework.models import RGCNDelete
from oracle import gnndelete_ref as R
i, h, o, nr = 128, 128, 64, 51
data = _kg_request(700, 5000, nr, seed=3, n_df=60)
n = data.num_nodes
ni1, ni2 = R.non_df_masks(n, data.directed_df_edge_index, data.sdf_node_1hop_mask, data.sdf_node_2hop_mask)
torch.manual_seed(5)
hip = RGCNDelete(SimpleNamespace(in_dim=i, hidden_dim=h, out_dim=o), n, nr, ni1, ni2).cuda()
state = {k: v.clone() for k, v in hip.state_dict().items()}
ei, et = data.edge_index[:, data.dr_mask].cuda().contiguous(), data.edge_type[data.dr_mask].cuda().contiguous()
pos, pt = data.edge_index[:, data.df_mask], data.edge_type[data.df_mask]
fw = pt < nr
dec = pos[:, fw].cuda()
torch.manual_seed(9)
neg = R.negative_sampling_kg(pos[:, fw], pt[fw]).cuda()
with torch.no_grad():
    z1o, z2o = hip.get_original_embeddings(data.x.cuda(), ei, et, return_all_emb=True)
res = {}
for mode in ('1', '0'):
    for ug in (False, True):
        os.environ['GD_TAIL_FUSED_ONLY'] = mode
        hip.load_state_dict(state)
        eng = NodeembEngine(hip, data.x.cuda(), ei, z1o, z2o, dec, neg, ni1, ni2, loss_type='both_layerwise', alpha=0.4, lr=1e-2, use_graph=ug, edge_type=et)
        for _ in range(3):
            eng.step()
        torch.cuda.synchronize()
        res[(mode, ug)] = (eng.loss_history().clone(), hip.deletion1.deletion_weight.detach().clone(), hip.deletion2.deletion_weight.detach().clone())
        print('fused_only', mode, 'graph', ug, 'tail', eng._tail, 'fuse1', eng._fuse_loss1, 'fuse2', eng._fuse_l2, 's1', eng.s1, 's2', eng.s2, 'hist', eng.loss_history()[:, 0].tolist(),
              'w1', float(hip.deletion1.deletion_weight.sum()), 'w2', float(hip.deletion2.deletion_weight.sum()))
