import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from types import SimpleNamespace
from gnndelete_amd.engine import NodeembEngine
from gnndelete_amd.framework.data import prepare_edge_deletion
from gnndelete_amd.framework.graph_utils import negative_sampling
from gnndelete_amd.framework.models import GCNDelete
from gnndelete_amd.framework.synth import make_linkpred_dataset
from oracle import gnndelete_ref as R
data, dfm = make_linkpred_dataset(None, seed=7, shape=(3000, 48, 15000, 'dense'))
torch.manual_seed(7); prepare_edge_deletion(data, dfm['in'], 400)
gen = torch.Generator().manual_seed(7)
neg = negative_sampling(data.train_pos_edge_index, data.num_nodes, int(data.df_mask.sum()), generator=gen)
keep = torch.ones(data.num_nodes, dtype=torch.bool); keep[data.directed_df_edge_index.flatten().unique()] = False
ni1, ni2 = data.sdf_node_1hop_mask & keep, data.sdf_node_2hop_mask & keep
E = data.train_pos_edge_index; e_dr, e_sdf, pos = E[:, data.dr_mask], E[:, data.sdf_mask], E[:, data.df_mask]
torch.manual_seed(3)
base = GCNDelete(SimpleNamespace(in_dim=48, hidden_dim=128, out_dim=64), data.sdf_node_1hop_mask, data.sdf_node_2hop_mask)
import os
if os.environ.get('SCALE'):
    with torch.no_grad():
        for n_, p_ in base.named_parameters():
            if 'deletion' not in n_ and p_.dim() > 1: p_.mul_(2.0)
state = {k: v.clone() for k, v in base.state_dict().items()}
ref = R.TwoLayerDelete('gcn', 48, 128, 64, data.sdf_node_1hop_mask, data.sdf_node_2hop_mask); ref.load_state_dict(state)
with torch.no_grad(): z1o, z2o = ref.get_original_embeddings(data.x, e_dr, return_all_emb=True)
targets = dict(z1_ori=z1o, z2_ori=z2o, pos_edge=pos, neg_edge=neg, ni_mask1=ni1, ni_mask2=ni2)
opt = R.make_optimizer(ref, 'both_layerwise', 1e-2)
# the same recipe in float64: the yardstick for how far fp32 itself drifts
ref64 = R.TwoLayerDelete('gcn', 48, 128, 64, data.sdf_node_1hop_mask, data.sdf_node_2hop_mask); ref64.load_state_dict(state); ref64 = ref64.double()
x64 = data.x.double()
targets64 = dict(targets, z1_ori=z1o.double(), z2_ori=z2o.double())
opt64 = R.make_optimizer(ref64, 'both_layerwise', 1e-2)
engs = {}
for name, kw in [('graph', dict(use_graph=True))]:
    m = GCNDelete(SimpleNamespace(in_dim=48, hidden_dim=128, out_dim=64), data.sdf_node_1hop_mask, data.sdf_node_2hop_mask)
    m.load_state_dict(state); m = m.cuda()
    engs[name] = (m, NodeembEngine(m, data.x.cuda(), e_sdf.cuda().contiguous(), z1o.cuda(), z2o.cuda(), pos.cuda(), neg.cuda(), ni1, ni2, loss_type='both_layerwise', alpha=0.5, lr=1e-2, **kw))
rel = lambda a, b: float((a.double().cpu() - b.double()).norm() / b.double().norm())
for ep in range(1, 61):
    R.nodeemb_epoch(ref, lambda: ref(data.x, e_sdf, return_all_emb=True), targets, opt, 'both_layerwise', 0.5, R.LOSSES['mse_mean'])
    R.nodeemb_epoch(ref64, lambda: ref64(x64, e_sdf, return_all_emb=True), targets64, opt64, 'both_layerwise', 0.5, R.LOSSES['mse_mean'])
    for name, (m, e) in engs.items(): e.step()
    if 4 <= ep <= 11:
        w32 = ref.deletion1.deletion_weight.detach(); mh = engs['graph'][0].deletion1.deletion_weight.detach().cpu()
        dd = (mh - w32).abs(); i = int(dd.argmax()); r, c = divmod(i, 128)
        st = opt[0].state[ref.deletion1.deletion_weight] if isinstance(opt, (list, tuple)) else None
        e = engs['graph'][1]
        print('ep', ep, 'argmax', (r, c), 'diff', float(dd.max()), 'w32', float(w32[r, c]), 'whip', float(mh[r, c]),
              'grad32', float(ref.deletion1.deletion_weight.grad[r, c]), 'g1hip', float(e.g1[r, c]),
              'm32', float(st['exp_avg'][r, c]) if st else None, 'mhip', float(e.adam1.m[r, c]),
              'v32', float(st['exp_avg_sq'][r, c]) if st else None, 'vhip', float(e.adam1.v[r, c]))
    if ep in (1, 2, 3, 5, 10, 20, 40, 60):
        w32, w64 = ref.deletion1.deletion_weight.detach(), ref64.deletion1.deletion_weight.detach()
        mh = engs['graph'][0].deletion1.deletion_weight.detach()
        dd = (mh.cpu() - w32).abs()
        print(ep, 'cpu32-vs-64', round(rel(w32, w64), 7), 'hip-vs-64', round(rel(mh, w64), 7), 'max|hip-cpu32|', float(dd.max()), 'n>1e-4', int((dd > 1e-4).sum()))
        print(ep, {name: (round(rel(m.deletion1.deletion_weight.detach(), ref.deletion1.deletion_weight.detach()), 7), round(rel(m.deletion2.deletion_weight.detach(), ref.deletion2.deletion_weight.detach()), 7)) for name, (m, e) in engs.items()})
