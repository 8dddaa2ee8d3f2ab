import sys, torch, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from types import SimpleNamespace
from test_engine_gpu import _kg_request
from gnndelete_amd.engine import NodeembEngine
from gnndelete_amd.framework.models import RGCNDelete
from oracle import gnndelete_ref as R
rel = lambda a, b: float((a.double().cpu() - b.double().cpu()).norm() / b.double().cpu().norm())
i, h, o, nr = 128, 128, 64, 51
data = _kg_request(700, 5000, nr, seed=3, n_df=60)
n = data.num_nodes
ni1, ni2 = R.non_df_masks(n, data.directed_df_edge_index, data.sdf_node_1hop_mask, data.sdf_node_2hop_mask)
torch.manual_seed(5)
hip = RGCNDelete(SimpleNamespace(in_dim=i, hidden_dim=h, out_dim=o), n, nr, ni1, ni2)
with torch.no_grad():
    for name, p_ in hip.named_parameters():
        if name.endswith('bias'):
            p_.copy_(torch.randn_like(p_) * 0.1)
ref = R.TwoLayerDelete('rgcn', i, h, o, ni1, ni2, num_nodes=n, num_edge_type=nr)
ref.load_state_dict(hip.state_dict(), strict=False)
ei, et = data.edge_index[:, data.dr_mask], data.edge_type[data.dr_mask]
pos, pt = data.edge_index[:, data.df_mask], data.edge_type[data.df_mask]
fw = pt < nr
dec, dec_t = pos[:, fw], pt[fw]
torch.manual_seed(9)
neg = R.negative_sampling_kg(dec, dec_t)
with torch.no_grad():
    z1o, z2o = ref.get_original_embeddings(data.x, ei, et, return_all_emb=True)
targets = dict(z1_ori=z1o, z2_ori=z2o, pos_edge=dec, neg_edge=neg, ni_mask1=ni1, ni_mask2=ni2)
for lt in ('both_all',):
    opt = R.make_optimizer(ref, lt, 1e-2)
    hipg = hip.cuda()
    eng = NodeembEngine(hipg, data.x.cuda(), ei.cuda().contiguous(), z1o.cuda(), z2o.cuda(), dec.cuda(), neg.cuda(), ni1, ni2,
                        loss_type=lt, alpha=0.4, lr=1e-2, use_graph=False, edge_type=et.cuda().contiguous())
    print('split1', eng._split1, 'split2', eng._split2, 'fuse1', eng._fuse_loss1, 'fuse2', eng._fuse_l2, 'folded', eng.t1.folded, eng.t2.folded)
    prev1 = torch.zeros(h, h); prev2 = torch.zeros(o, o)
    for it in range(5):
        log = R.nodeemb_epoch(ref, lambda: ref(data.x, ei, et, return_all_emb=True), targets, opt, lt, 0.4, R.LOSSES['mse_mean'])
        eng.step()
        g1r, g2r = ref.deletion1.deletion_weight.grad.clone(), ref.deletion2.deletion_weight.grad.clone()
        print(it, 'g1 acc rel', rel(eng.g1, g1r), 'g2 acc rel', rel(eng.g2, g2r), ' per-iter g1 rel', rel(eng.g1.cpu() - prev1, g1r - prev1),
              'W1', rel(hipg.deletion1.deletion_weight.detach(), ref.deletion1.deletion_weight.detach()), 'z1', rel(eng.z1[ni1.cuda()], log['z1'][ni1]))
        zr = log['z1'][ni1]; zh = eng.z1[ni1.cuda()].cpu()
        flips = ((zr > 0) != (zh > 0))
        print('   gate flips', int(flips.sum()), 'min |z| among flips', float(zr[flips].abs().min()) if flips.any() else None, ' entries with |z|<1e-6:', int((zr.abs() < 1e-6).sum()))
        prev1 = g1r.clone()
