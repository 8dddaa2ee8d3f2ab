"""north_star parity at BASELINE's full size (synth-collab, GCN / GraphSAGE, 5 % IN): the HIP engine and the CPU
oracle train the same request for EPOCHS steps from the same state with the same negatives; report the
affected-node embedding gap (rel-L2 over the S_Df rows) and the link-prediction AUCs of both."""
import sys, os, time, torch
sys.path.insert(0, '.')
import bench
from oracle import gnndelete_ref as R
from sklearn.metrics import roc_auc_score
sys.argv = ['bench.py', '--gnn', os.environ.get('GNN', 'gcn')]
args = bench.parse()
epochs = int(os.environ.get('EPOCHS', 30)); lr = float(os.environ.get('LR', 1e-3))
dev = torch.device('cuda')
data, model, neg, ni1, ni2 = bench.build_request(args, dev)
state = {k: v.clone() for k, v in model.state_dict().items()}
torch.set_num_threads(32)
ref = R.TwoLayerDelete(args.gnn, data.x.shape[1], 128, 64, data.sdf_node_1hop_mask, data.sdf_node_2hop_mask)
ref.load_state_dict(state, strict=False)
E = data.train_pos_edge_index
e_dr, e_sdf = E[:, data.dr_mask], E[:, data.sdf_mask]
with torch.no_grad():
    z1o, z2o = ref.get_original_embeddings(data.x, e_dr, return_all_emb=True)
targets = dict(z1_ori=z1o, z2_ori=z2o, pos_edge=E[:, data.df_mask], neg_edge=neg, ni_mask1=ni1, ni_mask2=ni2)
opt = R.make_optimizer(ref, args.loss_type, lr)
t0 = time.time()
for _ in range(epochs):
    R.nodeemb_epoch(ref, lambda: ref(data.x, e_sdf, return_all_emb=True), targets, opt, args.loss_type, 0.5, R.LOSSES['mse_mean'])
t_cpu = time.time() - t0
from gnndelete_amd.engine import NodeembEngine
hip = model.to(dev)
eng = NodeembEngine(hip, data.x.to(dev), e_sdf.to(dev).contiguous(), z1o.to(dev), z2o.to(dev), E[:, data.df_mask].to(dev), neg.to(dev), ni1, ni2, loss_type=args.loss_type, alpha=0.5, lr=lr,
                    cache_layer1=os.environ.get('TRAINER_DEFAULTS') == '1', affected_rows_only=os.environ.get('TRAINER_DEFAULTS') == '1')
torch.cuda.synchronize(); t0 = time.time()
for _ in range(epochs): eng.step()
torch.cuda.synchronize(); t_gpu = time.time() - t0
with torch.no_grad():
    r1, r2 = ref(data.x, e_dr, return_all_emb=True)
    h1, h2 = hip(data.x.to(dev), e_dr.to(dev).contiguous(), return_all_emb=True)
h1, h2 = h1.cpu(), h2.cpu()
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
m1, m2 = data.sdf_node_1hop_mask, data.sdf_node_2hop_mask
def auc(z, p, n_):
    ei = torch.cat([p, n_], 1); s = (z[ei[0]] * z[ei[1]]).sum(-1).sigmoid()
    return roc_auc_score(torch.cat([torch.ones(p.shape[1]), torch.zeros(n_.shape[1])]).numpy(), s.numpy())
g = torch.Generator().manual_seed(0)
dr_s = e_dr[:, torch.randperm(e_dr.shape[1], generator=g)[:data.directed_df_edge_index.shape[1]]]
out = dict(gnn=args.gnn, epochs=epochs, lr=lr, z1_sdf_rel_l2=rel(h1[m1], r1[m1]), z2_sdf_rel_l2=rel(h2[m2], r2[m2]),
           dt_auc_cpu=auc(r2, data.test_pos_edge_index, data.test_neg_edge_index), dt_auc_hip=auc(h2, data.test_pos_edge_index, data.test_neg_edge_index),
           df_auc_cpu=auc(r2, dr_s, data.directed_df_edge_index), df_auc_hip=auc(h2, dr_s, data.directed_df_edge_index),
           wd1_rel_l2=rel(hip.deletion1.deletion_weight.detach().cpu(), ref.deletion1.deletion_weight.detach()),
           cpu_s=round(t_cpu, 1), gpu_s=round(t_gpu, 3))
print(out)
