"""Which stage of the R-GCN Del backward carries the one wrong dz1 row at biokg size?"""
import sys, torch
sys.path.insert(0, '.')
from types import SimpleNamespace
import torch.nn.functional as F
from oracle import gnndelete_ref as R
from gnndelete_amd.framework.models import RGCNDelete
from gnndelete_amd.framework.synth import make_kg_dataset
torch.set_num_threads(32)
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
ref = R.TwoLayerDelete('rgcn', 128, 128, 64, m1, m2, num_nodes=n, num_edge_type=nr).double()
ref.load_state_dict({k: v.double() for k, v in hip.state_dict().items()}, strict=False)
def run(m, x, ei_, et_, a, b):
    with torch.no_grad():
        p1 = m.conv1(m.node_emb(x), ei_, et_)
    x1 = m.deletion1(p1); x1.retain_grad()
    r = F.relu(x1); r.retain_grad()
    c2 = m.conv2(r, ei_, et_); c2.retain_grad()
    x2 = m.deletion2(c2); x2.retain_grad()
    ((x1[a] ** 2).mean() + (x2[b] ** 2).mean()).backward()
    return dict(x1=x1.detach(), c2=c2.detach(), x2=x2.detach(), gx2=x2.grad, gc2=c2.grad, gr=r.grad, gx1=x1.grad)
o = run(ref, data.x, ei, ety, m1, m2)
hip = hip.cuda()
h = run(hip, data.x.cuda(), ei.cuda(), ety.cuda(), m1.cuda(), m2.cuda())
for k in o:
    d = (h[k].double().cpu() - o[k]).norm(dim=1)
    w = torch.topk(d, 3)
    print(k, 'rel', float(d.norm() / o[k].norm()), 'worst rows', w.indices.tolist(), [f'{v:.3e}' for v in w.values.tolist()],
          'ref norms', [f'{v:.3e}' for v in o[k][w.indices].norm(dim=1).tolist()])
row = 30376
deg_out = int((ei[0] == row).sum()); deg_in = int((ei[1] == row).sum())
print('row', row, 'out-degree', deg_out, 'in-degree', deg_in, 'in m1', bool(m1[row]), 'in m2', bool(m2[row]))
