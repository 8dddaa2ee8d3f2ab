"""Debug: the partitioned R-GCN engine in a world of one, eager, synchronising after every segment."""
import sys, os
sys.path.insert(0, '.')
from types import SimpleNamespace
import torch
import bench
from gnndelete_amd.dist_engine import PartitionedNodeembEngine
dev = torch.device('cuda', 0)
args = SimpleNamespace(gnn='rgcn', workload='synth-kg-small', seed=42, df='in', df_size=2.5, loss_type=sys.argv[1] if len(sys.argv) > 1 else 'both_layerwise', no_graph=True)
data, model, neg, ni1, ni2 = bench.build_kg_request(args)
model = model.to(dev)
ei = data.edge_index[:, data.dr_mask].to(dev).contiguous()
et = data.edge_type[data.dr_mask].to(dev).contiguous()
x = data.x.to(dev)
with torch.no_grad():
    z1o, z2o = model.get_original_embeddings(x, ei, et, return_all_emb=True)
torch.cuda.synchronize(); print('orig ok', flush=True)
eng = PartitionedNodeembEngine(model, x, ei, z1o, z2o, data.kg_dec_edge.to(dev), neg.to(dev), ni1, ni2, 0, 1, loss_type=args.loss_type,
                               alpha=0.5, lr=1e-3, use_graph=False, edge_type=et)
torch.cuda.synchronize(); print('engine built', eng.s1, eng.s2, eng.t1.n_rows, eng.t2.n_rows, flush=True)
with torch.no_grad():
    for it in range(2):
        for op in eng._program():
            if isinstance(op, tuple):
                eng._comm(op[1], op[2])
            else:
                op()
            torch.cuda.synchronize()
            print(it, getattr(op, '__name__', op), 'ok', flush=True)
print('hist', eng.hist[:2].tolist())
