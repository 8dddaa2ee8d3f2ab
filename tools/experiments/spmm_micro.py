import sys, os, torch
sys.path.insert(0, '.')
from gnndelete_amd.framework.synth import dcsbm_edges
from gnndelete_amd.graph import build_csr
from gnndelete_amd import ops
n, m = 235868, 1179052
E, comm = dcsbm_edges(n, m, 42)
if os.environ.get('SORTED'):
    order = torch.argsort(comm * n + torch.arange(n)); new_id = torch.empty(n, dtype=torch.long); new_id[order] = torch.arange(n); E = new_id[E]
d = int(os.environ.get('D', 128))
ei = torch.cat([E, E.flip(0)], 1).cuda()
g = build_csr(ei, n, 'gcn')
x = torch.randn(n, d, device='cuda'); y = torch.empty_like(x)
for _ in range(5): ops._spmm_raw(g.rowptr, g.col, g.val, x, None, 0.0, n, g.plan, out=y)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): ops._spmm_raw(g.rowptr, g.col, g.val, x, None, 0.0, n, g.plan, out=y)
e1.record(); torch.cuda.synchronize()
print('avg us', e0.elapsed_time(e1)/10*1e3)
