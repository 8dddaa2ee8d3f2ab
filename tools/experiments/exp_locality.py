import sys, time, torch
sys.path.insert(0, '.')
from gnndelete_amd.framework.synth import dcsbm_edges
from gnndelete_amd.graph import build_csr
from gnndelete_amd import ops
n, m = 235868, 1179052
E, comm = dcsbm_edges(n, m, 42)
def bench(E, tag, d=128):
    ei = torch.cat([E, E.flip(0)], 1).cuda()
    g = build_csr(ei, n, 'gcn')
    x = torch.randn(n, d, device='cuda'); y = torch.empty_like(x)
    for _ in range(3): ops._spmm_raw(g.rowptr, g.col, g.val, x, None, 0.0, n, g.plan, out=y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops._spmm_raw(g.rowptr, g.col, g.val, x, None, 0.0, n, g.plan, out=y)
    e1.record(); torch.cuda.synchronize()
    print(f'{tag:28s} d={d} nnz={g.nnz} items={g.plan.n_items} {e0.elapsed_time(e1)/20*1e3:.1f} us')
for d in (128, 64):
    bench(E, 'random ids', d)
    # community-contiguous relabel
    order = torch.argsort(comm * n + torch.arange(n))       # nodes sorted by community
    new_id = torch.empty(n, dtype=torch.long); new_id[order] = torch.arange(n)
    bench(new_id[E], 'community-sorted ids', d)
    deg = torch.bincount(E.flatten(), minlength=n)
    order = torch.argsort(-deg); new_id = torch.empty(n, dtype=torch.long); new_id[order] = torch.arange(n)
    bench(new_id[E], 'degree-sorted ids', d)
    key = comm * (int(deg.max()) + 1) + (int(deg.max()) - deg)
    order = torch.argsort(key); new_id = torch.empty(n, dtype=torch.long); new_id[order] = torch.arange(n)
    bench(new_id[E], 'community, then degree', d)
