"""How much of the layer-1 SpMM's distance to the HBM roofline is the graph and how much the kernel: the same
kernel, N, nnz and degree sequence as the bench graph, but with every edge's far endpoint re-drawn within a
window of +-W node ids (W = inf keeps the bench graph in its locality order)."""
import sys, os, torch
sys.path.insert(0, '.')
from gnndelete_amd.framework.synth import dcsbm_edges
from gnndelete_amd.graph import build_csr
from gnndelete_amd.reorder import locality_order
from gnndelete_amd import ops
n, m, d = 235868, 1179052, int(os.environ.get('D', 128))
E, comm = dcsbm_edges(n, m, 42)
ei = torch.cat([E, E.flip(0)], 1).cuda()
perm, inv = locality_order(ei, n)
ei = inv[ei]
g0 = torch.Generator(device='cuda').manual_seed(1)
x = torch.randn(n, d, device='cuda'); y = torch.empty_like(x)
alg = 4 * (n + 1) + 8 * (ei.shape[1] + n) + 8 * n * d
WS = [None if w == 'None' else int(w) for w in os.environ['WINDOWS'].split(',')] if os.environ.get('WINDOWS') else [None, 65536, 16384, 4096, 1024, 64]
for W in WS:
    if W is None:
        e = ei
    else:
        src = ei[0]
        off = torch.randint(-W, W + 1, src.shape, device='cuda', generator=g0)
        dst = (src + off) % n            # wrap (clamping would pile the ends up into two mega-hubs)
        e = torch.stack([src, dst])
    g = build_csr(e.contiguous(), n, 'gcn')
    for _ in range(5): ops._spmm_raw(g.rowptr, g.col, g.val, x, None, 0.0, n, g.plan, out=y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops._spmm_raw(g.rowptr, g.col, g.val, x, None, 0.0, n, g.plan, out=y)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    a = 4 * (n + 1) + 8 * g.nnz + 8 * n * d
    print(f'd={d} window={W}: nnz={g.nnz} {us:.1f} us  {a / us / 1e3:.0f} GB/s algorithmic = {a / us / 1e3 / 8000:.2f} of 8 TB/s')
