"""Layer-1 / layer-2 SpMM on the bench graph (synth-collab in its locality order): duration for a sweep of
grid caps (GD_SPMM_GRID_CAP is read at every launch)."""
import sys, os, torch
sys.path.insert(0, '.')
from gnndelete_amd.framework.synth import dcsbm_edges
from gnndelete_amd.graph import build_csr
from gnndelete_amd.reorder import locality_order
from gnndelete_amd import ops, _lib
if os.environ.get('GD_AB_LIB'): _lib.LIB_PATH = os.path.abspath(os.environ['GD_AB_LIB'])   # A/B against another build
n, m = 235868, 1179052
E, comm = dcsbm_edges(n, m, 42)
ei = torch.cat([E, E.flip(0)], 1).cuda()
perm, inv = locality_order(ei, n)
ei = inv[ei]
g = build_csr(ei.contiguous(), n, 'gcn')
caps = [int(c) for c in os.environ.get('CAPS', '2048,4096,8192,16384').split(',')]
for d in (128, 64):
    x = torch.randn(n, d, device='cuda'); y = torch.empty_like(x); b = torch.randn(d, device='cuda')
    ref = None
    for cap in caps:
        os.environ['GD_SPMM_GRID_CAP'] = str(cap)
        for _ in range(5): ops._spmm_raw(g.rowptr, g.col, g.val, x, b, 0.0, n, g.plan, out=y)
        torch.cuda.synchronize()
        if ref is None:
            ref = torch.sparse_csr_tensor(g.rowptr.long(), g.col.long(), g.val, (n, n)) @ x + b
            err = ((y - ref).norm() / ref.norm()).item()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): ops._spmm_raw(g.rowptr, g.col, g.val, x, b, 0.0, n, g.plan, out=y)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        a = 4 * (n + 1) + 8 * g.nnz + 8 * n * d
        print(f'd={d} cap={cap}: nnz={g.nnz} {us:.1f} us (with fix-up)  {a / us / 1e3:.0f} GB/s = {a / us / 8e6:.3f} of 8 TB/s  rel err {err:.2e}', flush=True)
