"""The three SpMM launches of the bench step (step graph d=128 / d=64 with bias, transposed d=64 without),
each timed back to back on the step's own operands; GNNDELETE_HIP_LIB selects the build."""
import sys, os, torch
sys.path.insert(0, '.')
import bench
from gnndelete_amd import ops
sys.argv = ['bench.py']
args = bench.parse()
dev = torch.device('cuda', 0)
data, model, neg, ni1, ni2 = bench.build_request(args, dev)
eng = bench.make_engine(args, data, model, neg, ni1, ni2, dev)
g, n = eng.graph, eng.n


def timed(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for d, tr in ((128, False), (64, False), (64, True)):
    x = torch.randn(n, d, device=dev); y = torch.empty_like(x); b = None if tr else torch.randn(d, device=dev)
    rp, col, val, plan = (g.rowptr_t, g.col_t, g.val_t, g.plan_t) if tr else (g.rowptr, g.col, g.val, g.plan)
    us = timed(lambda: ops._spmm_raw(rp, col, val, x, b, 0.0, n, plan, out=y))
    print(f'd={d} transposed={tr}: {us:.1f} us (with fix-up), items={plan.n_items} nnz={g.nnz}', flush=True)
