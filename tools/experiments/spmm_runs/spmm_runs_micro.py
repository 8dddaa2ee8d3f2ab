"""The three aggregations of the bench step on the run-item kernel (gd_spmm_csr_runs_f32) against the item kernel
(gd_spmm_csr_onepass_f32), each timed back to back on the step's own graph, for a few packing windows and grid caps.
GD_SPMM_RUN_WIN / GD_SPMM_RUN_LIGHT / GD_SPMM_RUNS_GRID are read per process: this script times ONE configuration
(tools/experiments/r05_spmm_runs.sh loops over them)."""
import os, sys, torch
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


tag = f"win={os.environ.get('GD_SPMM_RUN_WIN', '32')} light={os.environ.get('GD_SPMM_RUN_LIGHT', '32')} grid={os.environ.get('GD_SPMM_RUNS_GRID', '8192')}"
for d, tr in ((128, False), (64, False), (64, True)):
    x = torch.randn(n, d, device=dev); y = torch.empty_like(x); b = None if tr else torch.randn(d, device=dev)
    rp, col, val, plan = (g.rowptr_t, g.col_t, g.val_t, g.plan_t) if tr else (g.rowptr, g.col, g.val, g.plan)
    os.environ['GD_SPMM_RUNS'] = '0'
    us0 = timed(lambda: ops._spmm_raw(rp, col, val, x, b, 0.0, n, plan, out=y))
    ref = y.clone()
    os.environ['GD_SPMM_RUNS'] = '2'
    items, ni, _ = plan.runs(d)
    us1 = timed(lambda: ops._spmm_raw(rp, col, val, x, b, 0.0, n, plan, out=y))
    err = float((y - ref).norm() / ref.norm())
    print(f'{tag} d={d} transposed={tr}: item kernel {us0:.1f} us, run kernel {us1:.1f} us ({ni} items), rel diff {err:.1e}', flush=True)
