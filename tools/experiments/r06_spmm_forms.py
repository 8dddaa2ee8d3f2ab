"""Round 6, VERDICT r5 item 1: the 64-float aggregations of the bench step in other shapes (GD_SPMM_D64_FORM, csrc/spmm.hip
launch_persist), back to back on the step's own graph: forward (with bias) and transposed, weighted (val) and unweighted
(val = None; 'n' = the instantiation without a weight stream).  Checks every form against the product form first."""
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


torch.manual_seed(0)
x = torch.randn(n, 64, device=dev)
b = torch.randn(64, device=dev)
forms = ['', '16x1n', '8x2u2', '8x2u2n', '8x2u4', '8x2u4n']
for tr in (False, True):
    rp, col, val, plan = (g.rowptr_t, g.col_t, g.val_t, g.plan_t) if tr else (g.rowptr, g.col, g.val, g.plan)
    bias = None if tr else b
    ref = {}
    for weighted in (True, False):
        v = val if weighted else None
        for form in forms:
            if weighted and form.endswith('n'):
                continue
            os.environ['GD_SPMM_D64_FORM'] = form
            y = torch.empty_like(x)
            ops._spmm_raw(rp, col, v, x, bias, 0.0, n, plan, out=y)
            torch.cuda.synchronize()
            if form == '':
                ref[weighted] = y.clone()
            rel = float((y - ref[weighted]).norm() / ref[weighted].norm())
            rounds = [timed(lambda: ops._spmm_raw(rp, col, v, x, bias, 0.0, n, plan, out=y)) for _ in range(3)]
            print(f'transposed={int(tr)} weighted={int(weighted)} form={form or "16x1 (product)":16s} {min(rounds):6.1f} us (3 rounds: '
                  f'{" ".join(f"{r:.1f}" for r in rounds)})  rel diff to product form {rel:.1e}', flush=True)
os.environ['GD_SPMM_D64_FORM'] = ''
