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

# fused aggregate-then-transform (gd_agg_gemm_f32) against transform + aggregate, layer-1 shapes
from gnndelete_amd.graph import CappedCSR
cap = CappedCSR(g.rowptr, g.col, g.val, n)
x = torch.randn(n, 128, device=dev); w1 = torch.randn(128, 128, device=dev) / 11.3; b1 = torch.randn(128, device=dev)
xe = cap.operand(x); y = torch.empty(n, 128, device=dev); t1 = torch.empty(n, 128, device=dev)
us_sep = timed(lambda: (ops.rows_gemm(x, None, w1, trans_w=True, out=t1), ops._spmm_raw(g.rowptr, g.col, g.val, t1, b1, 0.0, n, g.plan, out=y)))
ref = y.clone()
us_fused = timed(lambda: (ops.aggregate_hubs(cap, xe), ops.agg_gemm(cap, xe, w1, bias=b1, out=y)))
us_hub = timed(lambda: ops.aggregate_hubs(cap, xe))
print(f'layer 1 (128 -> 128): transform + aggregate {us_sep:.1f} us; fused {us_fused:.1f} us (hub pre-aggregation {us_hub:.1f} us, '
      f'{cap.n_hub} hub rows); rel diff {((y - ref).norm() / ref.norm()).item():.2e}', flush=True)
y_ext = torch.empty(n + g.plan.n_slots, 128, device=dev)
us_items = timed(lambda: ops.agg_gemm_items(g, x, w1, b1, y_ext))
print(f'layer 1 fused, work-item form (pieces + fix-up, no operand copy): {us_items:.1f} us; rel diff {((y_ext[:n] - ref).norm() / ref.norm()).item():.2e}', flush=True)
# backward shape: (A^T dp2)[S1] W2 gated, d_in = 64 -> 128 on the S1 rows
capt = CappedCSR(g.rowptr_t, g.col_t, g.val_t, n)
dp = torch.randn(n, 64, device=dev); w2 = torch.randn(64, 128, device=dev) / 8; idx1 = eng.idx1
bits = torch.randint(-2**31, 2**31 - 1, (idx1.numel(), 4), device=dev, dtype=torch.int32)
dpe = capt.operand(dp); dh = torch.empty(n, 128, device=dev); dt = torch.empty(n, 64, device=dev)
dhc = torch.empty(idx1.numel(), 128, device=dev)
us_sep = timed(lambda: (ops._spmm_raw(g.rowptr_t, g.col_t, g.val_t, dp, None, 0.0, n, g.plan_t, out=dt),
                        ops.rows_gemm(dt, idx1, w2, gate_bits=bits, out=dh)))
us_fused = timed(lambda: (dpe[:n].copy_(dp), ops.aggregate_hubs(capt, dpe), ops.agg_gemm(capt, dpe, w2, rows=idx1, gate_bits=bits, out=dh, w_out_in=False)))
print(f'layer-2 input gradient (64 -> 128 on {idx1.numel()} rows): aggregate + gated transform {us_sep:.1f} us; fused incl. operand copy {us_fused:.1f} us', flush=True)
