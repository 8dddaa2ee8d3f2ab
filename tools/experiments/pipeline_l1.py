"""Layer-1 of the GCN step, two ways, on the bench graph (locality order):
   A  t1 = x W1^T ; z1 = A_hat t1 + b                      (today: two dependent full-size kernels)
   B  ax = A_hat x in row blocks on stream 1, z1[block] = ax[block] W1^T + b on stream 2 as blocks finish
      (the SpMM is bound by the gather path, the GEMM by the matrix pipe: they can share the chip)"""
import sys, os, torch
sys.path.insert(0, '.')
from gnndelete_amd.framework.synth import dcsbm_edges
from gnndelete_amd.graph import build_csr, SplitPlan
from gnndelete_amd.reorder import locality_order
from gnndelete_amd import ops
n, m, d = 235868, 1179052, 128
E, comm = dcsbm_edges(n, m, 42)
keep = torch.rand(E.shape[1]) < 0.68                      # the S_Df edge subset is ~2/3 of the training graph
E = E[:, keep]
ei = torch.cat([E, E.flip(0)], 1).cuda()
perm, inv = locality_order(ei, n); ei = inv[ei]
g = build_csr(ei.contiguous(), n, 'gcn')
x = torch.randn(n, d, device='cuda'); w = torch.randn(d, d, device='cuda') * 0.1; b = torch.randn(d, device='cuda')
t1 = torch.empty(n, d, device='cuda'); z1 = torch.empty(n, d, device='cuda'); ax = torch.empty(n, d, device='cuda'); z1b = torch.empty(n, d, device='cuda')
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
def graphed(fn):
    s_ = torch.cuda.Stream(); s_.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s_):
        fn()
    torch.cuda.current_stream().wait_stream(s_)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        fn()
    return gr.replay
def A():
    ops.rows_gemm(x, None, w, trans_w=True, out=t1)
    ops._spmm_raw(g.rowptr, g.col, g.val, t1, b, 0.0, n, g.plan, out=z1)
print(f'A (GEMM -> SpMM): eager {timeit(A):.1f} us, graph {timeit(graphed(A)):.1f} us   nnz={g.nnz}')
side = torch.cuda.Stream()
for B in (2, 4, 8, 16):
    bounds = [round(i * n / B / 32) * 32 for i in range(B)] + [n]
    plans = [SplitPlan(g.rowptr, row_range=(bounds[i], bounds[i + 1])) for i in range(B)]
    evs = [torch.cuda.Event() for _ in range(B)]
    def Bfn():
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        for i in range(B):
            ops._spmm_raw(g.rowptr, g.col, g.val, x, None, 0.0, n, plans[i], out=ax)
            evs[i].record(main)
            with torch.cuda.stream(side):
                side.wait_event(evs[i])
                lo, hi = bounds[i], bounds[i + 1]
                ops.rows_gemm(ax[lo:hi], None, w, trans_w=True, bias=b, out=z1b[lo:hi])
        main.wait_stream(side)
    t = timeit(Bfn); tg = timeit(graphed(Bfn))
    A(); Bfn(); torch.cuda.synchronize()
    print(f'B blocks={B}: eager {t:.1f} us, graph {tg:.1f} us   rel diff {float((z1b - z1).norm() / z1.norm()):.2e}')
