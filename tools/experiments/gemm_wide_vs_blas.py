"""gd_gemm_f32 (K-tiled MFMA kernel) against torch.mm (rocBLAS / hipBLASLt) on the bag-of-words layer-1 shapes, same
process, alternating, HIP events on the current stream.  python tools/experiments/gemm_wide_vs_blas.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gnndelete_amd import ops

def timeit(fn, reps=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3

dev = torch.device('cuda:0')
for name, m, k, n in [('synth-dblp', 17716, 1639, 128), ('synth-cora', 19793, 8710, 128), ('cora', 2708, 1433, 128),
                      ('synth-dblp->64', 17716, 1639, 64)]:
    x = torch.randn(m, k, device=dev)
    w = torch.randn(k, n, device=dev) * 0.05
    wt = w.t().contiguous()
    out = torch.empty(m, n, device=dev)
    ref = x.double() @ w.double()
    y = ops.gemm_wide(x, w, out=out, const_x=True)
    err = ((y.double() - ref).norm() / ref.norm()).item()
    err_b = (((x @ w).double() - ref).norm() / ref.norm()).item()
    flop = 2.0 * m * k * n
    for rnd in range(2):
        t_blas = timeit(lambda: torch.mm(x, w, out=out))
        t_lin = timeit(lambda: torch.nn.functional.linear(x, wt))
        t_gd = timeit(lambda: ops.gemm_wide(x, w, out=out, const_x=True))
        ops.set_matrix_split(6)                         # opt-in: fp32 products from six exact bf16 partial products
        ys = ops.gemm_wide(x, w, out=torch.empty_like(out), const_x=True)
        err_s = ((ys.double() - ref).norm() / ref.norm()).item()
        t_sp = timeit(lambda: ops.gemm_wide(x, w, out=out, const_x=True))
        ops.set_matrix_split(0)
        print(f'{name:16s}   split arithmetic: gd_gemm_f32 {t_sp*1e6:7.1f} us {flop/t_sp/1e12:5.1f} TF fp32-equivalent | rel err {err_s:.1e}', flush=True)
        print(f'{name:16s} M={m} K={k} N={n}: torch.mm {t_blas*1e6:7.1f} us {flop/t_blas/1e12:5.1f} TF | F.linear {t_lin*1e6:7.1f} us '
              f'{flop/t_lin/1e12:5.1f} TF | gd_gemm_f32 {t_gd*1e6:7.1f} us {flop/t_gd/1e12:5.1f} TF | rel err gd {err:.1e} blas {err_b:.1e}', flush=True)
