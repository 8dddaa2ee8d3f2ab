"""Where does the dense 128->128 row GEMM lose time?  (a) normal, (b) every tile reads the SAME 32 rows
(all operand loads hit in cache: pure issue / MFMA pipeline), (c) different problem sizes (tail effects)."""
import sys, os, torch
sys.path.insert(0, '.')
from gnndelete_amd import ops
def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
d = 128
w = torch.randn(d, d, device='cuda') * 0.1
for n in (235868, 262144, 524288, 131072):
    x = torch.randn(n, d, device='cuda'); out = torch.empty(n, d, device='cuda')
    t = timeit(lambda: ops.rows_gemm(x, None, w, trans_w=True, out=out))
    print(f'dense n={n}: {t:.1f} us {2*n*d*d/t/1e6:.1f} TF')
    same = (torch.arange(n, device='cuda') % 32).int()
    out2 = torch.empty(32, d, device='cuda').expand(n, d) if False else out
    t = timeit(lambda: ops.rows_gemm(x, same, w, trans_w=True, out=out))
    print(f'  all tiles gather rows 0..31 (writes collide on 32 rows): {t:.1f} us {2*n*d*d/t/1e6:.1f} TF')
