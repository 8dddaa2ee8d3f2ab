import os, sys, torch
os.environ['GD_ROWS_GEMM_WS_MIN_ROWS'] = '1'
sys.path.insert(0, os.getcwd())
from gnndelete_amd import ops
dev = torch.device('cuda')
def avg_us(fn, reps=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / reps
for din, dout in ((128, 128), (128, 64), (64, 128)):
    w = torch.randn(dout, din, device=dev) * 0.1
    for n in (16, 4096, 16384, 65536, 236000):
        x = torch.randn(n, din, device=dev)
        out = torch.empty(n, dout, device=dev)
        t = avg_us(lambda: ops.rows_gemm(x, None, w, trans_w=True, const_w=True, out=out))
        print(f'{din}->{dout} rows {n}: {t:.1f} us')
e = torch.empty(1, device=dev)
print('empty launch (fill_):', avg_us(lambda: e.fill_(0.0)))
