import sys, os, torch
sys.path.insert(0, '.')
from gnndelete_amd import ops
n = 235868
d_in, d_out = int(os.environ.get('DIN', 128)), int(os.environ.get('DOUT', 128))
x = torch.randn(n, d_in, device='cuda'); w = torch.randn(d_out, d_in, device='cuda') * 0.1
out = torch.empty(n, d_out, device='cuda')
idx = None
if os.environ.get('IDX'):
    idx = (torch.rand(n, device='cuda') < 0.76).nonzero().flatten().int()
for _ in range(3): ops.rows_gemm(x, idx, w, trans_w=True, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): ops.rows_gemm(x, idx, w, trans_w=True, out=out)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 10 * 1e3
rows = n if idx is None else idx.numel()
print(f'rows_gemm {rows}x{d_in}->{d_out}: {t:.1f} us, {2*rows*d_in*d_out/t/1e6:.1f} TF, {(rows*(d_in+d_out)*4)/t/1e6:.2f} TB/s')
ref = torch.nn.functional.linear(x, w)
e0.record()
for _ in range(10): torch.nn.functional.linear(x, w)
e1.record(); torch.cuda.synchronize()
print(f'rocblas: {e0.elapsed_time(e1)/10*1e3:.1f} us')
