import sys, torch
sys.path.insert(0, '.')
from gnndelete_amd import ops
n, d = 524288, 128
x = torch.randn(n, d, device='cuda'); out = torch.empty(n, d, device='cuda'); w = torch.randn(d, d, device='cuda') * 0.1
for _ in range(6): ops.rows_gemm(x, None, w, trans_w=True, out=out)
torch.cuda.synchronize()
