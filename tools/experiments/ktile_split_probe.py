"""gd_gemm_f32 with split arithmetic on the synth-cora layer-1 shape, a few launches (for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gnndelete_amd import ops
m, k, n = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (19793, 8710, 128)))
x = torch.randn(m, k, device='cuda')
w = torch.randn(k, n, device='cuda') * 0.05
out = torch.empty(m, n, device='cuda')
for mode in (6, 0):
    ops.set_matrix_split(mode)
    for _ in range(6):
        ops.gemm_wide(x, w, out=out, const_x=True)
    torch.cuda.synchronize()
