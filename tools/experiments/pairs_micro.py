"""Edge-probability NI term at DBLP / Cora scale: fused pair kernel (csrc/pairs.hip) vs the dense formulation
(z z^T over the S block + masked sigmoid + MSE through autograd)."""
import sys, time, torch
sys.path.insert(0, '.')
from gnndelete_amd import ops
torch.manual_seed(0)
for n, s, d in [(17716, 6000, 64), (17716, 12000, 64), (19793, 19793, 64)]:
    z = (torch.randn(n, d, device='cuda') * 0.3).requires_grad_(True)
    nodes = torch.randperm(n, device='cuda')[:s].sort().values
    target = torch.rand(s, (s + 3) // 4 * 4, device='cuda')
    mask = torch.ones(s, target.shape[1], dtype=torch.bool, device='cuda').tril_(-1)
    target.masked_fill_(~mask, -1.0)
    count = int(mask.sum())
    n32 = nodes.int()

    def fused():
        z.grad = None
        ops.pairs_sigmoid_mse(z, n32, target, count).backward()

    def dense():
        z.grad = None
        zs = z[nodes]
        torch.nn.functional.mse_loss((zs @ zs.t())[mask[:, :s]].sigmoid(), target[:, :s][mask[:, :s]]).backward()

    for name, fn in [('fused', fused), ('dense', dense)]:
        for _ in range(2):
            fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
        flops = 2 * 2.0 * s * s * d            # two products over the full square
        print(f'S={s} d={d} {name}: {dt*1e3:.2f} ms' + (f'  ({flops/dt/1e12:.1f} TF on the two tile products, target read {s*s*4*1.0/dt/1e9:.0f} GB/s)' if name == 'fused' else ''))
    g1 = z.grad.clone(); fused(); print('  grad rel diff fused vs dense', float((z.grad - g1).norm() / g1.norm()))
