"""R-GCN layer at ogbl-biokg scale (synthetic: N=93,773, 102 typed relations incl. reverse, ~9.5 M typed edges,
128 -> 128, 4 diagonal blocks): fused typed conv (csrc/rgcn.hip) vs the materialised [R, N, d] path."""
import sys, time, torch
sys.path.insert(0, '.')
from gnndelete_amd.nn import RGCNConv
torch.manual_seed(0)
n, R, E = 93773, 102, int(sys.argv[1]) if len(sys.argv) > 1 else 9_500_000
zipf = torch.distributions.Categorical(probs=1.0 / torch.arange(1, R + 1).float())
et = zipf.sample((E,)).cuda()
ei = torch.randint(0, n, (2, E), device='cuda')
x = torch.randn(n, 128, device='cuda')
conv = RGCNConv(128, 128, R, num_blocks=4).cuda()
for name, frozen in [('fused (frozen weights)', True), ('materialised [R,N,d] + einsum', False)]:
    def run():
        if frozen:
            with torch.no_grad():
                return conv(x, ei, et)
        return conv(x, ei, et)
    y = run(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): y = run()
    torch.cuda.synchronize()
    print(f'{name}: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms, peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB')
    if frozen: y0 = y
    else: print('rel diff', float((y - y0).norm() / y.norm()))
    torch.cuda.reset_peak_memory_stats()
