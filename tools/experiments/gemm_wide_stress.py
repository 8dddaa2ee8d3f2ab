"""Randomised shapes for gd_gemm_f32 (equal unit ranges per block: pieces inside one row group, across groups, whole groups
per block), with bias and gathered row subsets, against float64.  python tools/experiments/gemm_wide_stress.py [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gnndelete_amd import ops

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
g = torch.Generator().manual_seed(99)
worst = 0.0
for case in range(n_cases):
    m = int(torch.randint(1, 30000, (1,), generator=g))
    k = int(torch.randint(1, 5000, (1,), generator=g))
    n = [32, 64, 96, 128][case % 4]
    x = torch.randn(m, k, generator=g)
    w = torch.randn(k, n, generator=g) / k ** 0.5
    b = torch.randn(n, generator=g) if case % 2 else None
    idx = None
    if case % 3 == 0:
        idx = torch.randperm(m, generator=g)[:max(1, m // 2)].sort().values.to(torch.int32)
    out = torch.full((m, n), 7.0).cuda()
    ops.gemm_wide(x.cuda(), w.cuda(), b.cuda() if b is not None else None, idx=idx.cuda() if idx is not None else None, out=out)
    want = x.double().cuda() @ w.double().cuda() + (b.double().cuda() if b is not None else 0)
    rows = idx.long().cuda() if idx is not None else torch.arange(m).cuda()
    err = float((out[rows].double() - want[rows]).norm() / want[rows].norm())
    worst = max(worst, err)
    assert err < 1e-5, (case, m, k, n, err)
    if idx is not None:
        rest = torch.ones(m, dtype=torch.bool, device='cuda')
        rest[rows] = False
        assert bool((out[rest] == 7.0).all()), (case, 'rows outside idx touched')
    if case % 10 == 0:
        print(f'case {case}: M={m} K={k} N={n} ok ({err:.1e})', flush=True)
print('all', n_cases, 'cases ok; worst', worst)
