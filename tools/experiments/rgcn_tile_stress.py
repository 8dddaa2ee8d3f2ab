"""Randomised stress of gd_rgcn_tile_conv_f32 against the node-major kernel: node counts around the 64-node tile size,
relation counts from 1 to 130, hubs, isolated nodes, multi-edges, all supported width / block combinations, both
directions.  python tools/experiments/rgcn_tile_stress.py [n_cases]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gnndelete_amd import ops
from gnndelete_amd.graph import TypedNodeCSR

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
g = torch.Generator().manual_seed(1234)
worst = 0.0
for case in range(n_cases):
    n = int(torch.randint(1, 700, (1,), generator=g))
    R = int(torch.randint(1, 131, (1,), generator=g))
    m = int(torch.randint(0, 40000, (1,), generator=g))
    din, dout = [(128, 128), (128, 64), (64, 128), (64, 64)][case % 4]
    nb = 4 if case % 3 else 1
    ei = torch.randint(0, n, (2, m), generator=g)
    et = torch.randint(0, R, (m,), generator=g)
    if m > 50 and case % 2:
        ei[1, :m // 3] = int(torch.randint(0, n, (1,), generator=g))          # a hub target
        et[:m // 6] = int(torch.randint(0, R, (1,), generator=g))
        ei[0, m // 3:m // 2] = int(torch.randint(0, n, (1,), generator=g))     # a hub source
    x = torch.randn(n, din, generator=g).cuda()
    w = (torch.randn(R, nb, din // nb, dout // nb, generator=g) * 0.2).cuda()
    tg = TypedNodeCSR(ei.cuda(), et.cuda(), n, R)
    for trans in (0, 1):
        xin = x if not trans else torch.randn(n, dout, generator=g).cuda()
        d_o = dout if not trans else din
        y0 = torch.randn(n, d_o, generator=g).cuda()
        y_tile, y_node = y0.clone(), y0.clone()
        os.environ.pop('GD_RGCN_NODE_MAJOR', None)
        ops.rgcn_typed_accumulate(tg, xin, w, nb, trans, y_tile)
        os.environ['GD_RGCN_NODE_MAJOR'] = '1'
        ops.rgcn_typed_accumulate(tg, xin, w, nb, trans, y_node)
        os.environ.pop('GD_RGCN_NODE_MAJOR', None)
        torch.cuda.synchronize()
        err = float((y_tile - y_node).norm() / y_node.norm().clamp(min=1e-30))
        worst = max(worst, err)
        assert err < 2e-6 and bool(torch.isfinite(y_tile).all()), (case, n, R, m, din, dout, nb, trans, err)
    if case % 10 == 0:
        print(f'case {case}: n={n} R={R} m={m} {din}->{dout} blocks={nb} ok', flush=True)
print('all', n_cases, 'cases ok; worst rel diff', worst)
