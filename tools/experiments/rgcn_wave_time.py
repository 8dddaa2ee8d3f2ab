"""Typed conv of the synth-biokg request, the three launches of a Del step (layer 1 forward 128 -> 128, layer 2 forward
128 -> 64, layer 2 transposed 64 -> 128): wave-private kernel (gd_rgcn_wave_conv_f32) vs the tile kernel, same process.
    python tools/experiments/rgcn_wave_time.py [--workload synth-biokg]"""
import argparse
import os
import sys
import time
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def avg_us(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='synth-biokg')
    a = ap.parse_args()
    from gnndelete_amd import ops
    from gnndelete_amd.graph import TypedNodeCSR
    args = SimpleNamespace(workload=a.workload, gnn='rgcn', df='in', df_size=2.5, seed=42)
    data, model, neg, ni1, ni2 = bench.build_kg_request(args)
    dev = torch.device('cuda')
    ei, et = data.edge_index[:, data.dr_mask].to(dev), data.edge_type[data.dr_mask].to(dev)
    n, R = data.num_nodes, int(et.max()) + 1
    tg = TypedNodeCSR(ei, et, n, R)
    model = model.to(dev)
    t0 = time.time()
    for tr in (False, True):
        p = tg.wave_plan(tr)
        print(f"wave plan trans={tr}: units {p['n_units']}, pieces {p['n_pieces']}, fill {p['n_pieces'] / (16.0 * p['n_units']):.2f}, "
              f"max units per tile {p['max_units']}, mean {p['n_units'] / p['n_tiles']:.1f}; edges {int(tg.fwd[3].numel())}, runs {int(tg.fwd[1].numel()) - 1}")
    torch.cuda.synchronize()
    print(f'plans built in {time.time() - t0:.1f} s')
    g = torch.Generator().manual_seed(0)
    for name, conv, din, dout, tr in (('layer 1 forward 128 -> 128', model.conv1, 128, 128, 0), ('layer 2 forward 128 -> 64', model.conv2, 128, 64, 0),
                                      ('layer 2 transposed 64 -> 128', model.conv2, 64, 128, 1)):
        x = torch.randn(n, din, generator=g).to(dev)
        w = conv.weight.detach()
        nb = conv.num_blocks or 1
        res = {}
        for form, depth in (('1', '3'), ('1', '1'), ('0', '3')):
            os.environ['GD_RGCN_WAVE'], os.environ['GD_RGCN_WAVE_DEPTH'] = form, depth
            y = torch.zeros(n, dout, device=dev)
            ops.rgcn_typed_accumulate(tg, x, w, nb, tr, y)
            res[form + depth] = (y.clone(), avg_us(lambda: ops.rgcn_typed_accumulate(tg, x, w, nb, tr, y)))
        if din == 64:                # 16-wide blocks: two blocks per wave (opt-in; the default is one)
            os.environ['GD_RGCN_WAVE'], os.environ['GD_RGCN_WAVE_BPW'] = '1', '2'
            for depth in ('3', '1'):
                os.environ['GD_RGCN_WAVE_DEPTH'] = depth
                print(f'   {name}: two blocks per wave, depth {depth}: {avg_us(lambda: ops.rgcn_typed_accumulate(tg, x, w, nb, tr, y)):.1f} us')
            os.environ.pop('GD_RGCN_WAVE_BPW')
        d = float((res['13'][0] - res['03'][0]).norm() / res['03'][0].norm())
        runs = int(tg.fwd[1].numel()) - 1
        fl = 2.0 * runs * din * dout / nb
        print(f"{name}: wave depth 3 {res['13'][1]:.1f} us ({fl / res['13'][1] / 1e6:.1f} TF), depth 1 {res['11'][1]:.1f} us, "
              f"tile {res['03'][1]:.1f} us ({fl / res['03'][1] / 1e6:.1f} TF), rel-L2 apart {d:.2e}")
    os.environ.pop('GD_RGCN_WAVE', None)
    os.environ.pop('GD_RGCN_WAVE_DEPTH', None)
    # knock-out: every source row from a 1,024-row window (L2-resident) - what the kernel costs without the fabric
    p = tg.wave_plan(False)
    keep = p['unit_edges'].clone()
    used = keep[..., 0] != n
    p['unit_edges'][..., 0] = torch.where(used, keep[..., 0] % 1024, keep[..., 0])
    x = torch.randn(n, 128, generator=g).to(dev)
    y = torch.zeros(n, 128, device=dev)
    w = model.conv1.weight.detach()
    for depth in ('3', '1'):
        os.environ['GD_RGCN_WAVE_DEPTH'] = depth
        t = avg_us(lambda: ops.rgcn_typed_accumulate(tg, x, w, 4, 0, y))
        print(f'layer 1 forward, sources folded into 1,024 rows (cache-resident), depth {depth}: {t:.1f} us')
    p['unit_edges'].copy_(keep)
    os.environ.pop('GD_RGCN_WAVE_DEPTH', None)


if __name__ == '__main__':
    main()
