"""How much of the typed conv's gathered traffic a node order can remove (synth-biokg, layer-1 launch, wave-private kernel):
ids as generated / the engine's label-propagation order / the generator's PLANTED communities made contiguous (an upper bound for
any clustering on this graph).  Prints the launch time per order.
    python tools/experiments/rgcn_order_study.py"""
import os
import sys
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from rgcn_wave_time import avg_us  # noqa: E402


def main():
    from gnndelete_amd import ops
    from gnndelete_amd.framework.synth import KG_SHAPES, dcsbm_edges
    from gnndelete_amd.graph import TypedNodeCSR
    from gnndelete_amd.reorder import locality_order
    args = SimpleNamespace(workload='synth-biokg', gnn='rgcn', df='in', df_size=2.5, seed=42)
    data, model, neg, ni1, ni2 = bench.build_kg_request(args)
    dev = torch.device('cuda')
    ei, et = data.edge_index[:, data.dr_mask].to(dev), data.edge_type[data.dr_mask].to(dev)
    n, R = data.num_nodes, int(et.max()) + 1
    nn_, _, m = KG_SHAPES['synth-biokg']
    _, community = dcsbm_edges(nn_, m, 42, comm_size=64, p_in=0.6)
    model = model.to(dev)
    w = model.conv1.weight.detach()
    g = torch.Generator().manual_seed(0)
    x0 = torch.randn(n, 128, generator=g).to(dev)
    orders = {'as generated': None}
    perm, inv = locality_order(ei, n)
    orders['label propagation (engine)'] = inv
    pc = torch.argsort(community.to(dev) * n + torch.arange(n, device=dev))
    invc = torch.empty_like(pc)
    invc[pc] = torch.arange(n, device=dev)
    orders['planted communities contiguous'] = invc
    for name, inv_ in orders.items():
        e2 = ei if inv_ is None else inv_[ei]
        tg = TypedNodeCSR(e2, et, n, R)
        x = x0 if inv_ is None else x0[torch.argsort(inv_)]
        y = torch.zeros(n, 128, device=dev)
        t = avg_us(lambda: ops.rgcn_typed_accumulate(tg, x, w, 4, 0, y))
        p = tg.wave_plan(False)
        print(f'{name}: layer-1 launch {t:.1f} us, units {p["n_units"]}, fill {p["n_pieces"] / (16.0 * p["n_units"]):.2f}')


if __name__ == '__main__':
    main()
