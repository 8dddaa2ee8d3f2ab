"""Which per-block loss partial changes from run to run?  (follow-up of loss_log_determinism.py)"""
import os
import sys
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    dev = torch.device('cuda')
    for no_graph in (True, False):
        args = SimpleNamespace(workload='synth-small', gnn='gcn', df='in', df_size=5.0, seed=42, loss_type='both_layerwise', no_graph=no_graph,
                               unroll=1)
        data, model, neg, ni1, ni2 = bench.build_request(args, dev)
        state = {k: v.detach().clone() for k, v in model.state_dict().items()}
        ref = None
        for r in range(30):
            model.load_state_dict(state)
            eng = bench.make_engine(args, data, model, neg, ni1, ni2, dev, 0, 1)
            eng.step()
            torch.cuda.synchronize()
            got = {'lp1': eng._lp1.clone(), 'lp2': eng._lp2.clone(), 'hist': eng.loss_history().clone(), 'z1': eng.z1.clone(),
                   'pre1': eng.pre1.clone(), 'dh': eng.dh.clone(), 'W1': model.deletion1.deletion_weight.detach().clone()}
            if ref is None:
                ref = got
                print(f'no_graph={no_graph}: fuse_loss1={getattr(eng, "_fuse_loss1", None)} tail={getattr(eng, "_tail", None)} lp1 blocks {eng._lp1_blocks} '
                      f'S1 {eng.s1} hist {got["hist"].tolist()}')
            else:
                for k in got:
                    a, b = ref[k].nan_to_num(), got[k].nan_to_num()
                    if not torch.equal(a, b):
                        idx = (a != b).nonzero()
                        print(f'  repeat {r}: {k} differs at {idx.shape[0]} places, first {idx[0].tolist()}: {float(a[tuple(idx[0])]):.9e} vs {float(b[tuple(idx[0])]):.9e}')
            del eng


if __name__ == '__main__':
    main()
