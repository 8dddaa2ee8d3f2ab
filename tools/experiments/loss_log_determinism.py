"""Is the logged loss of the single-GPU engine bit-reproducible from run to run?  (The two-rank self-test of bench.py saw the logged
loss of a step differ by one ulp between two engines running identical kernels.)  Builds the same small request R times per model,
runs 4 steps each, compares Del weights and the loss history bit for bit against the first run.
    python tools/experiments/loss_log_determinism.py [reps]"""
import os
import sys
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    dev = torch.device('cuda')
    for gnn in ('sage', 'gcn', 'gat'):
        args = SimpleNamespace(workload='synth-small', gnn=gnn, df='in', df_size=5.0, seed=42, loss_type='both_layerwise', no_graph=False,
                               unroll=1)
        data, model, neg, ni1, ni2 = bench.build_request(args, dev)
        state = {k: v.detach().clone() for k, v in model.state_dict().items()}
        ref, bad = None, []
        for r in range(reps):
            model.load_state_dict(state)
            eng = bench.make_engine(args, data, model, neg, ni1, ni2, dev, 0, 1)
            for _ in range(4):
                eng.step()
            torch.cuda.synchronize()
            got = (model.deletion1.deletion_weight.detach().clone(), model.deletion2.deletion_weight.detach().clone(),
                   eng.loss_history().clone())
            if ref is None:
                ref = got
            elif not all(torch.equal(a.nan_to_num(), b.nan_to_num()) for a, b in zip(ref, got)):
                which = [n for n, a, b in zip(('W_D1', 'W_D2', 'log'), ref, got) if not torch.equal(a.nan_to_num(), b.nan_to_num())]
                idx = (ref[2].nan_to_num() != got[2].nan_to_num()).nonzero().tolist()
                bad.append((r, which, idx[:4]))
            del eng
        print(f'{gnn}: {len(bad)} of {reps - 1} repeats differ from the first run', bad[:5])


if __name__ == '__main__':
    main()
