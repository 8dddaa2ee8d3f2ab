"""How far do the two ways of forming the fp32 products drift apart over a training run?  The bench request (synth-collab,
GCN, 5 % IN, trained backbone), 200 Del iterations from the same state with gd_set_matrix_split(0) and (6): Del weights,
loss history and the affected-node embeddings of the final model, relative differences.
python tools/experiments/split_trajectory.py [--steps 200]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.argv = [sys.argv[0]] + [a for a in sys.argv[1:]]
import torch

import bench
from gnndelete_amd import ops


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def main():
    steps = 200
    if '--steps' in sys.argv:
        i = sys.argv.index('--steps')
        steps = int(sys.argv[i + 1])
        del sys.argv[i:i + 2]
    args = bench.parse()
    device = torch.device('cuda', 0)
    data, model, neg, ni1, ni2 = bench.build_request(args, device)
    bench.train_backbone(model, data, device, args.pretrain_epochs)
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    out = {}
    for mode in (0, 6):
        ops.set_matrix_split(mode)
        model.load_state_dict(state)
        eng = bench.make_engine(args, data, model, neg, ni1, ni2, device, 0, 1)
        for _ in range(steps):
            eng.step()
        torch.cuda.synchronize()
        E = data.train_pos_edge_index.to(device)
        with torch.no_grad():
            z1, z2 = model(data.x.to(device), E[:, data.dr_mask.to(device)].contiguous(), return_all_emb=True)
        out[mode] = dict(w1=model.deletion1.deletion_weight.detach().clone(), w2=model.deletion2.deletion_weight.detach().clone(),
                         hist=eng.loss_history().clone(), z1=z1[data.sdf_node_1hop_mask.to(device)].clone(),
                         z2=z2[data.sdf_node_2hop_mask.to(device)].clone())
    ops.set_matrix_split(0)
    a, b = out[6], out[0]
    print(f'{steps} Del iterations, split products vs fp32 instruction (rel-L2): W_D1 {rel(a["w1"], b["w1"]):.2e}  W_D2 {rel(a["w2"], b["w2"]):.2e}  '
          f'z1[S1] {rel(a["z1"], b["z1"]):.2e}  z2[S2] {rel(a["z2"], b["z2"]):.2e}')
    h6, h0 = a['hist'], b['hist']
    print('loss history (train_loss) first / last: fp32', float(h0[0, 0]), float(h0[-1, 0]), '| split', float(h6[0, 0]), float(h6[-1, 0]),
          '| max rel diff over the run', float(((h6[:, 0] - h0[:, 0]).abs() / h0[:, 0].abs()).max()))


if __name__ == '__main__':
    main()
