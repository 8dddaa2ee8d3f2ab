"""Three-way parity over the reference's real horizon (framework/training_args.py:129-132: 600 epochs for ogbl-* graphs
with gnndelete): the fp64 oracle, the fp32 oracle and the HIP engine train the SAME request from the SAME state with
the SAME negatives.  Two correct fp32 implementations of this training map drift apart over hundreds of epochs (the ReLU
between the layers gates the layer-2 gradient with [z1 > 0]; an entry of z1 within summation-order noise of zero lands
on different sides, and Adam's g / sqrt(v) carries the difference forward) - so the meaningful statement is not "HIP ==
fp32 oracle to 1e-4 after 600 epochs" but "HIP is as close to the fp64 trajectory as the fp32 oracle is", plus the
task-level contract of north_star: post-deletion AUC within +-0.002.

  distances = rel-L2 to the fp64 run of (W_D1, W_D2, z1 on the 1-hop S_Df nodes, z2 on the 2-hop S_Df nodes), embeddings
  taken on the retained edges (evaluation semantics, framework/trainer/base.py:238-242)
  assert  d(HIP, fp64) <= RATIO * max over the fp32 ENSEMBLE of d(member, fp64) + (max - min over the ensemble)   for every
          quantity, at epochs 100 / 300 / 600.  The ensemble = the fp32 oracle as is + the same oracle with its edge lists
          permuted (a different summation order in every scatter: another correct fp32 implementation) - at synth-small as
          CPU runs (index_add_ adds in edge order there, so a permutation really is another association; torch's GPU scatter
          sums a 3,000-node graph in the same order whatever the permutation - four GPU members had identical digits,
          ADVICE r4), plus one GPU member.  The allowance is the ensemble's OWN measured spread, not a constant: members that
          have and have not passed their first gate flip at a check are (max - min) apart.  One member is not enough: the distance
          grows in JUMPS (the first ReLU flip takes it from 3e-7 to 2e-4 in one epoch) and WHICH run flips first is chance -
          with one-row SpMM items the fp32 oracle had flipped by epoch 100 and HIP had not (1.9e-4 vs 3.4e-7), with two-row
          items (another association of the same sums) it was the other way round (3.8e-7 vs 2.2e-4); both are in
          profiles/r03_long_parity.txt.  RATIO = 2: every run is one sample of a chaotic map.
  assert  |AUC(HIP) - AUC(fp64)| <= 0.002 and |AUC(HIP) - AUC(fp32 oracle)| <= 0.002 + |AUC(fp32 oracle) - AUC(fp64)|
          (test edges; Df vs Dr)

All oracles run as plain torch ops on the GPU (fp64: fast fp64 units; 600 CPU epochs at the bench's size would take 15
minutes); at synth-small the fp32 ensemble is four CPU runs + one GPU run, at the bench's size three GPU runs (its scatters do
sum in different orders there: the members differ in every digit); GCN at both sizes, GAT at synth-small."""
from types import SimpleNamespace

import pytest
import torch

from helpers import oracle_runner, rel_l2

pytestmark = pytest.mark.gpu

EPOCHS, CHECK = 600, (100, 300, 600)
RATIO = 2.0                   # measured ratios: profiles/r03_long_parity.txt, profiles/r04_long_parity.txt
# fp32 ensemble: (device, edge-order seed; None = as given).  At synth-small the CPU members are the ones whose association really
# differs (a 600-epoch CPU run takes seconds there); at the bench's size an oracle run costs 20 s of the suite's time limit (the
# full-size test adds the CPU oracle + three more GPU members at its 20-iteration horizon)
# (four CPU members: with six the two synth-small cases took 50 s more of a suite that is asked to stay near 800 s, and the record of the
#  seven-member runs - profiles/r05_long_parity.txt - has its extremes among the first five; running the CPU members in threads was
#  tried and is slower: 136 / 160 s against 81 / 102.  Round 6: they run as CHILD PROCESSES (tests/parity_member.py), started
#  before the GPU runs of the test and collected after them - same members, same seeds, same arithmetic)
PERMS = {'synth-small': (('cuda', None), ('cpu', None), ('cpu', 1), ('cpu', 2), ('cpu', 3)),
         # (round 6: two GPU members at the bench's size, 22 s of the suite's limit each - in every record of rounds 3-5 HIP is
         #  2.5-3 x CLOSER to the fp64 run than either member at all three checks, profiles/r05_long_parity.txt)
         'synth-collab': (('cuda', None), ('cuda', 1))}


def _auc(z, pos, neg):
    from gnndelete_amd.framework.metrics import batched_roc_auc
    ei = torch.cat([pos, neg], 1).to(z.device)
    score = (z[ei[0]].double() * z[ei[1]].double()).sum(-1).sigmoid()
    label = torch.cat([torch.ones(pos.shape[1]), torch.zeros(neg.shape[1])]).to(z.device)
    return float(batched_roc_auc(score.float(), label)[0])


LT, ALPHA, LR = 'both_layerwise', 0.5, 1e-3
_PREPARED = {}


def prepare(workload, gnn, queue=None):
    """The request of one case - seeded build, a backbone with signal trained on the GPU (set-up) - and, for the cases with CPU
    ensemble members, those members as child processes (tests/parity_member.py): started here, one process each, when the test
    runs on its own; with `queue` (a list) only DESCRIBED - (request file, edge-order seed, out file) appended to it - for the
    session-start background worker of tests/oracle_jobs.py, which trains them one after the other while other tests use the
    GPU (tests/conftest.py).  Cached: the test picks the prepared case up."""
    key = (workload, gnn)
    if key in _PREPARED:
        return _PREPARED[key]
    import os
    import subprocess
    import sys
    import tempfile
    import bench
    dev = torch.device('cuda')
    args = SimpleNamespace(workload=workload, gnn=gnn, df='in', df_size=5.0, seed=42)
    data, model, neg, ni1, ni2 = bench.build_request(args, dev)
    bench.train_backbone(model, data, dev, 30)                      # a backbone with signal (set-up)
    state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    children, tmp = [], None
    cpu_members = [perm for where, perm in PERMS[workload] if where == 'cpu']
    if cpu_members:
        tmp = tempfile.TemporaryDirectory()
        req = os.path.join(tmp.name, 'request.pt')
        torch.save(dict(gnn=gnn, data={k: (v.cpu() if torch.is_tensor(v) else v) for k, v in data.items()}, state=state, neg=neg.cpu(),
                        ni1=ni1.cpu(), ni2=ni2.cpu(), loss_type=LT, alpha=ALPHA, lr=LR, check=CHECK), req)
        here = os.path.dirname(os.path.abspath(__file__))
        for perm in cpu_members:
            out_ = os.path.join(tmp.name, f'member_{perm}.pt')
            if queue is not None:
                queue.append((req, perm, out_))
                children.append((perm, out_, None))
                continue
            # (a run of this test alone: four children of 4 threads - the boxes run under a 16-CPU quota)
            env = dict(os.environ, OMP_NUM_THREADS='4', HIP_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES='')
            children.append((perm, out_, subprocess.Popen([sys.executable, os.path.join(here, 'parity_member.py'), req,
                                                           'none' if perm is None else str(perm), out_], env=env,
                                                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    _PREPARED[key] = dict(data=data, model=model, neg=neg, ni1=ni1, ni2=ni2, state=state, children=children, tmp=tmp)
    return _PREPARED[key]


@pytest.mark.parametrize('workload,gnn', [('synth-small', 'gcn'), ('synth-collab', 'gcn'), ('synth-small', 'gat')])
def test_hip_tracks_the_fp64_trajectory_as_closely_as_the_fp32_oracle(workload, gnn):
    import os
    from gnndelete_amd.engine import NodeembEngine
    dev = torch.device('cuda')
    case = prepare(workload, gnn)
    _PREPARED.pop((workload, gnn))                                   # (a case is consumed once: its children are collected below)
    data, model, neg, ni1, ni2, state = (case[k] for k in ('data', 'model', 'neg', 'ni1', 'ni2', 'state'))
    children, tmp = case['children'], case['tmp']
    m1, m2 = data.sdf_node_1hop_mask, data.sdf_node_2hop_mask
    E = data.train_pos_edge_index
    e_dr, e_sdf, pos = E[:, data.dr_mask], E[:, data.sdf_mask], E[:, data.df_mask]
    lt, alpha, lr = LT, ALPHA, LR
    import gc
    names = ('W_D1', 'W_D2', 'z1[S1]', 'z2[S2]')

    def run_oracle(dtype, perm, where='cuda'):
        """One oracle to the end (torch ops on the GPU: fp64 has fast units there), one at a time - an oracle's autograd
        tapes at the bench's size are tens of GB.  -> snapshots at the CHECK epochs (+ its original embeddings)."""
        step, snap, z_ori = oracle_runner(gnn, data, state, neg, ni1, ni2, dtype, torch.device(where), lt, alpha, lr, perm=perm)
        snaps, done_ = [], 0
        for upto in CHECK:
            for _ in range(upto - done_):
                step()
            done_ = upto
            snaps.append(snap())
        z_ori = tuple(z.detach().float().clone() for z in z_ori)
        del step, snap
        gc.collect()
        torch.cuda.empty_cache()
        return snaps, z_ori
    # (the CPU members of the ensemble are child processes started by prepare(): they train while the GPU runs below go on)
    s64_all, _ = run_oracle(torch.float64, None)
    d_members, s32_last, z_ori32 = [], None, None
    for where, perm in PERMS[workload]:       # the fp32 ensemble: distances to the fp64 run at every check
        if where == 'cpu':
            continue
        snaps, z_ori = run_oracle(torch.float32, perm, where)
        d_members.append([[rel_l2(sn[i], s64[i]) for i in range(4)] for sn, s64 in zip(snaps, s64_all)])
        if perm is None and where == 'cuda':
            s32_last, z_ori32 = snaps[-1], z_ori
        del snaps
    for perm, out_, proc in children:
        if proc is None:                         # trained by the session-start worker (or here, should that worker have died)
            import oracle_jobs
            import parity_member
            if not oracle_jobs.wait_for_file(out_):
                parity_member.run(os.path.join(os.path.dirname(out_), 'request.pt'), perm, out_, threads=8)
        else:
            log_, _ = proc.communicate(timeout=900)
            assert proc.returncode == 0, f'CPU ensemble member (edge-order seed {perm}) failed:\n{log_[-2000:]}'
        snaps = torch.load(out_, weights_only=False)
        d_members.append([[rel_l2(sn[i], s64[i]) for i in range(4)] for sn, s64 in zip(snaps, s64_all)])
    if tmp is not None:
        tmp.cleanup()
    z1o, z2o = z_ori32
    model.load_state_dict(state)
    hip = model.to(dev)
    xg, edg = data.x.to(dev), e_dr.to(dev).contiguous()
    eng = NodeembEngine(hip, xg, e_sdf.to(dev).contiguous(), z1o.to(dev), z2o.to(dev), pos.to(dev), neg.to(dev), ni1, ni2,
                        loss_type=lt, alpha=alpha, lr=lr)

    def snap_hip():
        with torch.no_grad():
            z1, z2 = hip(xg, edg, return_all_emb=True)
        return (hip.deletion1.deletion_weight.detach().double().cpu(), hip.deletion2.deletion_weight.detach().double().cpu(),
                z1[m1.to(dev)].double().cpu(), z2[m2.to(dev)].double().cpu(), z2.detach())
    done = 0
    for c, upto in enumerate(CHECK):
        eng.run(upto - done)
        done = upto
        torch.cuda.synchronize()
        s64, sh = s64_all[c], snap_hip()
        d_ens = [dm[c] for dm in d_members]
        d32 = [max(d[i] for d in d_ens) for i in range(4)]
        spread = [d32[i] - min(d[i] for d in d_ens) for i in range(4)]
        dh = [rel_l2(sh[i], s64[i]) for i in range(4)]
        print(f'[{workload} {gnn}] epoch {upto}: ' + ', '.join(
            f'{n} fp32 ' + ' '.join(f'{d[i]:.2e}' for d in d_ens) + f' / HIP {dh[i]:.2e}' for i, n in enumerate(names)))
        for n, a_, sp_, b_ in zip(names, d32, spread, dh):
            assert b_ <= RATIO * a_ + sp_, (workload, gnn, upto, n, 'fp32 ensemble max', a_, 'spread', sp_, 'HIP', b_)
    s32 = s32_last
    tp, tn = data.test_pos_edge_index, data.test_neg_edge_index
    k = data.directed_df_edge_index.shape[1]
    gen = torch.Generator().manual_seed(0)
    dr_sub = e_dr[:, torch.randperm(e_dr.shape[1], generator=gen)[:k]]
    aucs = {}
    for name, snap in (('fp64', s64), ('fp32', s32), ('hip', sh)):
        z2 = snap[4]
        aucs[name] = (_auc(z2, tp, tn), _auc(z2, dr_sub, data.directed_df_edge_index))
    print(f'[{workload} {gnn}] AUC (test edges, Df vs Dr): ' + ', '.join(f'{k_} {v[0]:.6f} / {v[1]:.6f}' for k_, v in aucs.items()))
    # north_star's +-0.002 against the exact (fp64) trajectory; against the fp32 oracle the same bound widened by that
    # oracle's OWN distance to the fp64 run (after 600 epochs at synth-small it is 0.0017 off in the Df-vs-Dr AUC while HIP
    # is 0.0003 off: profiles/r04_long_parity.txt)
    for q in (0, 1):
        assert abs(aucs['hip'][q] - aucs['fp64'][q]) <= 0.002, aucs
        assert abs(aucs['hip'][q] - aucs['fp32'][q]) <= 0.002 + abs(aucs['fp32'][q] - aucs['fp64'][q]), aucs
