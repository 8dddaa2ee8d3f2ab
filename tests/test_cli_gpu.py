"""End-to-end run of the CLI scripts on the GPU (prepare -> train original -> unlearn -> test),
and the HIP trainer's eval against the values the reference's Trainer.eval produced (golden)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from helpers import hip_model, load_golden, rel_l2, split_fixture, t, free_port

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(cmd, cwd):
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable] + cmd, cwd=cwd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


@pytest.fixture(scope='module')
def prepared(tmp_path_factory):
    """prepare_dataset.py run ONCE per stand-in for this module (it is the same seeded command in every pipeline test; a
    subprocess start costs ~3 s of the suite's time limit): -> copy(cwd, dataset) that puts its output under a test's cwd."""
    import shutil
    base = str(tmp_path_factory.mktemp('prepared'))
    done = set()

    def copy(cwd, dataset):
        if dataset not in done:
            run([os.path.join(ROOT, 'prepare_dataset.py'), '--dataset', dataset, '--seeds', '42'], base)
            done.add(dataset)
        shutil.copytree(os.path.join(base, 'data', dataset), os.path.join(cwd, 'data', dataset))
    return copy


def test_trainer_eval_matches_reference_golden(tmp_path):
    from types import SimpleNamespace
    from gnndelete_amd.framework.data import Data
    from gnndelete_amd.framework.trainer.gnndelete_nodeemb import GNNDeleteNodeembTrainer
    fx = load_golden('eval.npz')
    state, data, rest = split_fixture(fx)
    m = hip_model('gat', state, data['sdf_node_1hop_mask'], data['sdf_node_2hop_mask'])
    args = SimpleNamespace(unlearning_model='gnndelete_nodeemb', dataset='Cora', checkpoint_dir=str(tmp_path),
                           eval_on_cpu=False)
    tr = GNNDeleteNodeembTrainer(args)
    torch.manual_seed(int(rest['eval_seed']))
    loss, dt_auc, dt_aup, df_auc, df_aup, df_logit, _, log = tr.eval(m, Data(data), 'val')
    assert torch.equal(torch.stack(tr.df_pos_edge), t(rest['df_pos_masks']))
    assert abs(loss - float(rest['val_loss'])) < 1e-5
    assert abs(dt_auc - float(rest['val_dt_auc'])) < 2e-3 and abs(df_auc - float(rest['val_df_auc'])) < 2e-3
    np.testing.assert_allclose(np.array(df_logit), rest['val_df_logit'], rtol=1e-4)
    torch.save({'model_state': m.state_dict()}, os.path.join(str(tmp_path), 'model_best.pt'))
    out = tr.test(m, Data(data))
    assert abs(out[1] - float(rest['test_dt_auc'])) < 2e-3 and abs(out[3] - float(rest['test_df_auc'])) < 2e-3
    assert abs(tr.trainer_log['auc_sum'] - float(rest['test_auc_sum'])) < 4e-3
    tr.save_log()
    assert os.path.exists(os.path.join(str(tmp_path), 'trainer_log.json'))


@pytest.mark.parametrize('gnn,method,loss_type', [('gcn', 'gnndelete_nodeemb', 'both_layerwise'),
                                                  ('gat', 'gnndelete_nodeemb', 'both_all'),
                                                  ('gin', 'gnndelete', 'both_layerwise'),
                                                  ('sage', 'gnndelete_nodeemb', 'both_layerwise')])
def test_cli_pipeline(tmp_path, prepared, gnn, method, loss_type):
    cwd = str(tmp_path)
    prepared(cwd, 'synth-tiny')
    common = ['--dataset', 'synth-tiny', '--gnn', gnn, '--random_seed', '42']
    run([os.path.join(ROOT, 'train_gnn.py')] + common + ['--epochs', '30', '--valid_freq', '10'], cwd)
    orig = os.path.join(cwd, 'checkpoint', 'synth-tiny', gnn, 'original', '42')
    # upstream forces 2000 epochs for the original model; the files must exist either way
    assert os.path.exists(os.path.join(orig, 'model_best.pt')) and os.path.exists(os.path.join(orig, 'pred_proba.pt'))
    run([os.path.join(ROOT, 'delete_gnn.py')] + common + ['--unlearning_model', method, '--df', 'in', '--df_size', '5',
                                                          '--epochs', '40', '--valid_freq', '20', '--loss_type', loss_type], cwd)
    if 'nodeemb' in method:
        out = os.path.join(cwd, 'checkpoint', 'synth-tiny', gnn, method, f'mse_mean-{loss_type}-0.5-non_connected',
                           'in-5.0-42')
    else:
        out = os.path.join(cwd, 'checkpoint', 'synth-tiny', gnn, method, f'mse_mean-{loss_type}-0.5-non_connected',
                           'in-5.0-42')
    with open(os.path.join(out, 'trainer_log.json')) as f:
        log = json.load(f)
    for key in ['dt_auc', 'df_auc', 'auc_sum', 'auc_gap', 'df_logit', 'log']:
        assert key in log, key
    assert 0.0 <= log['dt_auc'] <= 1.0 and len(log['log']) >= 2
    for name in ['model_best.pt', 'model_final.pt', 'training_args.json', 'pred_proba.pt']:
        assert os.path.exists(os.path.join(out, name)), name
    state = torch.load(os.path.join(out, 'model_final.pt'))['model_state']
    assert not torch.allclose(state['deletion2.deletion_weight'], torch.full_like(state['deletion2.deletion_weight'], 1e-3))


@pytest.mark.parametrize('gnn', ['rgcn', 'rgat'])
def test_cli_pipeline_knowledge_graph(tmp_path, monkeypatch, prepared, gnn):
    """R-GCN / R-GAT on a synthetic KG: original training, then Del training on random-walk batches."""
    cwd = str(tmp_path)
    monkeypatch.setenv('GNNDELETE_FORCE_EPOCHS', '3')
    monkeypatch.setenv('GNNDELETE_FORCE_VALID_FREQ', '3')
    monkeypatch.setenv('GNNDELETE_FORCE_NUM_STEPS', '4')
    prepared(cwd, 'synth-kg-tiny')
    common = ['--dataset', 'synth-kg-tiny', '--gnn', gnn, '--random_seed', '42', '--in_dim', '32', '--hidden_dim', '32',
              '--out_dim', '16']
    run([os.path.join(ROOT, 'train_gnn.py')] + common, cwd)
    run([os.path.join(ROOT, 'delete_gnn.py')] + common + ['--unlearning_model', 'gnndelete', '--df', 'in', '--df_size', '5'], cwd)
    out = os.path.join(cwd, 'checkpoint', 'synth-kg-tiny', gnn, 'gnndelete', 'mse_mean-both_layerwise-0.5-non_connected',
                       'in-5.0-42')
    with open(os.path.join(out, 'trainer_log.json')) as f:
        log = json.load(f)
    assert 0.0 <= log['dt_auc'] <= 1.0 and 0.0 <= log['df_auc'] <= 1.0
    state = torch.load(os.path.join(out, 'model_final.pt'))['model_state']
    assert 'node_emb.weight' in state and 'W' in state and state['conv1.weight'].shape[0] == 8


def test_cli_knowledge_graph_fullgraph_fused_step(tmp_path, monkeypatch, prepared):
    """delete_gnn.py --gnn rgcn --fullgraph: the fused R-GCN engine behind the KG trainer (one hipGraph per epoch on the
    whole Dr graph) - runs end to end, lowers its loss and writes the reference's checkpoint layout (the engine itself is
    checked against the oracle in tests/test_engine_gpu.py::test_rgcn_engine_matches_oracle_training)."""
    cwd = str(tmp_path)
    monkeypatch.setenv('GNNDELETE_FORCE_EPOCHS', '6')
    monkeypatch.setenv('GNNDELETE_FORCE_VALID_FREQ', '3')
    prepared(cwd, 'synth-kg-tiny')
    common = ['--dataset', 'synth-kg-tiny', '--gnn', 'rgcn', '--random_seed', '42', '--in_dim', '32', '--hidden_dim', '32',
              '--out_dim', '16']
    run([os.path.join(ROOT, 'train_gnn.py')] + common, cwd)
    run([os.path.join(ROOT, 'delete_gnn.py')] + common + ['--unlearning_model', 'gnndelete', '--df', 'in', '--df_size', '5',
                                                          '--fullgraph'], cwd)
    out = os.path.join(cwd, 'checkpoint', 'synth-kg-tiny', 'rgcn', 'gnndelete', 'mse_mean-both_layerwise-0.5-non_connected',
                       'in-5.0-42')
    with open(os.path.join(out, 'trainer_log.json')) as f:
        log = json.load(f)
    hist = np.array(log['loss_history'])
    assert hist.shape[0] == 6 and np.isfinite(hist).all() and hist[-1, 0] < hist[0, 0]
    assert 0.0 <= log['dt_auc'] <= 1.0 and 0.0 <= log['df_auc'] <= 1.0
    state = torch.load(os.path.join(out, 'model_final.pt'))['model_state']
    assert 'deletion1.deletion_weight' in state and 'node_emb.weight' in state


@pytest.mark.parametrize('gnn', ['gat', 'gcn'])
def test_cli_node_deletion(tmp_path, monkeypatch, gnn):
    """delete_node.py: node unlearning with accuracy / F1 evaluation (out_dim = #classes = 4: the engine pads layer 2 with
    zero columns to the width of its fused forms, engine._padded_out_shadow; the checkpoint keeps the 4 x 4 W_D2)."""
    cwd = str(tmp_path)
    monkeypatch.setenv('GNNDELETE_FORCE_EPOCHS', '20')
    monkeypatch.setenv('GNNDELETE_FORCE_VALID_FREQ', '10')
    common = ['--dataset', 'synth-tiny', '--gnn', gnn, '--random_seed', '42']
    run([os.path.join(ROOT, 'train_node.py')] + common, cwd)
    run([os.path.join(ROOT, 'delete_node.py')] + common + ['--unlearning_model', 'gnndelete_nodeemb', '--df', 'in',
                                                           '--df_size', '5'], cwd)
    out = os.path.join(cwd, 'checkpoint_node', 'synth-tiny', gnn, 'gnndelete_nodeemb-node_deletion',
                       'mse_mean-both_layerwise-0.5-non_connected', 'in-5.0-42')
    with open(os.path.join(out, 'trainer_log.json')) as f:
        log = json.load(f)
    assert 0.0 <= log['dt_acc'] <= 1.0 and 'dt_f1' in log
    state = torch.load(os.path.join(out, 'model_final.pt'))['model_state']
    assert state['deletion2.deletion_weight'].shape == (4, 4)
    assert not torch.allclose(state['deletion1.deletion_weight'], torch.full((128, 128), 1e-3))
    if gnn == 'gat':
        # delete_node_feature.py: same flow, feature rows of the Df nodes zeroed, own checkpoint root
        run([os.path.join(ROOT, 'delete_node_feature.py')] + common + ['--unlearning_model', 'gnndelete_nodeemb', '--df',
                                                                       'in', '--df_size', '5'], cwd)
        out = os.path.join(cwd, 'checkpoint_node_feature', 'synth-tiny', gnn, 'gnndelete_nodeemb-node_deletion',
                           'mse_mean-both_layerwise-0.5-non_connected', 'in-5.0-42')
        with open(os.path.join(out, 'trainer_log.json')) as f:
            assert 'dt_acc' in json.load(f)


def test_minibatch_trainer_runs(tmp_path, monkeypatch):
    """--minibatch: the GraphSAINT-style loop of gnndelete_nodeemb.py:352-495 (per-batch CSR,
    per-batch mask overrides, autograd path)."""
    cwd = str(tmp_path)
    monkeypatch.setenv('GNNDELETE_FORCE_EPOCHS', '2')
    monkeypatch.setenv('GNNDELETE_FORCE_VALID_FREQ', '2')
    monkeypatch.setenv('GNNDELETE_FORCE_NUM_STEPS', '3')
    data_dir = os.path.join(cwd, 'data', 'ogbl-synth')
    os.makedirs(data_dir)
    from gnndelete_amd.framework.synth import make_linkpred_dataset
    data, df = make_linkpred_dataset(None, seed=42, shape=(800, 32, 4000, 'dense'))
    data.save(os.path.join(data_dir, 'd_42.pt'))
    torch.save(df, os.path.join(data_dir, 'df_42.pt'))
    common = ['--dataset', 'ogbl-synth', '--gnn', 'gcn', '--random_seed', '42', '--batch_size', '200']
    run([os.path.join(ROOT, 'train_gnn.py')] + common, cwd)
    run([os.path.join(ROOT, 'delete_gnn.py')] + common + ['--unlearning_model', 'gnndelete_nodeemb', '--df', 'in',
                                                          '--df_size', '5', '--minibatch'], cwd)
    out = os.path.join(cwd, 'checkpoint', 'ogbl-synth', 'gcn', 'gnndelete_nodeemb', 'mse_mean-both_layerwise-0.5-non_connected',
                       'in-5.0-42')
    with open(os.path.join(out, 'trainer_log.json')) as f:
        log = json.load(f)
    assert 0.0 <= log['dt_auc'] <= 1.0


def test_minibatch_trainer_is_data_parallel_under_torch_distributed(tmp_path):
    """SURVEY 8e, option 1: `torch.distributed.run --nproc-per-node 2 delete_gnn.py ... --minibatch` - every rank draws
    its own GraphSAINT batches on the device, the Del-weight gradients are all-reduced (mean) before each optimizer step
    (gloo here: both ranks share the box's one GPU; RCCL on the node).  The ranks must end with IDENTICAL Del weights
    (same initial state, same averaged gradients) that differ from a single-process run (other batches)."""
    import subprocess
    import sys
    cwd = str(tmp_path)
    env = dict(os.environ, GNNDELETE_FORCE_EPOCHS='2', GNNDELETE_FORCE_VALID_FREQ='2', GNNDELETE_FORCE_NUM_STEPS='3',
               GNNDELETE_DIST_BACKEND='gloo', PYTHONPATH=ROOT)
    data_dir = os.path.join(cwd, 'data', 'ogbl-synth')
    os.makedirs(data_dir)
    from gnndelete_amd.framework.synth import make_linkpred_dataset
    data, df = make_linkpred_dataset(None, seed=42, shape=(800, 32, 4000, 'dense'))
    data.save(os.path.join(data_dir, 'd_42.pt'))
    torch.save(df, os.path.join(data_dir, 'df_42.pt'))
    common = ['--dataset', 'ogbl-synth', '--gnn', 'gcn', '--random_seed', '42', '--batch_size', '200']
    delete = common + ['--unlearning_model', 'gnndelete_nodeemb', '--df', 'in', '--df_size', '5', '--minibatch']
    subprocess.run([sys.executable, os.path.join(ROOT, 'train_gnn.py')] + common, cwd=cwd, env=env, check=True, capture_output=True)
    out = os.path.join(cwd, 'checkpoint', 'ogbl-synth', 'gcn', 'gnndelete_nodeemb', 'mse_mean-both_layerwise-0.5-non_connected',
                       'in-5.0-42')
    subprocess.run([sys.executable, os.path.join(ROOT, 'delete_gnn.py')] + delete, cwd=cwd, env=env, check=True, capture_output=True)
    single = torch.load(os.path.join(out, 'model_final.pt'))['model_state']
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
                        '--master-port', str(free_port()), os.path.join(ROOT, 'delete_gnn.py')] + delete, cwd=cwd, env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    w0 = torch.load(os.path.join(out, 'model_final.pt'))['model_state']
    w1 = torch.load(os.path.join(out, 'rank1', 'model_final.pt'))['model_state']
    for k in ('deletion1.deletion_weight', 'deletion2.deletion_weight'):
        assert torch.equal(w0[k], w1[k]), k
        assert not torch.equal(w0[k], single[k]) and bool(torch.isfinite(w0[k]).all()), k


@pytest.mark.parametrize('gnn', ['gcn', 'gat'])
def test_edgeprob_trainer_reproduces_reference_trajectory(tmp_path, monkeypatch, gnn):
    """GNNDeleteTrainer.train_fullbatch on the HIP path (fused pair kernel, csrc/pairs.hip) against the
    trajectory of the reference's real loop (gnndelete.py:138-309), same injected negatives."""
    from types import SimpleNamespace
    from gnndelete_amd.framework.data import Data
    from gnndelete_amd.framework.trainer import gnndelete as TE
    fx = load_golden(f'traj_edgeprob_{gnn}.npz')
    state, data, rest = split_fixture(fx)
    m = hip_model(gnn, state, data['sdf_node_1hop_mask'], data['sdf_node_2hop_mask'])
    neg = t(rest['neg']).cuda()
    monkeypatch.setattr(TE, 'negative_sampling', lambda **kw: neg)
    epochs = int(rest['epochs'])
    args = SimpleNamespace(unlearning_model='gnndelete', dataset='Cora', checkpoint_dir=str(tmp_path),
                           eval_on_cpu=False, epochs=epochs, valid_freq=1, lr=float(rest['lr']))
    opt = torch.optim.Adam([p for n, p in m.named_parameters() if 'del' in n], lr=args.lr)
    tr = TE.GNNDeleteTrainer(args)
    torch.manual_seed(int(rest['eval_seed']))
    tr.train(m, Data(data), opt, args, logits_ori=t(rest['logits_ori']))
    logs = [r for r in tr.trainer_log['log'] if 'train_loss_l' in r]
    assert len(logs) == epochs
    np.testing.assert_allclose([r['train_loss'] for r in logs], rest['train_loss'], rtol=1e-4)
    np.testing.assert_allclose([r['train_loss_l'] for r in logs], rest['loss_l'], rtol=1e-4)
    np.testing.assert_allclose([r['train_loss_r'] for r in logs], rest['loss_r'], rtol=1e-4)
    assert rel_l2(m.deletion1.deletion_weight.detach().cpu(), rest['final_w1']) < 1e-4
    assert rel_l2(m.deletion2.deletion_weight.detach().cpu(), rest['final_w2']) < 1e-4


@pytest.mark.parametrize('gnn', ['gcn', 'gat', 'gin'])
def test_original_trainer_reproduces_reference_training(tmp_path, monkeypatch, gnn):
    """Trainer.train (original-model training) on the HIP convs - weight gradients included -
    against the reference's real train_fullbatch loop (base.py:75-142), same injected negatives."""
    from types import SimpleNamespace
    from gnndelete_amd.framework import get_model
    from gnndelete_amd.framework.data import Data
    from gnndelete_amd.framework.trainer import base as TB
    fx = load_golden(f'orig_{gnn}.npz')
    state, data, rest = split_fixture(fx)
    key = {'gcn': 'conv1.lin.weight', 'gat': 'conv1.lin_src.weight', 'gin': 'conv1.nn.weight'}[gnn]
    key2 = key.replace('conv1', 'conv2')
    args = SimpleNamespace(unlearning_model='original', gnn=gnn, dataset='Cora', checkpoint_dir=str(tmp_path),
                           in_dim=state[key].shape[1], hidden_dim=state[key].shape[0], out_dim=state[key2].shape[0],
                           eval_on_cpu=False, epochs=int(rest['epochs']), valid_freq=1, lr=float(rest['lr']))
    m = get_model(args)
    res = m.load_state_dict(state, strict=False)
    assert not res.unexpected_keys and not [k for k in res.missing_keys if 'lin_dst' not in k]
    neg = t(rest['neg']).cuda()
    monkeypatch.setattr(TB, 'negative_sampling', lambda *a, **k: neg)
    opt = torch.optim.Adam(m.parameters(), lr=args.lr)
    tr = TB.Trainer(args)
    torch.manual_seed(int(rest['eval_seed']))
    tr.train(m, Data(data), opt, args)
    logs = [r for r in tr.trainer_log['log'] if 'train_loss' in r]
    np.testing.assert_allclose([r['train_loss'] for r in logs], rest['train_loss'], rtol=1e-4)
    final = {k[len('final::'):]: v for k, v in fx.items() if k.startswith('final::')}
    for k, v in m.state_dict().items():
        if k in final:
            assert rel_l2(v.cpu(), final[k]) < 1e-4, k


def test_node_unlearning_trainer_reproduces_reference_trajectory(tmp_path, monkeypatch):
    """GNNDeleteNodeClassificationTrainer (delete_node.py's trainer) on the HIP path against the reference's real
    loop (gnndelete_nodeemb.py:498-657) incl. its accuracy / micro-F1 evaluation; out_dim = 4 classes, so the
    last-layer Del takes the generic-width kernels."""
    from types import SimpleNamespace
    from gnndelete_amd.framework.data import Data
    from gnndelete_amd.framework.trainer import gnndelete_nodeemb as TN
    fx = load_golden('traj_nodecls_gat.npz')
    state, data, rest = split_fixture(fx)
    m = hip_model('gat', state, data['sdf_node_1hop_mask'], data['sdf_node_2hop_mask'])
    neg = t(rest['neg']).cuda()
    monkeypatch.setattr(TN, 'negative_sampling', lambda *a, **k: neg)
    epochs = int(rest['epochs'])
    args = SimpleNamespace(unlearning_model='gnndelete_nodeemb', dataset='DBLP', checkpoint_dir=str(tmp_path),
                           eval_on_cpu=False, epochs=epochs, valid_freq=3, lr=float(rest['lr']), alpha=float(rest['alpha']),
                           loss_fct='mse_mean', loss_type='both_layerwise', gnn='gat')
    opt = [torch.optim.Adam(m.deletion1.parameters(), lr=args.lr), torch.optim.Adam(m.deletion2.parameters(), lr=args.lr)]
    tr = TN.GNNDeleteNodeClassificationTrainer(args)
    tr.train(m, Data(data), opt, args)
    assert rel_l2(m.deletion1.deletion_weight.detach().cpu(), rest['final_w1']) < 1e-4
    assert rel_l2(m.deletion2.deletion_weight.detach().cpu(), rest['final_w2']) < 1e-4
    vals = [r for r in tr.trainer_log['log'] if 'val_dt_acc' in r]
    np.testing.assert_allclose([v['val_dt_acc'] for v in vals], rest['val_dt_acc'], atol=1e-9)
    np.testing.assert_allclose([v['val_loss'] for v in vals], rest['val_loss'], rtol=1e-4)
    tl = [r for r in tr.trainer_log['log'] if 'train_loss' in r]
    np.testing.assert_allclose([r['train_loss'] for r in tl], rest['train_loss'][[2, 5]], rtol=1e-4)


def test_large_graph_df_subsets_are_drawn_on_the_device(monkeypatch):
    """Above Trainer.FAST_SUBSETS_ABOVE Dr edges the 500 resampled Dr subsets come from device-side permutations
    (same statistic, different random stream - upstream's 500 host permutations are 19 s per request at ogbl-collab
    size): every subset has |Df| distinct in-range edges, the subsets differ, and the seed reproduces them."""
    from types import SimpleNamespace
    from gnndelete_amd.framework.trainer.base import Trainer
    monkeypatch.setattr(Trainer, 'FAST_SUBSETS_ABOVE', 1000)
    dev = torch.device('cuda')

    def draw(seed):
        torch.manual_seed(seed)
        tr = Trainer.__new__(Trainer)
        tr.df_pos_edge = []
        tr._ensure_df_subsets(5000, 37, dev)
        return tr

    a, b, c = draw(3), draw(3), draw(4)
    idx = a._df_subset_index
    assert idx.shape == (500, 37) and idx.is_cuda and len(a.df_pos_edge) == 500
    assert int(idx.min()) >= 0 and int(idx.max()) < 5000
    assert all(len(set(r.tolist())) == 37 for r in idx[:20])
    assert not torch.equal(idx[0], idx[1])
    assert torch.equal(idx, b._df_subset_index) and not torch.equal(idx, c._df_subset_index)


# ----------------------------------------------------------------------------- round-2 fixtures (injected batches)
def _lists(fx, prefix, count_key):
    return [t(fx[f'{prefix}::{i}']) for i in range(int(fx[count_key]))]


def test_minibatch_trainer_reproduces_reference_trajectory(tmp_path, monkeypatch):
    """The GraphSAINT mini-batch loop (gnndelete_nodeemb.py:352-495) on the HIP path, on the node sets and negatives
    the reference's real loop consumed: per-step losses, final Del weights (the W_D1 gradient that loss2 leaves
    behind for the next batch included), validation AUCs."""
    from types import SimpleNamespace
    from gnndelete_amd.framework.data import Data
    from gnndelete_amd.framework.trainer import gnndelete_nodeemb as TN
    from gnndelete_amd.framework.trainer import sampler as S
    fx = load_golden('traj_minibatch_gat.npz')
    state, data, rest = split_fixture(fx)
    m = hip_model('gat', state, data['sdf_node_1hop_mask'], data['sdf_node_2hop_mask'])
    sets, negs = _lists(fx, 'batch', 'n_batches'), iter(_lists(fx, 'negs', 'n_negs'))
    monkeypatch.setattr(S, 'make_sampler', lambda d, batch_size, num_steps, walk_length=2: S.FixedNodeSets(d, sets))
    monkeypatch.setattr(S, 'negative_sampling', lambda ei, n, k: next(negs).to(ei.device))
    epochs = int(rest['epochs'])
    args = SimpleNamespace(unlearning_model='gnndelete_nodeemb', dataset='ogbl-synth', checkpoint_dir=str(tmp_path),
                           eval_on_cpu=False, epochs=epochs, valid_freq=epochs, lr=float(rest['lr']),
                           alpha=float(rest['alpha']), loss_fct='mse_mean', loss_type='both_layerwise', gnn='gat',
                           batch_size=40, num_steps=len(sets), minibatch=True)
    opt = [torch.optim.Adam(m.deletion1.parameters(), lr=args.lr), torch.optim.Adam(m.deletion2.parameters(), lr=args.lr)]
    tr = TN.GNNDeleteNodeembTrainer(args)
    torch.manual_seed(int(rest['eval_seed']))
    tr.train(m, Data(data), opt, args)
    steps = tr.trainer_log['steps']
    assert len(steps) == len(rest['train_loss'])
    for key in ['train_loss', 'train_loss_l', 'train_loss_r']:
        np.testing.assert_allclose([s_[key] for s_ in steps], rest[key], rtol=1e-4, atol=1e-8, err_msg=key)
    assert rel_l2(m.deletion1.deletion_weight.detach().cpu(), rest['final_w1']) < 1e-4
    assert rel_l2(m.deletion2.deletion_weight.detach().cpu(), rest['final_w2']) < 1e-4
    vals = [r for r in tr.trainer_log['log'] if 'val_dt_auc' in r]
    assert abs(vals[-1]['val_dt_auc'] - float(rest['val_dt_auc'][-1])) < 2e-3
    assert abs(vals[-1]['val_df_auc'] - float(rest['val_df_auc'][-1])) < 2e-3


@pytest.mark.parametrize('gnn', ['gcn', 'gat'])
def test_edgeprob_minibatch_trainer_reproduces_reference_trajectory(gnn, tmp_path, monkeypatch):
    """The edge-probability mini-batch loop (framework/trainer/gnndelete.py:312-450) on the HIP path, on the node sets and
    negatives the reference's loop consumed (run with the data.dtrain_mask upstream never sets injected as dr_mask): the
    epoch log with upstream's double division and swapped names, final Del weights, validation AUCs."""
    from types import SimpleNamespace
    from gnndelete_amd.framework.data import Data
    from gnndelete_amd.framework.trainer import gnndelete as TG
    from gnndelete_amd.framework.trainer import sampler as S
    fx = load_golden(f'traj_edgeprob_minibatch_{gnn}.npz')
    state, data, rest = split_fixture(fx)
    m = hip_model(gnn, state, data['sdf_node_1hop_mask'], data['sdf_node_2hop_mask'])
    sets, negs = _lists(fx, 'batch', 'n_batches'), iter(_lists(fx, 'negs', 'n_negs'))
    monkeypatch.setattr(S, 'make_sampler', lambda d, batch_size, num_steps, walk_length=2: S.FixedNodeSets(d, sets))
    monkeypatch.setattr(S, 'negative_sampling', lambda ei, n, k: next(negs).to(ei.device))
    epochs = int(rest['epochs'])
    args = SimpleNamespace(unlearning_model='gnndelete', dataset='ogbl-synth', checkpoint_dir=str(tmp_path), eval_on_cpu=False,
                           epochs=epochs, valid_freq=epochs, lr=float(rest['lr']), gnn=gnn, batch_size=40, num_steps=len(sets),
                           minibatch=True)
    opt = torch.optim.Adam([{'params': [p for n_, p in m.named_parameters() if 'del' in n_], 'weight_decay': 0.0}], lr=args.lr)
    tr = TG.GNNDeleteTrainer(args)
    torch.manual_seed(int(rest['eval_seed']))
    data.pop('dtrain_mask', None)                                     # the trainer defaults it to dr_mask
    tr.train(m, Data(data), opt, args)
    assert len(tr.trainer_log['steps']) == epochs * len(sets)
    logged = [r for r in tr.trainer_log['log'] if 'train_loss' in r][-1]
    for key in ['train_loss', 'train_loss_l', 'train_loss_e']:
        np.testing.assert_allclose(logged[key], rest['log_' + key][-1], rtol=1e-4, atol=1e-9, err_msg=key)
    assert rel_l2(m.deletion1.deletion_weight.detach().cpu(), rest['final_w1']) < 1e-4
    assert rel_l2(m.deletion2.deletion_weight.detach().cpu(), rest['final_w2']) < 1e-4
    vals = [r for r in tr.trainer_log['log'] if 'val_dt_auc' in r]
    assert abs(vals[-1]['val_dt_auc'] - float(rest['val_dt_auc'][-1])) < 2e-3
    assert abs(vals[-1]['val_df_auc'] - float(rest['val_df_auc'][-1])) < 2e-3


def _kg_fixture(name):
    from gnndelete_amd.framework.data import Data
    fx = load_golden(name)
    state, data, rest = split_fixture(fx)
    R_ = int(rest['num_edge_type'])
    m = hip_model('rgcn', state, data['sdf_node_1hop_mask'], data['sdf_node_2hop_mask'], num_nodes=data['num_nodes'],
                  num_edge_type=R_)
    return fx, m, Data(data), rest, R_


def test_kg_trainer_reproduces_reference_trajectory(tmp_path, monkeypatch):
    """KGGNNDeleteNodeembTrainer.train (gnndelete_nodeemb.py:659-846) on the HIP path (typed conv kernels, 21
    relation types -> block-diagonal weights) on the reference's batches; negative_sampling_kg and the 500 Dr
    subsets of the closing validation re-draw the reference's random stream from the recorded seed."""
    from types import SimpleNamespace
    from gnndelete_amd.framework.trainer import kg as TK
    from gnndelete_amd.framework.trainer import sampler as S
    fx, m, data, rest, R_ = _kg_fixture('traj_kg_rgcn.npz')
    sets = _lists(fx, 'batch', 'n_batches')
    monkeypatch.setattr(S, 'make_sampler', lambda d, batch_size, num_steps, walk_length=2: S.FixedNodeSets(d, sets))
    epochs = int(rest['epochs'])
    args = SimpleNamespace(unlearning_model='gnndelete_nodeemb', dataset='WordNet18', checkpoint_dir=str(tmp_path),
                           eval_on_cpu=False, epochs=epochs, valid_freq=epochs, lr=float(rest['lr']), alpha=float(rest['alpha']),
                           loss_fct='mse_mean', loss_type='both_layerwise', gnn='rgcn', batch_size=30, num_steps=len(sets),
                           num_edge_type=R_)
    opt = [torch.optim.Adam(m.deletion1.parameters(), lr=args.lr), torch.optim.Adam(m.deletion2.parameters(), lr=args.lr)]
    tr = TK.KGGNNDeleteNodeembTrainer(args)
    torch.manual_seed(int(rest['seed']))
    tr.train(m, data, opt, args)
    steps = tr.trainer_log['steps']
    assert len(steps) == len(rest['train_loss'])
    for key in ['train_loss', 'loss_r', 'loss_l']:
        np.testing.assert_allclose([s_[key] for s_ in steps], rest[key], rtol=1e-4, atol=1e-8, err_msg=key)
    assert rel_l2(m.deletion1.deletion_weight.detach().cpu(), rest['final_w1']) < 1e-4
    assert rel_l2(m.deletion2.deletion_weight.detach().cpu(), rest['final_w2']) < 1e-4
    vals = [r for r in tr.trainer_log['log'] if 'val_dt_auc' in r]
    assert abs(vals[-1]['val_loss'] - float(rest['val_loss'][-1])) < 1e-4
    assert abs(vals[-1]['val_dt_auc'] - float(rest['val_dt_auc'][-1])) < 2e-3
    assert abs(vals[-1]['val_df_auc'] - float(rest['val_df_auc'][-1])) < 2e-3


def test_kg_eval_matches_reference_golden(tmp_path):
    """KGTrainer.eval (base.py:495-567) on the HIP path: DistMult scores without sigmoid for the Dt loss / AUC / AUP,
    500 fresh host permutations for the Df-vs-Dr statistics (same seed -> same subsets as upstream)."""
    from types import SimpleNamespace
    from gnndelete_amd.framework.trainer import kg as TK
    fx, m, data, rest, R_ = _kg_fixture('eval_kg.npz')
    args = SimpleNamespace(unlearning_model='gnndelete_nodeemb', dataset='WordNet18', checkpoint_dir=str(tmp_path),
                           eval_on_cpu=False, num_edge_type=R_)
    tr = TK.KGTrainer(args)
    torch.manual_seed(int(rest['eval_seed']))
    loss, dt_auc, dt_aup, df_auc, df_aup, df_logit, _, log = tr.eval(m, data, 'test')
    assert abs(loss - float(rest['test_loss'])) < 1e-4
    assert abs(dt_auc - float(rest['test_dt_auc'])) < 2e-3 and abs(dt_aup - float(rest['test_dt_aup'])) < 2e-3
    assert abs(df_auc - float(rest['test_df_auc'])) < 2e-3 and abs(df_aup - float(rest['test_df_aup'])) < 2e-3
    np.testing.assert_allclose(np.array(df_logit), rest['test_df_logit'], rtol=1e-4)


def test_retrain_trainer_reproduces_reference(tmp_path, monkeypatch):
    """RetrainTrainer (retrain.py:39-131) on the HIP convs: Dr-only message passing / positives / negative count,
    model selection on dt_auc + df_auc; and Trainer.test's verification error against a second model
    (evaluation.py:63-81)."""
    from types import SimpleNamespace
    from gnndelete_amd.framework import get_model, get_trainer
    from gnndelete_amd.framework.data import Data
    from gnndelete_amd.framework.evaluation import verification_error
    from gnndelete_amd.framework.trainer import retrain as TR
    fx = load_golden('retrain_gcn.npz')
    state, data, rest = split_fixture(fx)
    w1, w2 = state['conv1.lin.weight'], state['conv2.lin.weight']
    epochs = int(rest['epochs'])
    args = SimpleNamespace(unlearning_model='retrain', gnn='gcn', dataset='Cora', checkpoint_dir=str(tmp_path),
                           in_dim=w1.shape[1], hidden_dim=w1.shape[0], out_dim=w2.shape[0], eval_on_cpu=False, epochs=epochs,
                           valid_freq=epochs, lr=float(rest['lr']))
    m = get_model(args)
    m.load_state_dict(state)
    negs = iter(_lists(fx, 'negs', 'n_negs'))
    n_dr = int(data['dr_mask'].sum())

    def fake_neg(edge_index, num_nodes, num_neg_samples):
        assert edge_index.shape[1] == n_dr and num_neg_samples == n_dr          # Dr only
        return next(negs).to(edge_index.device)
    monkeypatch.setattr(TR, 'negative_sampling', fake_neg)
    tr = get_trainer(args)
    assert isinstance(tr, TR.RetrainTrainer)
    opt = torch.optim.Adam(m.parameters(), lr=args.lr)
    torch.manual_seed(int(rest['eval_seed']))
    d = Data(data)
    d.dtrain_mask = d.dr_mask
    tr.train(m, d, opt, args)
    np.testing.assert_allclose([s_['train_loss'] for s_ in tr.trainer_log['steps']], rest['train_loss'], rtol=1e-4)
    final = {k[len('final::'):]: v for k, v in fx.items() if k.startswith('final::')}
    for k, v in m.state_dict().items():
        assert rel_l2(v.cpu(), final[k]) < 1e-4, k
    vals = [r for r in tr.trainer_log['log'] if 'val_dt_auc' in r]
    assert abs(vals[-1]['val_dt_auc'] - float(rest['val_dt_auc'][-1])) < 2e-3
    assert abs(vals[-1]['val_df_auc'] - float(rest['val_df_auc'][-1])) < 2e-3
    assert os.path.exists(os.path.join(str(tmp_path), 'model_best.pt'))
    other = get_model(args)
    other.load_state_dict({k[len('other::'):]: t(v) for k, v in fx.items() if k.startswith('other::')})
    ve = float(verification_error(m, other.cuda()))
    assert abs(ve - float(rest['ve'])) < 1e-3 * float(rest['ve'])


def test_original_minibatch_trainers_reproduce_reference(tmp_path, monkeypatch):
    """Original-model MINI-BATCH training on the HIP convs (conv weight gradients through the kernels) on the batches
    the reference's loops consumed: Trainer.train_minibatch (base.py:144-227, GCN) and KGTrainer.train
    (base.py:394-493, RGCN with 21 relation types: per-relation autograd path, DistMult, negative_sampling_kg
    re-drawn from the recorded seed)."""
    from types import SimpleNamespace
    from gnndelete_amd.framework import get_model
    from gnndelete_amd.framework.data import Data
    from gnndelete_amd.framework.trainer import base as TB
    from gnndelete_amd.framework.trainer import kg as TK
    from gnndelete_amd.framework.trainer import sampler as S
    fx = load_golden('orig_minibatch_gcn.npz')
    state, data, rest = split_fixture(fx)
    w1, w2 = state['conv1.lin.weight'], state['conv2.lin.weight']
    sets, negs = _lists(fx, 'batch', 'n_batches'), iter(_lists(fx, 'negs', 'n_negs'))
    monkeypatch.setattr(S, 'make_sampler', lambda d, batch_size, num_steps, walk_length=2: S.FixedNodeSets(d, sets))
    monkeypatch.setattr(S, 'negative_sampling', lambda ei, n, k: next(negs).to(ei.device))
    epochs = int(rest['epochs'])
    args = SimpleNamespace(unlearning_model='original', gnn='gcn', dataset='ogbl-synth', checkpoint_dir=str(tmp_path / 'a'),
                           in_dim=w1.shape[1], hidden_dim=w1.shape[0], out_dim=w2.shape[0], eval_on_cpu=False, epochs=epochs,
                           valid_freq=epochs, lr=float(rest['lr']), batch_size=40, num_steps=len(sets), minibatch=True)
    m = get_model(args)
    m.load_state_dict(state)
    opt = torch.optim.Adam(m.parameters(), lr=args.lr)
    tr = TB.Trainer(args)
    torch.manual_seed(int(rest['eval_seed']))
    tr.train(m, Data(data), opt, args)
    np.testing.assert_allclose([s_['train_loss'] for s_ in tr.trainer_log['steps']], rest['train_loss'], rtol=1e-4)
    final = {k[len('final::'):]: v for k, v in fx.items() if k.startswith('final::')}
    for k, v in m.state_dict().items():
        assert rel_l2(v.cpu(), final[k]) < 1e-4, k
    vals = [r for r in tr.trainer_log['log'] if 'val_loss' in r]
    assert abs(vals[-1]['val_loss'] - float(rest['val_loss'][-1])) < 1e-4

    fx = load_golden('orig_kg_rgcn.npz')
    state, data, rest = split_fixture(fx)
    R_ = int(rest['num_edge_type'])
    sets = _lists(fx, 'batch', 'n_batches')
    monkeypatch.setattr(S, 'make_sampler', lambda d, batch_size, num_steps, walk_length=2: S.FixedNodeSets(d, sets))
    epochs = int(rest['epochs'])
    args = SimpleNamespace(unlearning_model='original', gnn='rgcn', dataset='WordNet18', checkpoint_dir=str(tmp_path / 'b'),
                           in_dim=state['node_emb.weight'].shape[1], hidden_dim=state['conv1.root'].shape[1],
                           out_dim=state['conv2.root'].shape[1], eval_on_cpu=False, epochs=epochs, valid_freq=epochs,
                           lr=float(rest['lr']), num_steps=len(sets), num_edge_type=R_)
    m = get_model(args, num_nodes=data['num_nodes'], num_edge_type=R_)
    m.load_state_dict(state)
    opt = torch.optim.Adam(m.parameters(), lr=args.lr)
    tr = TK.KGTrainer(args)
    torch.manual_seed(int(rest['seed']))
    tr.train(m, Data(data), opt, args)
    np.testing.assert_allclose([s_['train_loss'] for s_ in tr.trainer_log['steps']], rest['train_loss'], rtol=2e-4)
    final = {k[len('final::'):]: v for k, v in fx.items() if k.startswith('final::')}
    for k, v in m.state_dict().items():
        assert rel_l2(v.cpu(), final[k]) < 2e-4, k
    vals = [r for r in tr.trainer_log['log'] if 'val_dt_aup' in r]
    assert abs(vals[-1]['val_dt_aup'] - float(rest['val_dt_aup'][-1])) < 2e-3


def test_kg_retrain_trainer_reproduces_reference(tmp_path, monkeypatch):
    """KGRetrainTrainer.train (framework/trainer/retrain.py:235-339) on the HIP convs, on the batches / negatives stream the
    reference's loop consumed (retrain_kg_rgcn.npz, recorded from the reference's own KGRetrainTrainer + RGCN at 21
    relation types): per-step losses, final weights (gradient-norm clipping included), validation figures."""
    from types import SimpleNamespace
    from gnndelete_amd.framework import get_model
    from gnndelete_amd.framework.data import Data
    from gnndelete_amd.framework.trainer import retrain as TR
    from gnndelete_amd.framework.trainer import sampler as S
    fx = load_golden('retrain_kg_rgcn.npz')
    state, data, rest = split_fixture(fx)
    R_ = int(rest['num_edge_type'])
    sets = _lists(fx, 'batch', 'n_batches')
    monkeypatch.setattr(S, 'make_sampler', lambda d, batch_size, num_steps, walk_length=2: S.FixedNodeSets(d, sets))
    epochs = int(rest['epochs'])
    args = SimpleNamespace(unlearning_model='retrain', gnn='rgcn', dataset='WordNet18', checkpoint_dir=str(tmp_path),
                           in_dim=state['node_emb.weight'].shape[1], hidden_dim=state['conv1.root'].shape[1],
                           out_dim=state['conv2.root'].shape[1], eval_on_cpu=False, epochs=epochs, valid_freq=epochs,
                           lr=float(rest['lr']), num_steps=len(sets), num_edge_type=R_)
    m = get_model(args, num_nodes=data['num_nodes'], num_edge_type=R_)
    m.load_state_dict(state)
    opt = torch.optim.Adam(m.parameters(), lr=args.lr)
    tr = TR.KGRetrainTrainer(args)
    torch.manual_seed(int(rest['seed']))
    tr.train(m, Data(data), opt, args)
    np.testing.assert_allclose([s_['train_loss'] for s_ in tr.trainer_log['steps']], rest['train_loss'], rtol=2e-4)
    final = {k[len('final::'):]: v for k, v in fx.items() if k.startswith('final::')}
    for k, v in m.state_dict().items():
        assert rel_l2(v.cpu(), final[k]) < 2e-4, k
    vals = [r for r in tr.trainer_log['log'] if 'val_dt_aup' in r]
    assert abs(vals[-1]['val_dt_aup'] - float(rest['val_dt_aup'][-1])) < 2e-3
    assert os.path.exists(os.path.join(str(tmp_path), 'model_best.pt'))
