"""CPU-oracle legs of the long GPU tests as CHILD PROCESSES (the GPU suite against its time limit).

The full-size parity tests spend most of their wall time in the CPU oracle (20 full-graph GCN iterations at collab size are
30 s of host time during which the GPU idles); the 600-epoch tests in their four CPU ensemble members.  Every such leg is a
pure function of seeds (the requests are built by seeded, CPU-only code that parent and child both run; the child's checksum
of state + negatives is compared with the parent's), so it can run in a child process while the parent - and, when the whole
suite runs, OTHER tests - use the GPU.  Same oracle code, same inputs, same thread count as before; nothing about what is
compared changes.

  start(name)        -> launches the job's own child once (idempotent; a test run on its own overlaps its CPU leg with its GPU legs)
  result(name)       -> waits for it, returns what it saved (torch.load)
  prefetch(names, members)
                     -> ONE background worker that runs the named jobs and then the given ensemble members of the 600-epoch
                        tests (tests/parity_member.py) one after the other on BG_THREADS host threads (tests/conftest.py calls it at
                        session start when the whole suite runs).  One worker, few threads, on purpose: the GPU boxes run under a
                        cgroup quota of 16 CPUs (256 are visible) - twelve children with 128 threads between them throttled the
                        whole session (the suite took 980 s instead of 640, profiles/NOTES.md round 6).

Run as a script (`python tests/oracle_jobs.py <name> <out.pt>` / `--queue <spec.pt>`) this file IS the child."""
import os
import subprocess
import sys
import tempfile

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

# name -> (kind, parameters).  iters = tests/test_full_size_gpu.py ITERS
JOBS = {
    'full-collab-gcn': ('linkpred', dict(workload='synth-collab', gnn='gcn', df='in', df_size=5.0, iters=20)),
    'full-collab-sage': ('linkpred', dict(workload='synth-collab', gnn='sage', df='in', df_size=5.0, iters=10)),
    'full-collab-gat': ('linkpred', dict(workload='synth-collab', gnn='gat', df='in', df_size=5.0, iters=10)),
    'full-nodecls-gat': ('nodecls', dict(epochs=10, lr=1e-2, alpha=0.5)),
    # (CPU-suite check of this machinery itself, never prefetched: tests/test_host_utils.py)
    'selftest-small': ('linkpred', dict(workload='synth-small', gnn='gcn', df='in', df_size=5.0, iters=2)),
}
# host threads per job: four jobs + the eight 8-thread ensemble members of the 600-epoch tests run side by side at session start
# (128 threads in all); the oracle's epoch is as fast on 16 threads as on 32 (1.68 against 1.56 s, bench.py's thread sweep)
THREADS = 16
BG_THREADS = 8            # the session-start worker: half of the box's CPU quota, the session itself keeps the rest
_running = {}
_tmp = None
_queue = None             # (worker process, {job name or member out file: out file})


def checksum(state, neg):
    """One number per request: parent and child must have built the same weights and negatives."""
    return float(sum(v.double().abs().sum() for v in state.values()) + neg.double().sum())


def linkpred_request(workload, gnn, df, df_size, seed=42):
    from types import SimpleNamespace
    sys.path.insert(0, ROOT)
    import bench
    args = SimpleNamespace(workload=workload, gnn=gnn, df=df, df_size=df_size, seed=seed)
    return bench.build_request(args, torch.device('cpu'))


def nodecls_request():
    """BASELINE config 5 at the size it names: delete_node.py's request (delete_node.py:77-142) on the collab-shaped
    node-classification stand-in; -> (data, state of a GATDelete built on it, negatives, model arguments)."""
    from types import SimpleNamespace
    from gnndelete_amd.framework.graph_utils import k_hop_subgraph, negative_sampling
    from gnndelete_amd.framework.models import GATDelete
    from gnndelete_amd.framework.synth import make_nodecls_dataset
    from gnndelete_amd.framework.utils import seed_everything
    data = make_nodecls_dataset('synth-collab', seed=42)
    n = data.num_nodes
    seed_everything(42)
    df_nodes = torch.randperm(n)[:int(0.05 * n)]
    gone = torch.zeros(n, dtype=torch.bool)
    gone[df_nodes] = True
    E = data.edge_index
    df_mask = gone[E[0]] | gone[E[1]]
    df_edge = E[:, df_mask]
    data.directed_df_edge_index = df_edge[:, df_edge[0] < df_edge[1]]
    seeds = df_edge.flatten().unique()
    _, e2, _, m2e = k_hop_subgraph(seeds, 2, E, num_nodes=n)
    _, e1, _, _ = k_hop_subgraph(seeds, 1, E, num_nodes=n)
    s1, s2 = torch.zeros(n, dtype=torch.bool), torch.zeros(n, dtype=torch.bool)
    s1[e1.flatten().unique()] = True
    s2[e2.flatten().unique()] = True
    data.sdf_node_1hop_mask, data.sdf_node_2hop_mask, data.sdf_mask, data.df_mask = s1, s2, m2e, df_mask
    data.dr_mask = data.dtrain_mask = ~df_mask
    torch.manual_seed(9)
    hip = GATDelete(SimpleNamespace(in_dim=data.x.shape[1], hidden_dim=128, out_dim=data.num_classes), s1, s2)
    state = {k: v.clone() for k, v in hip.state_dict().items()}
    neg = negative_sampling(E, n, int(df_mask.sum()), generator=torch.Generator().manual_seed(4))
    return data, hip, state, neg


def _run_linkpred(workload, gnn, df, df_size, iters):
    from oracle import gnndelete_ref as R
    data, model, neg, ni1, ni2 = linkpred_request(workload, gnn, df, df_size)
    state = {k: v.clone() for k, v in model.state_dict().items()}
    torch.set_num_threads(min(THREADS, torch.get_num_threads()))
    ref = R.TwoLayerDelete(gnn, data.x.shape[1], 128, 64, data.sdf_node_1hop_mask, data.sdf_node_2hop_mask)
    ref.load_state_dict(state, strict=False)
    E = data.train_pos_edge_index
    e_dr, e_sdf, pos = E[:, data.dr_mask], E[:, data.sdf_mask], E[:, data.df_mask]
    with torch.no_grad():
        z1o, z2o = ref.get_original_embeddings(data.x, e_dr, return_all_emb=True)
    targets = dict(z1_ori=z1o, z2_ori=z2o, pos_edge=pos, neg_edge=neg, ni_mask1=ni1, ni_mask2=ni2)
    opt = R.make_optimizer(ref, 'both_layerwise', 1e-3)
    logs = [R.nodeemb_epoch(ref, lambda: ref(data.x, e_sdf, return_all_emb=True), targets, opt, 'both_layerwise', 0.5,
                            R.LOSSES['mse_mean']) for _ in range(iters)]
    return dict(logs=logs, w1=ref.deletion1.deletion_weight.detach().clone(), w2=ref.deletion2.deletion_weight.detach().clone(),
                checksum=checksum(state, neg))


def _run_nodecls(epochs, lr, alpha):
    from oracle import gnndelete_ref as R
    data, _, state, neg = nodecls_request()
    torch.set_num_threads(min(THREADS, torch.get_num_threads()))
    ref = R.TwoLayerDelete('gat', data.x.shape[1], 128, data.num_classes, data.sdf_node_1hop_mask, data.sdf_node_2hop_mask)
    ref.load_state_dict(state, strict=False)
    d = {k: v for k, v in data.items()}
    d['train_pos_edge_index'] = data.edge_index
    logs, _ = R.nodeemb_fullbatch(ref, d, epochs, 'both_layerwise', alpha, 'mse_mean', lr, neg_edge=neg)
    return dict(logs=logs, w1=ref.deletion1.deletion_weight.detach().clone(), w2=ref.deletion2.deletion_weight.detach().clone(),
                checksum=checksum(state, neg))


def _tmpdir():
    global _tmp
    if _tmp is None:
        _tmp = tempfile.TemporaryDirectory(prefix='gd_oracle_jobs_')
    return _tmp.name


def start(name):
    if name in _running or (_queue is not None and name in _queue[1]):
        return
    out = os.path.join(_tmpdir(), name + '.pt')
    env = dict(os.environ, HIP_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES='', PYTHONPATH=ROOT + os.pathsep + HERE)
    env.pop('OMP_NUM_THREADS', None)
    proc = subprocess.Popen([sys.executable, os.path.abspath(__file__), name, out], env=env, stdout=subprocess.PIPE,
                            stderr=subprocess.STDOUT, text=True)
    _running[name] = (proc, out)


def wait_for_file(out, timeout=1800):
    """Block until the session-start worker has written `out` (it renames a finished file into place) -> True; False when the
    worker is gone without having written it."""
    import time
    t0 = time.time()
    while not os.path.exists(out):
        if _queue is None or _queue[0].poll() is not None:
            return os.path.exists(out)
        assert time.time() - t0 < timeout, f'timed out waiting for {out}'
        time.sleep(0.2)
    return True


def result(name, timeout=1800):
    if _queue is not None and name in _queue[1] and name not in _running:
        out = _queue[1][name]
        if wait_for_file(out, timeout):
            return torch.load(out, weights_only=False)
        log = _queue[0].stdout.read() if _queue[0].stdout else ''
        print(f'oracle_jobs: the background worker ended without {name} - running it directly\n{(log or "")[-1500:]}')
        _queue[1].pop(name)
    start(name)
    proc, out = _running[name]
    log, _ = proc.communicate(timeout=timeout)
    assert proc.returncode == 0, f'CPU-oracle job {name} failed:\n{(log or "")[-3000:]}'
    return torch.load(out, weights_only=False)


def prefetch(names, members=()):
    """names: jobs of JOBS; members: (request file, edge-order seed or None, out file) of tests/parity_member.py.  One worker
    process runs them in this order."""
    global _queue
    if _queue is not None or not (names or members):
        return
    outs = {n: os.path.join(_tmpdir(), n + '.pt') for n in names}
    spec = os.path.join(_tmpdir(), 'queue.pt')
    torch.save(dict(jobs=[(n, outs[n]) for n in names], members=list(members)), spec)
    env = dict(os.environ, HIP_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES='', PYTHONPATH=ROOT + os.pathsep + HERE,
               OMP_NUM_THREADS=str(BG_THREADS), OMP_WAIT_POLICY='PASSIVE', GD_ORACLE_JOB_THREADS=str(BG_THREADS))
    proc = subprocess.Popen([sys.executable, os.path.abspath(__file__), '--queue', spec], env=env, stdout=subprocess.PIPE,
                            stderr=subprocess.STDOUT, text=True)
    for _, _, out in members:
        outs[out] = out
    _queue = (proc, outs)


def _run(name, out):
    kind, params = JOBS[name]
    res = _run_linkpred(**params) if kind == 'linkpred' else _run_nodecls(**params)
    torch.save(res, out + '.part')
    os.replace(out + '.part', out)


if __name__ == '__main__':
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    THREADS = int(os.environ.get('GD_ORACLE_JOB_THREADS', THREADS))
    if sys.argv[1] == '--queue':
        spec = torch.load(sys.argv[2], weights_only=False)
        for name_, out_ in spec['jobs']:
            _run(name_, out_)
        import parity_member
        for req, perm, out_ in spec['members']:
            parity_member.run(req, perm, out_ + '.part', threads=THREADS)
            os.replace(out_ + '.part', out_)
    else:
        _run(sys.argv[1], sys.argv[2])
