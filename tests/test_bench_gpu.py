"""bench.py's contract on the GPU box: the one-GPU line, and the N > 1 launch exactly as the driver starts it
(torch.distributed.run, one process per rank) - here with both ranks on the box's single GPU over gloo
(GD_BENCH_BACKEND=gloo; RCCL refuses two ranks on one device), which runs the same partitioned step, halo exchanges
and replicas leg as an RCCL run."""
import json
import os
import subprocess
import sys

import pytest

from helpers import free_port

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ['--workload', 'synth-small', '--steps', '8', '--warmup', '2', '--pretrain_epochs', '3', '--no_cpu_baseline',
         '--no_cached_rate']


def _json_line(out):
    lines = [l for l in out.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, out[-3000:]
    return json.loads(lines[0])


def test_bench_single_gpu_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + SMALL + ['--cpu_baseline_iters', '2'], cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = _json_line(r.stdout)
    assert d['n_gpus'] == 1 and d['scaling'] == 'weak' and d['steps'] == 8 and d['warmup'] == 2 and d['value'] > 0
    assert d['unit'] == 'iters/s' and d['dtype'] == 'f32' and d['vs_baseline'] is None
    assert {'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'} <= set(d['roofline'])
    assert {'roofline_del_gemm', 'roofline_wgrad', 'roofline_spmm_d64'} <= set(d['extras'])


@pytest.mark.parametrize('probe', ['0', '1'])
def test_bench_two_ranks_run_the_partitioned_step(probe):
    """probe = 1 (+ GD_DIST_OVERLAP=1, the request for the overlapped program): both child-process self-tests run first - the
    partitioned step, then the overlapped exchanges against the synchronous ones (bit for bit) - and the parent runs with the
    exchanges under compute when the second one passed, synchronously with the reason in the line when it did not."""
    env = dict(os.environ, GD_BENCH_BACKEND='gloo', GD_BENCH_FORCE_PROBE=probe, HSA_ENABLE_IPC_MODE_LEGACY='0', GD_DIST_OVERLAP=probe)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--parallel', 'partition'] + SMALL
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    d = _json_line(r.stdout)
    assert d['n_gpus'] == 2 and d['scaling'] == 'strong', d
    how = d['config']['halo_exchanges']
    assert how.startswith('synchronous') if probe == '0' else (how.startswith('overlapped') or 'self-test failed' in how), how
    assert 'partition_fallback' not in d['config'], d['config']
    assert 'row-partition' in d['config']['parallelism']
    assert d['config']['halo']['recv_bytes_per_step'] > 0 and d['config']['ranks_seen'] == 2
    assert len(d['config']['halo_recv_send_bytes_per_rank']) == 2 and 'interior_rows_forward' in d['config']['halo']
    assert d['extras']['iters_per_s_independent_replicas'] > 0
    assert d['value'] > 0 and abs(d['value'] - d['steps'] / (d['ms_per_step'] * d['steps'] / 1e3)) < 1e-6 * d['value']


def test_bench_launches_its_own_ranks_and_auto_keeps_the_partitioned_headline():
    """`python bench.py --gpus 2` WITHOUT a torch.distributed environment (the form the one-GPU line is started in): bench.py
    starts the two ranks itself (a torch.distributed.run child, before any GPU call) and relays the line.  `--parallel auto`
    (the default): the headline is the partitioned step, "scaling": "strong", whatever the planner's estimate says; the
    replicas rate and the estimate are reported next to it (a curve must not mix strong- and weak-scaling points)."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(GD_BENCH_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'] + SMALL, cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert 'launching 2 ranks' in r.stderr
    d = _json_line(r.stdout)
    est = d['config']['parallel_auto']
    assert d['n_gpus'] == 2 and d['scaling'] == 'strong' and d['value'] > 0 and d['config']['ranks_seen'] == 2, d
    assert 'row-partition' in d['config']['parallelism'] and est['chosen'] == 'partition'
    assert est['predicted_partitioned_step_us'] > 0 and est['single_gpu_step_us'] > 0 and 0 < est['layer1_share_of_step'] < 1
    assert d['extras']['iters_per_s_independent_replicas'] > 0
    # (round 6) ... and the planner's model of the WHOLE curve from this one run: halo bytes and predicted step at N = 2 / 4 / 8
    curve = d['config']['planner_curve']['per_world']
    assert set(curve) == {'2', '4', '8'}, d['config']['planner_curve']
    for w_, rec in curve.items():
        assert len(rec['recv_bytes_per_step_every_rank']) == int(w_) and rec['predicted_step_us'] > 0 and rec['pair_bytes_per_step_max'] > 0
    # its N = 2 halo bytes are the measured engine's own
    assert curve['2']['recv_bytes_per_step_max_rank'] == max(v[0] for v in d['config']['halo_recv_send_bytes_per_rank'])


def test_bench_node_deletion_workload_line():
    """BASELINE config 5 as a bench line: `--workload synth-small-nodecls --gnn gat` - delete_node.py's request on
    the node-classification stand-in (out_dim = #classes = 4, padded inside the engine), the JSON contract with roofline,
    cpu_baseline (the oracle's epoch loop) and HIP-vs-oracle parity after the same iterations."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--workload', 'synth-small-nodecls', '--gnn', 'gat', '--steps', '8',
                        '--warmup', '2', '--cpu_baseline_iters', '3'], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = _json_line(r.stdout)
    assert d['n_gpus'] == 1 and d['unit'] == 'iters/s' and d['dtype'] == 'f32' and d['value'] > 0 and d['vs_baseline'] is None
    assert 'NODE deletion' in d['config']['workload'] and d['config']['out_dim'] == 4 and d['config']['out_dim_padded_to'] in (32, 64)
    assert {'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'} <= set(d['roofline'])
    assert d['cpu_baseline']['kind'] == 'port' and d['cpu_baseline']['value'] > 0 and d['cpu_baseline']['cores'] >= 1
    assert d['parity']['W_D1_rel_l2'] < 1e-4 and d['parity']['W_D2_rel_l2'] < 1e-4
    assert d['parity']['z1_affected_rel_l2'] < 1e-4 and d['parity']['z2_affected_rel_l2'] < 1e-4
    assert d['extras']['iters_per_s_trainer_default'] > 0 and d['extras']['iters_per_s_unpadded_class_dimension'] > 0


def test_bench_failing_ranks_give_a_nonzero_exit():
    """The self-launch returns the ranks' exit code: an impossible request must not look like a finished run."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(GD_BENCH_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--workload', 'no-such-workload'] + SMALL[2:], cwd=ROOT,
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and '{"metric"' not in r.stdout


def test_bench_eight_ranks_on_one_gpu():
    """The world size the scaling curve is asked for (1 / 2 / 4 / 8): planner, rendezvous ports, control group, the partition
    of a small request into eight row blocks with their halos - eight ranks over gloo on the box's one GPU,
    `bench.py --gpus 8` launching them itself.  (No statement about speed: eight processes share one device.)"""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(GD_BENCH_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0', OMP_NUM_THREADS='4')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--parallel', 'partition', '--workload', 'synth-small',
                        '--steps', '4', '--warmup', '1', '--pretrain_epochs', '0', '--no_cpu_baseline', '--no_cached_rate'], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    d = _json_line(r.stdout)
    assert d['n_gpus'] == 8 and d['scaling'] == 'strong' and d['config']['ranks_seen'] == 8 and d['value'] > 0, d
    assert len(d['config']['halo_recv_send_bytes_per_rank']) == 8 and 'partition_fallback' not in d['config'], d['config']


def test_bench_two_ranks_partition_the_rgcn_request():
    """BASELINE config 4 over two ranks (gloo on the one-GPU box): the R-GCN request's target rows partitioned with typed
    halos - `bench.py --gnn rgcn --gpus 2` as the driver launches it."""
    env = dict(os.environ, GD_BENCH_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--parallel', 'partition', '--gnn', 'rgcn', '--workload',
           'synth-kg-small', '--df_size', '2.5', '--steps', '6', '--warmup', '2', '--no_cpu_baseline']
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    d = _json_line(r.stdout)
    assert d['n_gpus'] == 2 and d['scaling'] == 'strong' and d['config']['ranks_seen'] == 2, d
    assert 'row-partition' in d['config']['parallelism'] and d['config']['halo']['recv_bytes_per_step'] > 0
    assert len(d['config']['halo_recv_send_bytes_per_rank']) == 2 and d['value'] > 0
