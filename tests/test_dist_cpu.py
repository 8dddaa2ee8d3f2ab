"""world_size-2 (and 3) runs over gloo on CPU: collectives, row-partition planner and the
algebra of the partitioned Del step (dense emulation) vs single-process autograd."""
import os
import subprocess
import sys

import pytest

from helpers import free_port

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def launch(mode, world, timeout=600, **extra_env):
    env = {**os.environ, 'PYTHONPATH': ROOT, 'OMP_NUM_THREADS': '2', **extra_env}
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={world}',
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()),
           os.path.join(ROOT, 'tests', 'dist_worker.py'), mode]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0 and 'DIST_OK' in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    return r.stdout


@pytest.mark.parametrize('world', [2, 3, 8])
def test_partitioned_step_algebra_and_collectives_over_gloo(world):
    """world 8 = the node the scaling curve is asked for: planner tiling, halo lists and the segment algebra with eight
    row blocks (one thread per rank: the container has eight cores)."""
    launch('cpu', world, OMP_NUM_THREADS='1' if world > 4 else '2')


@pytest.mark.parametrize('world', [2, 3, 8])
def test_typed_partition_of_the_rgcn_graph_over_gloo(world):
    """The R-GCN row partition (typed graph restricted to a rank's rows + halo lists of the union graph) at 2, 3 and 8 ranks
    on CPU: forward and input gradient of the own rows against the whole-graph oracle (tests/dist_worker.py rgcn_cpu_checks)."""
    out = launch('rgcn_cpu', world, OMP_NUM_THREADS='1')
    assert f'typed partition x{world}' in out


@pytest.mark.gpu
@pytest.mark.parametrize('world', [2, 3])
def test_partitioned_engine_matches_single_gpu_engine(world):
    """Ranks sharing cuda:0 over gloo (the box has one GPU): the real partitioned HIP engine (fused stages, halo
    exchanges packed / unpacked inside the hipGraph segments) vs the single-GPU engine, GCN / GIN / GraphSAGE / GAT, every
    --loss_type (two ranks; GCN both_layerwise at three), each partitioned run with synchronous and with overlapped
    exchanges (bit-identical)."""
    out = launch('gpu', world, timeout=900)
    assert out.count('partitioned == single') == (9 if world == 2 else 1)


@pytest.mark.gpu
def test_partitioned_engine_over_rccl_in_a_world_of_one():
    """The box has one GPU and RCCL refuses two ranks on one device, so the multi-rank runs above use gloo.  This one
    drives the REAL data-path calls - all_to_all_single with per-peer split lists and the packed all_reduce on an
    `nccl` (= RCCL) group, between the hipGraph segments - in a world of one (GD_FORCE_COLLECTIVES=1 keeps the
    engine from skipping them), and checks the partitioned engine against the single-GPU engine."""
    out = launch('rccl1', 1, timeout=600, GD_FORCE_COLLECTIVES='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    assert out.count('partitioned == single') == 3


@pytest.mark.gpu
def test_partitioned_engine_over_the_library_owned_rccl_communicator():
    """The same in a world of one over gd_comm_init / gd_allreduce_f32 / gd_exchange_rows_f32 (collectives.DirectComm): the
    C-ABI collectives of include/gnndelete_hip.h driven by the engine between its hipGraph segments, plus the raw calls
    (an all-reduce over one rank is the identity, a one-peer exchange a copy)."""
    out = launch('direct1', 1, timeout=600, GD_FORCE_COLLECTIVES='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    assert out.count('partitioned == single') == 3 and 'direct collectives ok' in out


@pytest.mark.gpu
@pytest.mark.parametrize('world', [2])
def test_partitioned_rgcn_engine_matches_single_gpu_engine(world):
    """BASELINE config 4's model over two ranks sharing cuda:0 (gloo; three ranks time-share the one GPU at 4 x the time): target rows partitioned, typed graph restricted
    to the own rows, h-wide halo rows forward and o-wide backward, against the single-GPU fused R-GCN step."""
    out = launch('rgcn', world, timeout=900)
    assert out.count('partitioned == single') == 3
