import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _release_gpu_memory_between_tests():
    """Full-size tests hold several oracles with their autograd tapes: drop what a finished (or failed) test left behind."""
    yield
    import gc
    import torch
    gc.collect()
    if torch.cuda.is_available():
        torch.cuda.empty_cache()


def pytest_collection_finish(session):
    """When the session runs the long GPU parity tests, start their CPU-oracle legs NOW as child processes (tests/oracle_jobs.py,
    tests/test_long_parity_gpu.prepare): they compute on the host while the tests in front of them use the GPU, and the long
    tests only collect the results (the suite against the driver's time limit).  Single-test runs start
    their own leg when they reach it."""
    import torch
    if not torch.cuda.is_available() or session.config.option.collectonly:
        return
    ids = [it.nodeid for it in session.items if 'gpu' in it.keywords]
    if len(ids) < 40:                        # (a targeted run: nothing to overlap with)
        return
    try:
        import oracle_jobs
        wanted = {'full-collab-gcn': 'test_full_size_training_parity[synth-collab-gcn', 'full-collab-sage': 'test_full_size_training_parity[synth-collab-sage',
                  'full-collab-gat': 'test_full_size_training_parity[synth-collab-gat', 'full-nodecls-gat': 'test_full_size_node_deletion_gat'}
        names = [job for job, needle in wanted.items() if any(needle in i for i in ids)]
        members = []
        import test_long_parity_gpu as LP
        for gnn in ('gcn', 'gat'):
            if any(f'fp32_oracle[synth-small-{gnn}]' in i for i in ids):
                LP.prepare('synth-small', gnn, queue=members)
        oracle_jobs.prefetch(names, members)          # ONE background worker, oracle_jobs.BG_THREADS threads, in this order
    except Exception as e:                   # noqa: BLE001  (a failed prefetch must not fail the session: the tests start their own legs)
        print(f'conftest: prefetch of the CPU-oracle legs skipped ({type(e).__name__}: {e})')
