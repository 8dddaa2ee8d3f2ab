"""One CPU member of the fp32 ensemble of tests/test_long_parity_gpu.py, run as a CHILD PROCESS so that the members train
concurrently with each other and with the GPU runs of the test (in-process threads were slower than running them one
after the other: the oracle's small host-side ops serialise on the interpreter).  argv: request file, edge-order seed
('none' = as given), output file.  Writes the member's snapshots (W_D1, W_D2, z1[S1], z2[S2]) at the check epochs."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import torch  # noqa: E402


def run(req_path, perm, out_path, threads=None):
    """perm: an edge-order seed, None / 'none' = the edge lists as given."""
    from gnndelete_amd.framework.data import Data
    from helpers import oracle_runner
    req = torch.load(req_path, weights_only=False)
    n_threads = int(threads or os.environ.get('OMP_NUM_THREADS', '8'))
    data = Data(req['data'])
    step, snap, _ = oracle_runner(req['gnn'], data, req['state'], req['neg'], req['ni1'], req['ni2'], torch.float32, torch.device('cpu'),
                                  req['loss_type'], req['alpha'], req['lr'], perm=None if perm in (None, 'none') else int(perm))
    torch.set_num_threads(n_threads)            # (after oracle_runner, which sets its own default)
    snaps, done = [], 0
    for upto in req['check']:
        for _ in range(upto - done):
            step()
        done = upto
        snaps.append(tuple(snap()[:4]))
    torch.save(snaps, out_path)


if __name__ == '__main__':
    run(sys.argv[1], sys.argv[2], sys.argv[3])
