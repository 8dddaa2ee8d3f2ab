import sys, time, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from types import SimpleNamespace
import bench
from helpers import oracle_runner
dev = torch.device('cuda')
args = SimpleNamespace(workload='synth-collab', gnn='gcn', df='in', df_size=5.0, seed=42)
data, model, neg, ni1, ni2 = bench.build_request(args, dev)
state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
for dtype in (torch.float32, torch.float64):
    step, snap, _ = oracle_runner('gcn', data, state, neg, ni1, ni2, dtype, dev)
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(20): step()
    torch.cuda.synchronize(); print(dtype, 'alone', (time.time() - t0) / 20, 'max mem GB', torch.cuda.max_memory_allocated() / 2**30)
s2 = [oracle_runner('gcn', data, state, neg, ni1, ni2, torch.float32, dev, perm=p) for p in (1, 2)]
for _ in range(3):
    for s in s2: s[0]()
torch.cuda.synchronize(); t0 = time.time()
for _ in range(20):
    for s in s2: s[0]()
torch.cuda.synchronize(); print('two interleaved, per oracle epoch', (time.time() - t0) / 40, 'max mem GB', torch.cuda.max_memory_allocated() / 2**30)
