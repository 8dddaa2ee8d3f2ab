import sys, time, os, torch
sys.path.insert(0, '.')
import bench
from types import SimpleNamespace
args = SimpleNamespace(workload='synth-collab', seed=42, df='in', df_size=5.0, gnn='gcn', loss_type='both_layerwise')
data, model, neg, ni1, ni2 = bench.build_request(args, None)
state = {k: v.clone() for k, v in model.state_dict().items()}
print('cpus', os.cpu_count())
for th in (16, 32, 64, 128):
    os.cpu_count_orig = os.cpu_count
    os.cpu_count = lambda th=th: th
    t = time.time()
    r = bench.cpu_baseline(args, data, state, neg, 1)
    print(th, r['value'], r['sample'], 'total', time.time() - t, flush=True)
