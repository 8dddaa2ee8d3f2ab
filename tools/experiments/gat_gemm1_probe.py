import sys, torch
sys.path.insert(0, '/root/repo')
import os
sys.argv = ['bench.py', '--gnn', os.environ.get('GNN', 'gat'), '--no_cpu_baseline', '--pretrain_epochs', '0']
import bench
args = bench.parse()
dev = torch.device('cuda', 0)
data, model, neg, ni1, ni2 = bench.build_request(args, dev)
eng = bench.make_engine(args, data, model, neg, ni1, ni2, dev)
from gnndelete_amd import ops
c = eng.model.conv1
w = c.lin_src.weight if hasattr(c, 'lin_src') else c.lin.weight
def t(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
out = torch.empty(eng.n, 128, device=dev)
print('gemm1 alone', t(lambda: ops.rows_gemm(eng.x, None, w, trans_w=True, const_w=True, out=out)))
print('w stats', float(w.abs().mean()), float(eng.x.abs().mean()))
for _ in range(3): eng.step()
torch.cuda.synchronize()
print('step', t(lambda: eng.step(), 20))
print('x ptr align', eng.x.data_ptr() % 256, eng.x.stride(0), 'w', w.shape, w.data_ptr() % 256)
