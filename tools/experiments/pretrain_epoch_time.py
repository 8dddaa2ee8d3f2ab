import sys, time, torch
sys.path.insert(0, '/root/repo')
sys.argv = ['bench.py', '--no_cpu_baseline', '--pretrain_epochs', '0']
import bench
args = bench.parse()
dev = torch.device('cuda', 0)
data, model, neg, ni1, ni2 = bench.build_request(args, dev)
bench.train_backbone(model, data, dev, 3)
torch.cuda.synchronize()
t0 = time.perf_counter()
bench.train_backbone(model, data, dev, 10)
torch.cuda.synchronize()
print('original-model training epoch at collab size: %.2f ms' % ((time.perf_counter() - t0) / 10 * 1e3))
