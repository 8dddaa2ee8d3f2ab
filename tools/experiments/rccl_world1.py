import os, torch, torch.distributed as dist, datetime
os.environ.setdefault('MASTER_ADDR','127.0.0.1'); os.environ.setdefault('MASTER_PORT','29533')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, timeout=datetime.timedelta(seconds=60), device_id=torch.device('cuda',0))
t = torch.ones(1024, device='cuda')
dist.all_reduce(t); torch.cuda.synchronize(); print('allreduce ok', float(t.sum()))
s = torch.arange(12, device='cuda', dtype=torch.float32).view(6,2); r = torch.empty_like(s)
dist.all_to_all_single(r, s, [6], [6]); torch.cuda.synchronize(); print('a2a ok', bool((r==s).all()))
e = torch.empty(0,2, device='cuda'); dist.all_to_all_single(torch.empty(0,2,device='cuda'), e, [0], [0]); torch.cuda.synchronize(); print('empty a2a ok')
g = torch.cuda.CUDAGraph()
x = torch.ones(8, device='cuda')
try:
    with torch.cuda.graph(g):
        dist.all_reduce(x)
    g.replay(); torch.cuda.synchronize(); print('graph-captured allreduce ok', x.tolist()[:2])
except Exception as ex:
    print('graph capture of allreduce failed:', type(ex).__name__, str(ex)[:200])
dist.destroy_process_group()
