# Can two ranks share the single GPU of the test box over RCCL?  (expected: no - duplicate device)
cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cat > /tmp/rccl_try.py <<'PY'
import os, torch, torch.distributed as dist, datetime
torch.cuda.set_device(0)
dist.init_process_group('nccl', timeout=datetime.timedelta(seconds=40))
t = torch.ones(4, device='cuda') * (dist.get_rank() + 1)
dist.all_reduce(t)
torch.cuda.synchronize()
print('rank', dist.get_rank(), 'allreduce ok', t.tolist())
PY
timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 /tmp/rccl_try.py 2>&1 | tail -12
