import torch, sys
sys.path.insert(0, '.')
from gnndelete_amd import ops
from gnndelete_amd.graph import build_csr
n=6
ei=torch.tensor([[0,1,2,3],[1,2,3,4]])
g=build_csr(ei.cuda(), n, 'gcn')
print('rowptr', g.rowptr.tolist()); print('col', g.col.tolist()); print('val', g.val.tolist())
for d in (128, 64, 4, 7):
    x=torch.arange(n).float()[:,None].repeat(1,d).contiguous().cuda()
    y=ops.spmm(x,g)
    print(d, y[:,0].tolist(), y[:,-1].tolist())
g2=build_csr(ei.cuda(), n, 'sum')
print(ops.spmm(torch.ones(n,128).cuda(), g2)[:,0].tolist())
