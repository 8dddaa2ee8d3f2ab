import torch, ctypes, os, sys
sys.path.insert(0, '.')
from gnndelete_amd import _lib
from gnndelete_amd.collectives import DirectComm
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
try:
    c = DirectComm(0, 1, dev)
    buf = torch.arange(100, dtype=torch.float32, device=dev)
    c.all_reduce(buf)
    s = torch.randn(5, 64, device=dev); r = torch.zeros(5, 64, device=dev)
    c.exchange(s, r, [5], [5])
    torch.cuda.synchronize()
    print('ok', torch.equal(r, s), float(buf.sum()))
except Exception as e:
    print('ERR', repr(e))
