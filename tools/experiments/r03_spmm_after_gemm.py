"""Why do the d = 64 aggregations take 67-73 us inside the step and 54-56 us back to back?  Time the layer-2 SpMM (events around
each launch) (a) after another SpMM, (b) after the t2 row GEMM that precedes it in the step, (c) after a streaming copy of
the same size as the GEMM's traffic, (d) after the GEMM plus an idle gap of ~30 us."""
import sys
sys.path.insert(0, '.')
import torch
import bench
from gnndelete_amd import ops
sys.argv = ['bench.py']
args = bench.parse()
dev = torch.device('cuda', 0)
data, model, neg, ni1, ni2 = bench.build_request(args, dev)
eng = bench.make_engine(args, data, model, neg, ni1, ni2, dev)
eng.step(); torch.cuda.synchronize()
g, n = eng.graph, eng.n
c2 = eng.model.conv2
t2 = torch.randn(n, 64, device=dev); y = torch.empty_like(t2)
big_a = torch.randn(n, 128, device=dev); big_b = torch.empty_like(big_a)


def spmm():
    ops._spmm_raw(g.rowptr, g.col, g.val, t2, c2.bias, 0.0, n, g.plan, out=y)


def gemm():
    eng._linear_relu_z1(c2.lin.weight)


def copy():
    big_b.copy_(big_a)


def idle():
    torch.cuda._sleep(60000)          # ~30 us of spinning on one CU


def measure(before, reps=40):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for _ in range(5):
        for f in before: f()
        spmm()
    torch.cuda.synchronize()
    for a, b in ev:
        for f in before: f()
        a.record(); spmm(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return ts[len(ts) // 2], ts[2], ts[-3]


for name, before in (('after another SpMM', [spmm]), ('after the t2 row GEMM', [gemm]), ('after a 121 MB copy', [copy]),
                     ('after GEMM + ~30 us idle', [gemm, idle]), ('after 2 x GEMM', [gemm, gemm])):
    med, lo, hi = measure(before)
    print(f'{name:28s}: median {med:6.1f} us  (p5 {lo:6.1f}, p95 {hi:6.1f})', flush=True)
