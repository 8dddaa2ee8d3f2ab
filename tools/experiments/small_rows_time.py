"""Lab: the Del-sized launches of the knowledge-graph step (15,434 selected rows of 93,773) back to back - how long do the
row GEMM with saved input + sign bits, the loss-fused weight gradient and the fused Del-2 kernel take when the launch has
~480 row tiles (fewer than two per compute unit)?"""
import os, sys, torch
sys.path.insert(0, '.')
from gnndelete_amd import ops, _lib
from gnndelete_amd._lib import ptr, check, stream_ptr
dev = 'cuda'
torch.manual_seed(0)
n, s = 93773, int(os.environ.get('S', 15434))
idx = torch.sort(torch.randperm(n, device=dev)[:s]).values.to(torch.int32)

def timed(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3

for d_in, d_out in ((128, 128), (64, 64), (128, 64), (64, 128)):
    x = torch.randn(n, d_in, device=dev)
    w = torch.randn(d_in, d_out, device=dev) / d_in ** 0.5
    out = torch.empty(n, d_out, device=dev)
    sav = torch.empty(s, d_in, device=dev)
    bits = torch.zeros(s, (d_out + 31) // 32, dtype=torch.int32, device=dev)
    print(f'{d_in}->{d_out} rows {s}: plain {timed(lambda: ops.rows_gemm(x, idx, w, out=out)):.1f} us, '
          f'save_in+signs {timed(lambda: ops.rows_gemm(x, idx, w, out=out, save_in=sav, sign_bits=bits)):.1f} us', end='')
    if d_in == d_out:
        z = x.clone()
        print(f', in place {timed(lambda: ops.rows_gemm(z, idx, w, out=z, save_in=sav, sign_bits=bits)):.1f} us', end='')
    print()

# the weight-gradient launch (partial products only, as the step runs it: dw = NULL) and the fused Del-2 kernel
for d in (128, 64):
    a = torch.randn(n, d, device=dev); g = torch.randn(n, d, device=dev)
    ws = torch.empty(max(1, _lib.lib().gd_rows_gemm_wgrad_workspace(s, d, d)), device=dev)
    f = lambda: check(_lib.lib().gd_rows_gemm_wgrad_f32(ptr(a), a.stride(0), ptr(idx), ptr(g), g.stride(0), ptr(idx), None, None, s, d, d,
                                                        None, 0, ptr(ws), stream_ptr(a.device)), 'wgrad')
    print(f'wgrad {d}x{d} rows {s}: {timed(f):.1f} us (partials only)')
d = 64
p = torch.randn(n, d, device=dev); w = torch.randn(d, d, device=dev) / 8
slot = torch.arange(s, dtype=torch.int32, device=dev); tm = torch.randn(s, d, device=dev)
coef = torch.rand(s, device=dev); cnt = torch.ones(s, device=dev)
dz2 = torch.zeros(n, d, device=dev)
nb = _lib.lib().gd_rows_gemm_wgrad_blocks(s)
lp = torch.zeros(2 * nb, device=dev); ws2 = torch.empty(max(1, _lib.lib().gd_rows_gemm_wgrad_workspace(s, d, d)), device=dev)
f = lambda: check(_lib.lib().gd_del_loss_bwd_wgrad_f32(ptr(p), p.stride(0), ptr(idx), s, ptr(w), d, ptr(slot), ptr(tm), ptr(coef), ptr(cnt), None, d,
                                                       ptr(dz2), dz2.stride(0), ptr(lp), ptr(ws2), stream_ptr(p.device)), 'del2')
print(f'fused Del-2 (64) rows {s}: {timed(f):.1f} us')
