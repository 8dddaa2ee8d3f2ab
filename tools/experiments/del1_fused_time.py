"""Lab: gd_del1_loss_wgrad_f32 back to back at the bench request's size (178,921 of 235,868 rows), with knock-outs
(GD_DEL1_DBG bits: 1 no target / g_add fetch after the first, 2 no z stores, 4 no P3, 16 no row fetch after the first two)."""
import os, sys, torch
sys.path.insert(0, '.')
from gnndelete_amd import ops, _lib
from gnndelete_amd._lib import ptr, check, stream_ptr
dev = 'cuda'
torch.manual_seed(0)
n, s, d = 235868, 178921, 128
idx = torch.sort(torch.randperm(n, device=dev)[:s]).values.to(torch.int32)
p = torch.randn(n, d, device=dev); w = (torch.eye(d, device=dev) + 0.05 * torch.randn(d, d, device=dev)).contiguous()
slot = torch.arange(s, dtype=torch.int32, device=dev); tm = torch.randn(s, d, device=dev)
coef = torch.rand(s, device=dev) * 1e-3; cnt = torch.ones(s, device=dev)
g_add = torch.randn(n, d, device=dev) * 1e-3
z = torch.zeros(n, d, device=dev); bits = torch.zeros(s, 4, dtype=torch.int32, device=dev)
lib = _lib.lib()
nb = lib.gd_rows_gemm_wgrad_blocks(s)
lp = torch.zeros(2 * nb, device=dev); ws = torch.empty(max(1, lib.gd_rows_gemm_wgrad_workspace(s, d, d)), device=dev)

def timed(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3

def fused(ga):
    return lambda: check(lib.gd_del1_loss_wgrad_f32(ptr(p), p.stride(0), ptr(idx), s, ptr(w), d, ptr(z), z.stride(0), ptr(bits), ptr(slot), ptr(tm),
                                                    ptr(coef), ptr(cnt), ptr(ga), d if ga is not None else 0, ptr(lp), ptr(ws), int(os.environ.get('PARTS', nb)), stream_ptr(p.device)), 'fused')
print(f'fused with g_add {timed(fused(g_add)):.1f} us, without {timed(fused(None)):.1f} us')
# cache-resident operands: every unit reads the same 16 rows / targets (what the kernel costs without memory)
idx_keep, slot_keep = idx.clone(), slot.clone()
idx.copy_((torch.arange(s, device=dev) % 16).to(torch.int32)); slot.copy_((torch.arange(s, device=dev) % 16).to(torch.int32))
print(f'cache-resident rows: fused with g_add {timed(fused(g_add)):.1f} us, without {timed(fused(None)):.1f} us')
idx.copy_(idx_keep); slot.copy_(slot_keep)
if not os.environ.get('GD_DEL1_DBG'):
    z2 = torch.zeros(n, d, device=dev)
    t1 = timed(lambda: ops.rows_gemm(p, idx, w, out=z2, sign_bits=bits))
    t2 = timed(lambda: check(lib.gd_rows_gemm_wgrad_loss_f32(ptr(p), p.stride(0), ptr(idx), ptr(z2), z2.stride(0), ptr(idx), ptr(slot), ptr(tm), ptr(coef), ptr(cnt),
                                                             ptr(g_add), s, d, d, None, 0, ptr(ws), ptr(lp), None, None, None, None, 0.0, 0.0, 0.0, 0.0, stream_ptr(p.device)), 'wl'))
    print(f'two launches: Del-1 {t1:.1f} us + loss-fused weight gradient {t2:.1f} us')

if os.environ.get('GNNDELETE_HIP_LIB'):      # lab build with -DGD_DEL1_TRACE: cycle stamps of block 0 / wave 0
    import ctypes, numpy as np
    h = ctypes.CDLL(os.environ['GNNDELETE_HIP_LIB'])
    fused(g_add)(); torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 128)()
    print('trace rc', h.gd_lab_del1_trace(buf))
    t = np.array(list(buf), dtype=np.int64).reshape(16, 8)[:, :5]
    t0 = t[0, 0]
    print('unit: top->P1 issued | ->loss done | ->fetch issued+barrier | ->P3 issued | next top   (cycles)')
    for k in range(12):
        nxt = t[k + 1, 0] - t[k, 4] if k + 1 < 16 else 0
        print(k, t[k, 0] - t0, '|', t[k, 1] - t[k, 0], t[k, 2] - t[k, 1], t[k, 3] - t[k, 2], t[k, 4] - t[k, 3], nxt)
