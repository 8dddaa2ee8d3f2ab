"""HIP kernels (through the C ABI / ctypes) vs the CPU oracle and closed-form fp64 math.
Tolerance: fp32 kernels, rel-L2 <= 1e-5 against fp64 closed forms (north_star allows 1e-4)."""
import pytest
import os

import numpy as np
import torch

from helpers import random_graph, rel_l2

pytestmark = pytest.mark.gpu
TOL = 1e-5


def dense_adj(ei, n):
    a = torch.zeros(n, n, dtype=torch.float64)
    a.index_put_((ei[1], ei[0]), torch.ones(ei.shape[1], dtype=torch.float64), accumulate=True)
    return a


@pytest.mark.parametrize('n,m,d', [(50, 200, 128), (50, 200, 64), (33, 90, 16), (70, 400, 256), (40, 100, 4),
                                   (25, 60, 7), (30, 0, 32), (64, 3000, 128), (20, 50, 1639), (300, 9000, 512)])
def test_spmm_gcn_matches_dense_closed_form(n, m, d):
    from gnndelete_amd import ops
    from gnndelete_amd.graph import build_csr
    ei = random_graph(n, m, seed=n + d, isolate=3)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, d, generator=g)
    b = torch.randn(d, generator=g)
    a = dense_adj(ei[:, ei[0] != ei[1]], n) + torch.eye(n, dtype=torch.float64)
    deg = a.sum(1)
    want = (a / deg.sqrt()[:, None] / deg.sqrt()[None, :]) @ x.double() + b.double()
    gr = build_csr(ei.cuda(), n, 'gcn')
    assert gr.nnz == int((ei[0] != ei[1]).sum()) + n
    xg = x.cuda().requires_grad_(True)
    y = ops.spmm(xg, gr, b.cuda())
    assert rel_l2(y.detach().cpu(), want) < TOL
    up = torch.randn(n, d, generator=g)
    y.backward(up.cuda())
    want_dx = (a / deg.sqrt()[:, None] / deg.sqrt()[None, :]).t() @ up.double()
    assert rel_l2(xg.grad.cpu(), want_dx) < TOL


@pytest.mark.parametrize('d', [64, 128, 10])
def test_spmm_sum_with_self_term_is_gin_aggregate(d):
    from gnndelete_amd import ops
    from gnndelete_amd.graph import build_csr
    n = 41
    ei = random_graph(n, 150, seed=d)                      # loops and multi-edges are kept
    x = torch.randn(n, d, generator=torch.Generator().manual_seed(2))
    want = (dense_adj(ei, n) + torch.eye(n, dtype=torch.float64)) @ x.double()
    y = ops.spmm(x.cuda(), build_csr(ei.cuda(), n, 'sum'), None, 1.0)
    assert rel_l2(y.cpu(), want) < TOL


def test_spmm_is_bit_reproducible():
    from gnndelete_amd import ops
    from gnndelete_amd.graph import build_csr
    n = 500
    ei = random_graph(n, 20000, seed=9).cuda()
    x = torch.randn(n, 128, device='cuda')
    g = build_csr(ei, n, 'gcn')
    y0 = ops.spmm(x, g)
    for _ in range(3):
        assert torch.equal(ops.spmm(x, build_csr(ei, n, 'gcn')), y0)


def test_gcn_norm_matches_oracle():
    from gnndelete_amd.graph import build_csr
    from oracle import pyg_semantics as pyg
    n = 60
    ei = random_graph(n, 300, seed=4)
    g = build_csr(ei.cuda(), n, 'gcn')
    ei2, w = pyg.gcn_norm(ei, n)
    dense = torch.zeros(n, n).index_put_((ei2[1], ei2[0]), w, accumulate=True)
    rows = torch.repeat_interleave(torch.arange(n), (g.rowptr[1:] - g.rowptr[:-1]).cpu().long())
    got = torch.zeros(n, n).index_put_((rows, g.col.cpu().long()), g.val.cpu(), accumulate=True)
    assert rel_l2(got, dense) < 1e-6
    # transposed arrays describe the same matrix
    rows_t = torch.repeat_interleave(torch.arange(n), (g.rowptr_t[1:] - g.rowptr_t[:-1]).cpu().long())
    got_t = torch.zeros(n, n).index_put_((g.col_t.cpu().long(), rows_t), g.val_t.cpu(), accumulate=True)
    assert rel_l2(got_t, dense) < 1e-6


@pytest.fixture(params=[0, 6], ids=['f32-instruction', 'bf16x6-split'])
def matrix_split(request):
    """Both ways the row GEMMs form their fp32 products (gd_set_matrix_split): same tests, same tolerances."""
    from gnndelete_amd import ops
    before = ops.matrix_split()
    ops.set_matrix_split(request.param)
    yield request.param
    ops.set_matrix_split(before)


def test_split_products_are_as_accurate_as_the_fp32_instruction():
    """gd_set_matrix_split(6) on the Del operator's shape, inputs spread over six orders of magnitude: the error against
    an fp64 product is not larger than the fp32 matrix instruction's, and the two fp32 results agree to a few ulps of
    the row scale."""
    from gnndelete_amd import ops
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(4096, 128, generator=g) * torch.exp(3.0 * torch.randn(4096, 128, generator=g))).cuda()
    w = (torch.randn(128, 128, generator=g) * 0.1 * torch.exp(2.0 * torch.randn(128, 128, generator=g))).cuda()
    want = x.double() @ w.double()
    before = ops.matrix_split()
    try:
        got = {}
        for mode in (0, 6):
            ops.set_matrix_split(mode)
            assert ops.matrix_split() == mode
            got[mode] = ops.rows_gemm(x, None, w)
    finally:
        ops.set_matrix_split(before)
    err = {m: float(((got[m].double() - want).norm() / want.norm())) for m in got}
    assert err[6] <= 1.2 * err[0] + 1e-9 and err[0] < 5e-7, err
    assert float((got[6] - got[0]).abs().max() / want.abs().max()) < 2e-6
    with pytest.raises(Exception):
        ops.set_matrix_split(3)


@pytest.mark.parametrize('n,d_in,d_out,frac', [(200, 128, 128, 0.4), (200, 64, 64, 0.5), (77, 128, 64, 1.0),
                                               (50, 32, 32, 0.3), (40, 12, 12, 0.5), (64, 128, 4, 0.5),
                                               (90, 64, 96, 0.7), (33, 128, 128, 0.0), (5000, 128, 128, 0.9)])
def test_rows_gemm_forward_transpose_and_inplace(n, d_in, d_out, frac, matrix_split):
    from gnndelete_amd import ops
    g = torch.Generator().manual_seed(n + d_in)
    x = torch.randn(n, d_in, generator=g)
    w = torch.randn(d_in, d_out, generator=g) * 0.2
    b = torch.randn(d_out, generator=g)
    mask = torch.rand(n, generator=g) < frac
    idx = mask.nonzero().flatten().int().cuda()
    want = x.double()[mask] @ w.double()
    out = torch.full((n, d_out), 7.0, device='cuda')
    save = torch.empty(int(mask.sum()), d_in, device='cuda')
    ops.rows_gemm(x.cuda(), idx, w.cuda(), out=out, save_in=save)
    assert rel_l2(out.cpu()[mask], want) < TOL or want.numel() == 0
    assert torch.all(out.cpu()[~mask] == 7.0)                          # untouched rows stay untouched
    assert torch.equal(save.cpu(), x[mask])
    # transposed weight + bias + relu on the input
    wt = w.t().contiguous()
    out2 = ops.rows_gemm(x.cuda(), idx, wt.cuda(), trans_w=True, bias=b.cuda(), relu_in=True,
                         out=torch.zeros(n, d_out, device='cuda'))
    want2 = x.double().clamp(min=0)[mask] @ w.double() + b.double()
    assert rel_l2(out2.cpu()[mask], want2) < TOL or want2.numel() == 0
    # dense (idx = None)
    out3 = ops.rows_gemm(x.cuda(), None, w.cuda())
    assert rel_l2(out3.cpu(), x.double() @ w.double()) < TOL
    # packed sign pattern of the output, and the ReLU backward gated by such a pattern
    n_words = (d_out + 31) // 32
    bits = torch.zeros(max(1, int(mask.sum())), n_words, dtype=torch.int32, device='cuda')
    out4 = ops.rows_gemm(x.cuda(), idx, w.cuda(), out=torch.zeros(n, d_out, device='cuda'), sign_bits=bits)
    assert torch.equal(out4, out.where(torch.from_numpy(np.asarray(mask))[:, None].cuda(), torch.zeros((), device='cuda')))
    sel = out4.cpu()[mask]
    got_bits = (bits.cpu()[:sel.shape[0], :, None] >> torch.arange(32, dtype=torch.int32)) & 1
    assert torch.equal(got_bits.reshape(sel.shape[0], n_words * 32)[:, :d_out].bool(), sel > 0)
    out5 = ops.rows_gemm(x.cuda(), idx, (w * 1.5).cuda(), out=torch.zeros(n, d_out, device='cuda'), gate_bits=bits)
    assert rel_l2(out5.cpu()[mask], 1.5 * want * (sel > 0)) < TOL or want.numel() == 0
    if d_in == d_out:                                                   # in place (the Del operator)
        z = x.clone().cuda()
        ops.rows_gemm(z, idx, w.cuda(), out=z)
        ref = x.double().clone()
        ref[mask] = want
        assert rel_l2(z.cpu(), ref) < TOL


@pytest.mark.parametrize('d_in,d_out', [(128, 128), (128, 64), (64, 128), (64, 64)])
def test_rows_gemm_weight_stationary_form(d_in, d_out):
    """The register-resident-weight form (csrc/rows_gemm_ws.hip) that the Del operator's widths take from 65,536 rows up:
    every mode it covers - index list / dense, [k][n] / [n][k] weights, ReLU on the input, rows from two buffers, packed
    sign pattern out, gate pattern in - against fp64 products; n is not a multiple of the 16-row work unit; rows outside
    the index list stay untouched; what it does not cover (bias, saved input, in place) still runs (LDS-operand form)."""
    from gnndelete_amd import _lib, ops
    n = 80_003
    assert _lib.lib().gd_rows_gemm_ws_covers(n - 7000, d_in, d_out) == 1 and _lib.lib().gd_rows_gemm_ws_covers(5000, d_in, d_out) == 0
    g = torch.Generator().manual_seed(d_in + 3 * d_out)
    x = torch.randn(n, d_in, generator=g).cuda()
    x2 = torch.randn(n, d_in, generator=g).cuda()
    w = (torch.randn(d_in, d_out, generator=g) * 0.2).cuda()
    mask = torch.rand(n, generator=g) < 0.93
    mask[-1] = True
    if int(mask.sum()) % 16 == 0:
        mask[int(mask.nonzero()[0])] = False
    maskg = mask.cuda()
    idx = mask.nonzero().flatten().int().cuda()
    assert idx.numel() >= 65536 and idx.numel() % 16 != 0
    xd, wd = x.double(), w.double()
    want = xd @ wd
    # index list, out of place, untouched rows
    out = torch.full((n, d_out), 7.0, device='cuda')
    ops.rows_gemm(x, idx, w, out=out)
    assert rel_l2(out[maskg], want[maskg]) < TOL
    assert bool((out[~maskg] == 7.0).all())
    # dense, [n][k] weight, ReLU on the input
    out2 = ops.rows_gemm(x, None, w.t().contiguous(), trans_w=True, relu_in=True)
    assert rel_l2(out2, xd.clamp(min=0) @ wd) < TOL
    # rows from two buffers
    sel = (torch.rand(n, generator=g) < 0.4).cuda()
    out3 = ops.rows_gemm_select(x, x2, sel.to(torch.uint8), w, relu_in=True)
    assert rel_l2(out3, torch.where(sel[:, None], x2.double(), xd).clamp(min=0) @ wd) < TOL
    # rows from two buffers AND an index list (round 6: the rows-only step's t2 product; the selector byte of a listed row is a
    # dependent load behind its id, fetched a unit later): listed rows right, the others untouched, bit-reproducible
    out3i = ops.rows_gemm_select(x, x2, sel.to(torch.uint8), w, relu_in=True, idx=idx, out=torch.full((n, d_out), 7.0, device='cuda'))
    assert torch.equal(out3i[maskg], out3[maskg]) and bool((out3i[~maskg] == 7.0).all())
    few = idx[::3].contiguous()                       # below the weight-stationary form's threshold: the LDS-operand kernel
    out3f = ops.rows_gemm_select(x, x2, sel.to(torch.uint8), w, relu_in=True, idx=few, out=torch.full((n, d_out), 7.0, device='cuda'))
    assert few.numel() < 65536 and rel_l2(out3f[few.long()], out3[few.long()]) < TOL
    # accumulate mode (round 6, gd_rows_gemm_accumulate_f32: GraphSAGE's root term added onto the aggregated rows): dense and
    # on the index list, [k][n] and [n][k] weights; rows outside the list untouched; bit-reproducible; refused below the threshold
    base = torch.randn(n, d_out, generator=g).cuda()
    acc1 = ops.rows_gemm_accumulate_(base.clone(), x, None, w)
    assert rel_l2(acc1, base.double() + want) < TOL
    acc2 = ops.rows_gemm_accumulate_(base.clone(), x, idx, w.t().contiguous(), trans_w=True)
    assert rel_l2(acc2[maskg], (base.double() + want)[maskg]) < TOL and torch.equal(acc2[~maskg], base[~maskg])
    assert torch.equal(ops.rows_gemm_accumulate_(base.clone(), x, None, w), acc1)
    assert ops.rows_gemm_accumulate_ok(n, d_in, d_out) and not ops.rows_gemm_accumulate_ok(5000, d_in, d_out)
    from gnndelete_amd._lib import GnnDeleteHipError
    with pytest.raises(GnnDeleteHipError):
        ops.rows_gemm_accumulate_(base[:5000].clone(), x[:5000], None, w)
    # sign pattern out (bit b of word q = output 32 q + b), gate pattern in
    n_words = d_out // 32
    bits = torch.zeros(idx.numel(), n_words, dtype=torch.int32, device='cuda')
    out4 = ops.rows_gemm(x, idx, w, out=torch.zeros(n, d_out, device='cuda'), sign_bits=bits)
    assert torch.equal(out4[maskg], out[maskg])
    got_bits = ((bits[:, :, None] >> torch.arange(32, dtype=torch.int32, device='cuda')) & 1).reshape(idx.numel(), d_out).bool()
    assert torch.equal(got_bits, out4[maskg] > 0)
    out5 = ops.rows_gemm(x, idx, w * 1.5, out=torch.full((n, d_out), 3.0, device='cuda'), gate_bits=bits)
    assert rel_l2(out5[maskg], 1.5 * want[maskg] * (out4[maskg] > 0)) < TOL
    assert bool((out5[~maskg] == 3.0).all())
    # bit-reproducible
    assert torch.equal(ops.rows_gemm(x, idx, w, out=torch.full((n, d_out), 7.0, device='cuda')), out)
    # row dots from the epilogue (GAT's attention logits; behind a 128-wide input): dense, rows from two buffers + ReLU, index list
    u1, u2 = torch.randn(d_out, generator=g).cuda(), torch.randn(d_out, generator=g).cuda()
    wt = w.t().contiguous()
    if d_in == 128:
        assert ops.rows_gemm_dots_ok(d_in, d_out, n)
        o, a1, a2 = ops.rows_gemm_dots(x, wt, u1, u2)
        assert rel_l2(o, want) < TOL and rel_l2(a1, want @ u1.double()) < TOL and rel_l2(a2, want @ u2.double()) < TOL
        o_, a1_, a2_ = ops.rows_gemm_dots(x, wt, u1, u2)
        assert torch.equal(o, o_) and torch.equal(a1, a1_) and torch.equal(a2, a2_)
        o, a1, a2 = ops.rows_gemm_dots(x, wt, u1, u2, inp_alt=x2, sel=sel.to(torch.uint8), relu_in=True)
        w3 = torch.where(sel[:, None], x2.double(), xd).clamp(min=0) @ wd
        assert rel_l2(o, w3) < TOL and rel_l2(a1, w3 @ u1.double()) < TOL and rel_l2(a2, w3 @ u2.double()) < TOL
        dots_out = (torch.full((n,), 5.0, device='cuda'), torch.full((n,), 6.0, device='cuda'))
        o, a1, a2 = ops.rows_gemm_dots(x, wt, u1, u2, out=torch.full((n, d_out), 7.0, device='cuda'), idx=idx, dots_out=dots_out)
        assert rel_l2(o[maskg], want[maskg]) < TOL and bool((o[~maskg] == 7.0).all())
        assert rel_l2(a1[maskg], want[maskg] @ u1.double()) < TOL and rel_l2(a2[maskg], want[maskg] @ u2.double()) < TOL
        assert bool((a1[~maskg] == 5.0).all()) and bool((a2[~maskg] == 6.0).all())
        # ... and with rows from two buffers on top of the index list (the rows-only GAT step's layer-2 product)
        assert ops.rows_gemm_dots_ok(d_in, d_out, int(idx.numel()), selected=True)
        dots_out = (torch.full((n,), 5.0, device='cuda'), torch.full((n,), 6.0, device='cuda'))
        o, a1, a2 = ops.rows_gemm_dots(x, wt, u1, u2, inp_alt=x2, sel=sel.to(torch.uint8), relu_in=True,
                                       out=torch.full((n, d_out), 7.0, device='cuda'), idx=idx, dots_out=dots_out)
        assert rel_l2(o[maskg], w3[maskg]) < TOL and bool((o[~maskg] == 7.0).all())
        assert rel_l2(a1[maskg], w3[maskg] @ u1.double()) < TOL and rel_l2(a2[maskg], w3[maskg] @ u2.double()) < TOL
        assert bool((a1[~maskg] == 5.0).all()) and bool((a2[~maskg] == 6.0).all())
    # the gated product after a rank-1 correction (GAT's input gradient; in front of a 128-wide output)
    if d_out == 128:
        ra, rb = torch.randn(n, generator=g).cuda(), torch.randn(n, generator=g).cuda()
        o7 = ops.rows_gemm(x, idx, w, out=torch.full((n, d_out), 3.0, device='cuda'), gate_bits=bits, rank1=(ra, u1, rb, u2))
        w7 = (want + ra.double()[:, None] * u1.double() + rb.double()[:, None] * u2.double())[maskg] * (out4[maskg] > 0)
        assert rel_l2(o7[maskg], w7) < TOL and bool((o7[~maskg] == 3.0).all())
        assert torch.equal(ops.rows_gemm(x, idx, w, out=torch.full((n, d_out), 3.0, device='cuda'), gate_bits=bits, rank1=(ra, u1, rb, u2)), o7)
    # bias (the accumulators of a unit start from it): dense, index list + ReLU on the input
    b = torch.randn(d_out, generator=g).cuda()
    out6 = ops.rows_gemm(x, None, w, bias=b)
    assert rel_l2(out6, want + b.double()) < TOL
    out6i = ops.rows_gemm(x, idx, w, bias=b, relu_in=True, out=torch.full((n, d_out), 7.0, device='cuda'))
    assert rel_l2(out6i[maskg], (xd.clamp(min=0) @ wd + b.double())[maskg]) < TOL and bool((out6i[~maskg] == 7.0).all())
    # not covered by this form: (square widths) the in-place call
    if d_in == d_out:
        z = x.clone()
        ops.rows_gemm(z, idx, w, out=z)
        ref = xd.clone()
        ref[maskg] = want[maskg]
        assert rel_l2(z, ref) < TOL


@pytest.mark.parametrize('n,frac,with_add', [(90_000, 0.85, True), (70_001, 1.0, False), (3000, 0.6, True)])
def test_wgrad_with_loss_formed_in_the_fetch(n, frac, with_add):
    """gd_rows_gemm_wgrad_loss_f32 at 128 x 128: dW = a[ia]^T (coef_u (z[iz] - tbar_u) + g_add[iz]) and the two loss sums
    against fp64: rows without a loss slot, a row count that is not a multiple of the tile height, partial matrices counted
    by gd_rows_gemm_wgrad_blocks."""
    from gnndelete_amd import _lib
    from gnndelete_amd._lib import check, ptr
    L = _lib.lib()
    d = 128
    g = torch.Generator().manual_seed(n)
    a = torch.randn(n, d, generator=g)
    z = torch.randn(n, d, generator=g)
    add = torch.randn(n, d, generator=g) * 0.1
    mask = torch.rand(n, generator=g) < frac
    if int(mask.sum()) % 16 == 0:
        mask[int(mask.nonzero()[0])] = False
    idx = mask.nonzero().flatten()
    s = idx.numel()
    has = torch.rand(s, generator=g) < 0.8
    n_slots = int(has.sum())
    slot = torch.full((s,), -1, dtype=torch.int32)
    slot[has] = torch.arange(n_slots, dtype=torch.int32)[torch.randperm(n_slots, generator=g)]
    tm = torch.randn(n_slots, d, generator=g)
    coef = torch.rand(n_slots, generator=g) + 0.1
    cnt = torch.randint(1, 4, (n_slots,), generator=g).float() * torch.where(torch.rand(n_slots, generator=g) < 0.5, -1.0, 1.0)
    sl = slot[has].long()
    gm = torch.zeros(s, d, dtype=torch.float64)
    df = z.double()[idx][has] - tm.double()[sl]
    gm[has] = coef.double()[sl][:, None] * df
    if with_add:
        gm += add.double()[idx]
    dw_want = a.double()[idx].t() @ gm
    sq = (df * df).sum(1) * cnt.double()[sl].abs()
    want_s = [float(sq[cnt[sl] >= 0].sum()), float(sq[cnt[sl] < 0].sum())]
    nb = L.gd_rows_gemm_wgrad_blocks(s)
    ws = torch.full((nb * d * d,), 3.0, device='cuda')
    lp = torch.full((2 * nb,), 5.0, device='cuda')
    dev = lambda t_: t_.cuda()
    ag, zg, addg, ig = dev(a), dev(z), dev(add), dev(idx.int())
    sg, tg, cg, ng = dev(slot), dev(tm), dev(coef), dev(cnt)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(2):                      # twice: the second call overwrites every partial (no accumulation, no stale slot)
        check(L.gd_rows_gemm_wgrad_loss_f32(ptr(ag), d, ptr(ig), ptr(zg), d, ptr(ig), ptr(sg), ptr(tg), ptr(cg), ptr(ng),
                                            ptr(addg) if with_add else None, s, d, d, None, 0, ptr(ws), ptr(lp), None, None, None, None,
                                            1e-3, 0.9, 0.999, 1e-8, st), 'gd_rows_gemm_wgrad_loss_f32')
    dw = ws.view(nb, d, d).double().sum(0).cpu()
    assert rel_l2(dw, dw_want) < TOL
    np.testing.assert_allclose(lp.view(-1, 2).double().sum(0).cpu().numpy(), want_s, rtol=1e-5)


@pytest.mark.parametrize('n,d_a,d_b,frac', [(300, 128, 128, 0.5), (300, 64, 64, 0.8), (100, 128, 64, 1.0),
                                            (50, 12, 12, 0.5), (60, 128, 4, 0.6), (20, 64, 64, 0.0),
                                            (70000, 128, 128, 0.7)])
def test_rows_gemm_wgrad(n, d_a, d_b, frac):
    from gnndelete_amd import ops
    g = torch.Generator().manual_seed(n + d_b)
    a = torch.randn(n, d_a, generator=g)
    gr = torch.randn(n, d_b, generator=g)
    rm = torch.randn(n, d_b, generator=g)
    mask = torch.rand(n, generator=g) < frac
    idx = mask.nonzero().flatten().int().cuda()
    s = int(idx.shape[0])
    want = a.double()[mask].t() @ gr.double()[mask]
    got = ops.rows_gemm_wgrad(a.cuda(), idx, gr.cuda(), idx, s)
    assert rel_l2(got.cpu(), want) < TOL or s == 0
    # compact a, relu mask, accumulate
    ac = a[mask].contiguous().cuda()
    base = torch.randn(d_a, d_b, generator=g)
    got2 = ops.rows_gemm_wgrad(ac, None, gr.cuda(), idx, s, relu_mask=rm.cuda(), out=base.clone().cuda(),
                               accumulate=True)
    want2 = base.double() + a.double()[mask].t() @ (gr.double() * (rm > 0))[mask]
    assert rel_l2(got2.cpu(), want2) < TOL
    # deterministic split-K
    assert torch.equal(ops.rows_gemm_wgrad(a.cuda(), idx, gr.cuda(), idx, s), got)
    # second gradient source added after the mask: a^T (mask(g) + g_add)
    g2 = torch.randn(n, d_b, generator=g)
    got3 = ops.rows_gemm_wgrad(ac, None, gr.cuda(), idx, s, relu_mask=rm.cuda(), g_add=g2.cuda())
    want3 = a.double()[mask].t() @ (gr.double() * (rm > 0) + g2.double())[mask]
    assert rel_l2(got3.cpu(), want3) < TOL or s == 0


def test_del_rows_autograd_matches_reference_golden():
    from gnndelete_amd import ops
    from helpers import load_golden, t
    fx = load_golden('del_layer.npz')
    for tag in ['partial', 'empty', 'full', 'odd']:
        x = t(fx[f'{tag}::x']).cuda().requires_grad_(True)
        w = t(fx[f'{tag}::w']).cuda().requires_grad_(True)
        idx = t(fx[f'{tag}::mask']).nonzero().flatten().int().cuda()
        y = ops.del_rows(x, w, idx)
        y.backward(t(fx[f'{tag}::up']).cuda())
        assert rel_l2(y.detach().cpu(), fx[f'{tag}::y']) < TOL, tag
        assert rel_l2(x.grad.cpu(), fx[f'{tag}::gx']) < TOL, tag
        if idx.numel():
            assert rel_l2(w.grad.cpu(), fx[f'{tag}::gw']) < TOL, tag
        else:
            assert float(w.grad.abs().max()) == 0.0


@pytest.mark.parametrize('d', [128, 64, 16, 256])
def test_rowpair_mse_value_and_gradient(d):
    from gnndelete_amd import _lib
    g = torch.Generator().manual_seed(d)
    n, n_seg = 90, 40
    z = torch.randn(n, d, generator=g)
    o = torch.randn(n, d, generator=g)
    rows = torch.randperm(n, generator=g)[:n_seg].sort().values
    cnt = torch.randint(0, 4, (n_seg,), generator=g)                     # some rows have no terms
    seg_ptr = torch.zeros(n_seg + 1, dtype=torch.int32)
    seg_ptr[1:] = cnt.cumsum(0)
    T = int(cnt.sum())
    term_o = torch.randint(0, n, (T,), generator=g).int()
    term_w = torch.rand(T, generator=g)
    kind = torch.randint(0, 2, (T,), generator=g).int()
    zr = z.double().requires_grad_(True)
    seg_of = torch.repeat_interleave(torch.arange(n_seg), cnt)
    diff = zr[rows[seg_of]] - o.double()[term_o.long()]
    sq = (diff ** 2).sum(1)
    want_s = [float(sq[kind == 0].sum()), float(sq[kind == 1].sum())]
    (sq * term_w.double()).sum().backward()
    for compact in (0, 1):
        dz = torch.full((n_seg if compact else n, d), 3.0, device='cuda')
        sums = torch.zeros(2, device='cuda')
        ws = torch.empty(_lib.lib().gd_rowpair_mse_workspace(n_seg), device='cuda')
        zc, oc = z.cuda(), o.cuda()
        dev = [x.cuda() for x in (seg_ptr, rows.int(), term_o, term_w, kind)]
        _lib.check(_lib.lib().gd_rowpair_mse_f32(zc.data_ptr(), d, oc.data_ptr(), d, d, dev[0].data_ptr(),
                                                 dev[1].data_ptr(), n_seg, dev[2].data_ptr(), dev[3].data_ptr(),
                                                 dev[4].data_ptr(), dz.data_ptr(), d, compact, sums.data_ptr(),
                                                 ws.data_ptr(), torch.cuda.current_stream().cuda_stream))
        assert abs(float(sums[0]) - want_s[0]) < 1e-4 * max(1, want_s[0])
        assert abs(float(sums[1]) - want_s[1]) < 1e-4 * max(1, want_s[1])
        got = dz.cpu() if compact else dz.cpu()[rows]
        assert rel_l2(got, zr.grad[rows]) < TOL
        if not compact:
            untouched = torch.ones(n, dtype=torch.bool)
            untouched[rows] = False
            assert torch.all(dz.cpu()[untouched] == 3.0)


@pytest.mark.parametrize('d', [64, 128, 10, 4])
def test_edge_dot_and_distmult(d):
    from gnndelete_amd import ops
    g = torch.Generator().manual_seed(d)
    n, m, r = 70, 333, 5
    z = torch.randn(n, d, generator=g)
    e = torch.randint(0, n, (2, m), generator=g)
    rel = torch.randn(r, d, generator=g)
    et = torch.randint(0, r, (m,), generator=g)
    zg = z.cuda().requires_grad_(True)
    out = ops.edge_dot(zg, e[0].cuda(), e[1].cuda())
    assert rel_l2(out.detach().cpu(), (z.double()[e[0]] * z.double()[e[1]]).sum(-1)) < TOL
    out.sum().backward()
    zd = z.double().requires_grad_(True)
    (zd[e[0]] * zd[e[1]]).sum().backward()
    assert rel_l2(zg.grad.cpu(), zd.grad) < TOL
    out2 = ops.edge_dot(z.cuda(), e[0].cuda(), e[1].cuda(), rel.cuda(), et.cuda())
    assert rel_l2(out2.cpu(), (z.double()[e[0]] * rel.double()[et] * z.double()[e[1]]).sum(-1)) < TOL
    assert ops.edge_dot(z.cuda(), e[0, :0].cuda(), e[1, :0].cuda()).shape == (0,)


def test_adam_matches_torch_optim():
    from gnndelete_amd import _lib
    g = torch.Generator().manual_seed(0)
    p0 = torch.randn(64, 64, generator=g)
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1e-2)
    p = p0.clone().cuda()
    m = torch.zeros_like(p)
    v = torch.zeros_like(p)
    step = torch.zeros(1, dtype=torch.int32, device='cuda')
    for it in range(12):
        gr = torch.randn(64, 64, generator=g) * (10.0 ** (-it % 5))
        ref.grad = gr.clone()
        opt.step()
        gg = gr.cuda()
        _lib.check(_lib.lib().gd_adam_f32(p.data_ptr(), gg.data_ptr(), m.data_ptr(), v.data_ptr(), step.data_ptr(),
                                          p.numel(), 1e-2, 0.9, 0.999, 1e-8, torch.cuda.current_stream().cuda_stream))
    assert int(step) == 12
    assert rel_l2(p.cpu(), ref.detach()) < 1e-6


def test_adam_rounding_sequence_and_fused_entry_points():
    """One step from a non-trivial state: the kernel spells out torch's single-tensor rounding
    sequence (fp64-derived scalars, lerp as fma, addcmul, IEEE sqrt/div), so it lands within a couple
    of ulp of torch.optim.Adam element by element (torch's vectorised CPU sqrt is itself not
    correctly rounded everywhere), and the stand-alone, counter-driven and reduction-fused
    entry points are bit-identical to each other."""
    from gnndelete_amd import _lib
    from gnndelete_amd._lib import ptr, check
    L = _lib.lib()
    s = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(7)
    d, S, t0 = 128, 3000, 4
    p0 = torch.randn(d, d, generator=g) * 1e-3
    m0 = torch.randn(d, d, generator=g) * 1e-4
    v0 = torch.rand(d, d, generator=g) * 1e-8
    a = torch.randn(S, d, generator=g).cuda()
    up = (torch.randn(S, d, generator=g) * 1e-4).cuda()
    ws = torch.empty(L.gd_rows_gemm_wgrad_workspace(S, d, d), device='cuda')
    dw = torch.empty(d, d, device='cuda')
    check(L.gd_rows_gemm_wgrad_f32(ptr(a), d, None, ptr(up), d, None, None, None, S, d, d, ptr(dw), 0, ptr(ws), s))
    hyper = (1e-3, 0.9, 0.999, 1e-8)

    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=hyper[0], betas=hyper[1:3], eps=hyper[3])
    opt.state[ref] = {'step': torch.tensor(float(t0)), 'exp_avg': m0.clone(), 'exp_avg_sq': v0.clone()}
    ref.grad = dw.cpu()
    opt.step()

    outs = []
    for kind in ('step', 'at', 'fused'):
        p, m, v = p0.clone().cuda(), m0.clone().cuda(), v0.clone().cuda()
        ctr = torch.tensor([t0], dtype=torch.int32, device='cuda')
        if kind == 'step':
            check(L.gd_adam_f32(ptr(p), ptr(dw), ptr(m), ptr(v), ptr(ctr), d * d, *hyper, s))
            assert int(ctr) == t0 + 1
        elif kind == 'at':
            check(L.gd_adam_at_f32(ptr(p), ptr(dw), ptr(m), ptr(v), ptr(ctr), d * d, *hyper, s))
            assert int(ctr) == t0
        else:
            dw2 = torch.empty_like(dw)
            check(L.gd_rows_gemm_wgrad_adam_f32(ptr(a), d, None, ptr(up), d, None, None, None, S, d, d, ptr(dw2), 0, ptr(ws),
                                                ptr(p), ptr(m), ptr(v), ptr(ctr), *hyper, s))
            assert torch.equal(dw2, dw)
        outs.append((p.cpu(), m.cpu(), v.cpu()))
    for other in outs[1:]:
        for x, y in zip(outs[0], other):
            assert torch.equal(x, y)
    st = opt.state[ref]
    assert torch.equal(outs[0][1], st['exp_avg'])
    assert torch.equal(outs[0][2], st['exp_avg_sq'])
    # p' = p + q: the two can cancel, so the bound is in ulps of the operands, not of the result
    ulp = float(np.spacing(np.float32(max(float(p0.abs().max()), hyper[0]))))
    assert float((outs[0][0] - ref.detach()).abs().max()) <= 2 * ulp


@pytest.mark.parametrize('d', [128, 64, 16])
def test_gat_aggregate_forward_backward_vs_oracle(d):
    from gnndelete_amd import ops
    from gnndelete_amd.graph import build_csr
    from oracle import pyg_semantics as pyg
    n = 60
    ei = random_graph(n, 400, seed=d, isolate=2)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(n, 20, generator=g, dtype=torch.float64)
    w = torch.randn(d, 20, generator=g, dtype=torch.float64) * 0.3
    a_s = torch.randn(1, 1, d, generator=g, dtype=torch.float64)
    a_d = torch.randn(1, 1, d, generator=g, dtype=torch.float64)
    b = torch.randn(d, generator=g, dtype=torch.float64)
    up = torch.randn(n, d, generator=g, dtype=torch.float64)
    xr = x.clone().requires_grad_(True)
    asr, adr = a_s.clone().requires_grad_(True), a_d.clone().requires_grad_(True)
    want = pyg.gat_conv(xr, ei, w, asr, adr, b)
    want.backward(up)

    gr = build_csr(ei.cuda(), n, 'gat')
    xg = x.float().cuda().requires_grad_(True)
    asg, adg = a_s.float().cuda().requires_grad_(True), a_d.float().cuda().requires_grad_(True)
    h = xg @ w.float().cuda().t()
    out = ops.gat_aggregate(h, (h * asg.view(1, -1)).sum(-1), (h * adg.view(1, -1)).sum(-1), gr, b.float().cuda())
    assert rel_l2(out.detach().cpu(), want.detach()) < TOL
    out.backward(up.float().cuda())
    assert rel_l2(xg.grad.cpu(), xr.grad) < 5e-5
    assert rel_l2(asg.grad.cpu(), asr.grad) < 5e-5
    assert rel_l2(adg.grad.cpu(), adr.grad) < 5e-5


@pytest.mark.parametrize('blocks', [None, 4])
def test_rgcn_conv_vs_oracle(blocks):
    from gnndelete_amd.nn import RGCNConv
    from oracle import pyg_semantics as pyg
    n, r, i, o = 45, 6, 16, 32
    g = torch.Generator().manual_seed(5)
    ei = torch.randint(0, n, (2, 300), generator=g)
    et = torch.randint(0, r, (300,), generator=g)
    x = torch.randn(n, i, generator=g)
    conv = RGCNConv(i, o, r, blocks)
    with torch.no_grad():
        conv.bias.copy_(torch.randn(o, generator=g))
    xr = x.double().requires_grad_(True)
    want = pyg.rgcn_conv(xr, ei, et, conv.weight.detach().double(), conv.root.detach().double(),
                         conv.bias.detach().double(), blocks)
    up = torch.randn(n, o, generator=g)
    want.backward(up.double())
    conv = conv.cuda()
    xg = x.cuda().requires_grad_(True)
    out = conv(xg, ei.cuda(), et.cuda())
    assert rel_l2(out.detach().cpu(), want.detach()) < TOL
    out.backward(up.cuda())
    assert rel_l2(xg.grad.cpu(), xr.grad) < TOL


def test_cpu_tensors_are_rejected_loudly():
    from gnndelete_amd import _lib, ops
    from gnndelete_amd.graph import build_csr
    with pytest.raises(_lib.GnnDeleteHipError):
        build_csr(torch.zeros(2, 3, dtype=torch.long), 4, 'gcn')
    g = build_csr(torch.zeros(2, 3, dtype=torch.long).cuda(), 4, 'gcn')
    with pytest.raises(_lib.GnnDeleteHipError):
        ops.spmm(torch.zeros(4, 8), g)


@pytest.mark.parametrize('d', [128, 64, 8, 260])
def test_balanced_spmm_splits_hub_rows_and_matches_plain_kernel(d):
    """A star + random graph: the hub has thousands of in-edges, so its row is cut into many work
    items whose partials are combined by the fix-up kernel; result must equal the dense product."""
    from gnndelete_amd import ops
    from gnndelete_amd.graph import build_csr
    n = 3000
    g = torch.Generator().manual_seed(d)
    ei = torch.cat([random_graph(n, 9000, seed=d), torch.stack([torch.arange(1, n), torch.zeros(n - 1, dtype=torch.long)]),
                    torch.stack([torch.randint(0, n, (700,), generator=g), torch.full((700,), 5)])], 1)
    x = torch.randn(n, d, generator=g)
    b = torch.randn(d, generator=g)
    gr = build_csr(ei.cuda(), n, 'gcn')
    assert gr.plan.n_split >= 2 and gr.plan.n_items > n and gr.plan.n_slots >= 50
    a = dense_adj(ei[:, ei[0] != ei[1]], n) + torch.eye(n, dtype=torch.float64)
    deg = a.sum(1)
    want = (a / deg.sqrt()[:, None] / deg.sqrt()[None, :]) @ x.double() + b.double()
    y = ops._spmm_raw(gr.rowptr, gr.col, gr.val, x.cuda(), b.cuda(), 0.0, n, gr.plan)
    assert rel_l2(y.cpu(), want) < TOL
    y_plain = ops._spmm_raw(gr.rowptr, gr.col, gr.val, x.cuda(), b.cuda(), 0.0, n, None)
    assert rel_l2(y.cpu(), y_plain.cpu()) < 1e-6
    assert torch.equal(y, ops._spmm_raw(gr.rowptr, gr.col, gr.val, x.cuda(), b.cuda(), 0.0, n, gr.plan))
    yt = ops._spmm_raw(gr.rowptr_t, gr.col_t, gr.val_t, x.cuda(), None, 0.5, n, gr.plan_t)
    want_t = (a / deg.sqrt()[:, None] / deg.sqrt()[None, :]).t() @ x.double() + 0.5 * x.double()
    assert rel_l2(yt.cpu(), want_t) < TOL


@pytest.mark.parametrize('d,self_coef', [(128, 0.0), (64, 1.0), (32, 0.5), (8, 0.0), (260, 0.0)])
def test_onepass_spmm_sums_hub_rows_inside_the_launch(d, self_coef, monkeypatch):
    """gd_spmm_csr_onepass_f32: hub rows (65 ... 5,000 in-edges) summed by whole blocks in the same launch that sweeps
    the light rows, with XCD range tables (>= 8 k items), on the full plan and on a row-subset plan; against the fp64
    dense product and against the two-launch form (pieces + fix-up), which associates a hub row's sum differently."""
    from gnndelete_amd import ops
    from gnndelete_amd.graph import SplitPlan, build_csr
    n = 12000
    g = torch.Generator().manual_seed(d)
    hubs = [(0, 5000), (7, 65), (11, 128), (12, 129), (4000, 257), (11999, 1000), (6000, 64)]
    star = [torch.stack([torch.randint(0, n, (k,), generator=g), torch.full((k,), h)]) for h, k in hubs]
    ei = torch.cat([random_graph(n, 40000, seed=d)] + star, 1)
    x = torch.randn(n, d, generator=g)
    b = torch.randn(d, generator=g)
    gr = build_csr(ei.cuda(), n, 'sum')
    items, n_items, bounds = gr.plan.onepass(d)
    assert gr.plan.n_split >= 6 and n_items % 4 == 0 and bool((bounds % 4 == 0).all()) and int((items[:, 3] == -2).sum()) == 4 * gr.plan.n_split
    a = dense_adj(ei, n)
    want = a @ x.double() + self_coef * x.double() + b.double()
    xc, bc = x.cuda(), b.cuda()
    y = ops._spmm_raw(gr.rowptr, gr.col, None, xc, bc, self_coef, n, gr.plan)
    assert rel_l2(y.cpu(), want) < TOL
    assert torch.equal(y, ops._spmm_raw(gr.rowptr, gr.col, None, xc, bc, self_coef, n, gr.plan))     # bit-reproducible
    monkeypatch.setenv('GD_SPMM_TWO_LAUNCH', '1')
    y2 = ops._spmm_raw(gr.rowptr, gr.col, None, xc, bc, self_coef, n, gr.plan)
    monkeypatch.delenv('GD_SPMM_TWO_LAUNCH')
    assert rel_l2(y.cpu(), y2.cpu()) < 1e-6
    light = (gr.rowptr[1:] - gr.rowptr[:-1]) <= 64
    assert int((~light).sum()) == gr.plan.n_split
    # (light rows travel in multi-row items at d <= 64: their edges are dealt to the lane groups by position in the item,
    #  so the two forms agree to fp32 rounding there too, not bit for bit)
    assert rel_l2(y[light].cpu(), y2[light].cpu()) < 1e-6
    monkeypatch.setenv('GD_SPMM_MULTIROW', '0')
    plan1 = SplitPlan(gr.rowptr)
    y3 = ops._spmm_raw(gr.rowptr, gr.col, None, xc, bc, self_coef, n, plan1)
    monkeypatch.delenv('GD_SPMM_MULTIROW')
    assert torch.equal(y3[light], y2[light])                        # one light row per item: the same sum, bit for bit
    assert d > 64 or int((items[:, 3] <= -16).sum()) > 1000
    # a row subset (the rows a request can influence): only those rows are written
    rows = torch.cat([torch.tensor([0, 7, 12, 11999]), torch.randperm(n, generator=g)[:9000]]).unique().cuda()
    plan = SplitPlan(gr.rowptr, rows=rows)
    out = torch.full((n, d), 7.0, device='cuda')
    ops._spmm_raw(gr.rowptr, gr.col, None, xc, bc, self_coef, n, plan, out=out)
    # (a subset pairs different light rows into items at d <= 64, so those sums are associated differently: equal to rounding;
    #  hub rows and one-row items bit for bit)
    assert rel_l2(out[rows].cpu(), y[rows].cpu()) < 1e-6
    hub_rows = rows[~light[rows]]
    assert torch.equal(out[hub_rows], y[hub_rows]) and (d <= 64 or torch.equal(out[rows], y[rows]))
    rest = torch.ones(n, dtype=torch.bool, device='cuda')
    rest[rows] = False
    assert bool((out[rest] == 7.0).all())
    # transposed graph with values
    val_t = torch.rand(gr.col_t.shape[0], generator=g).cuda()
    yt = ops._spmm_raw(gr.rowptr_t, gr.col_t, val_t, xc, None, 0.0, n, gr.plan_t)
    at = torch.zeros(n, n, dtype=torch.float64)
    rt = torch.repeat_interleave(torch.arange(n), (gr.rowptr_t[1:] - gr.rowptr_t[:-1]).long().cpu())
    at.index_put_((rt, gr.col_t.long().cpu()), val_t.double().cpu(), accumulate=True)
    assert rel_l2(yt.cpu(), at @ x.double()) < TOL


@pytest.mark.parametrize('d', [128, 64, 16, 260])
def test_rowtarget_mse_matches_direct_formula(d):
    from gnndelete_amd import _lib
    g = torch.Generator().manual_seed(d + 1)
    n, u = 500, 333
    z = torch.randn(n, d, generator=g)
    tm = torch.randn(u, d, generator=g)
    rows = torch.randperm(n, generator=g)[:u].sort().values
    cnt = torch.randint(1, 5, (u,), generator=g).float()
    coef = torch.rand(u, generator=g)
    kind = torch.randint(0, 2, (u,), generator=g).int()
    diff = z.double()[rows] - tm.double()
    sq = (diff ** 2).sum(1) * cnt.double()
    want = [float(sq[kind == 0].sum()), float(sq[kind == 1].sum())]
    dz = torch.full((n, d), 5.0, device='cuda')
    sums = torch.zeros(2, device='cuda')
    ws = torch.empty(_lib.lib().gd_rowtarget_mse_workspace(u), device='cuda')
    dev = [t_.cuda() for t_ in (z, tm, rows.int(), coef, cnt, kind)]
    _lib.check(_lib.lib().gd_rowtarget_mse_f32(dev[0].data_ptr(), d, dev[1].data_ptr(), d, dev[2].data_ptr(),
                                               dev[3].data_ptr(), dev[4].data_ptr(), dev[5].data_ptr(), u,
                                               dz.data_ptr(), d, sums.data_ptr(), ws.data_ptr(),
                                               torch.cuda.current_stream().cuda_stream))
    assert abs(float(sums[0]) - want[0]) < 1e-5 * want[0] and abs(float(sums[1]) - want[1]) < 1e-5 * want[1]
    assert rel_l2(dz.cpu()[rows], coef.double()[:, None] * diff) < TOL
    rest = torch.ones(n, dtype=torch.bool)
    rest[rows] = False
    assert torch.all(dz.cpu()[rest] == 5.0)


@pytest.mark.parametrize('n,s,d,n_df', [(400, 300, 64, 40), (200, 70, 64, 10), (90, 90, 128, 0), (50, 33, 32, 5),
                                        (40, 1, 64, 0), (40, 0, 64, 0), (60, 45, 20, 6), (3000, 2500, 64, 300)])
def test_pairs_sigmoid_mse_matches_oracle_pairs(n, s, d, n_df):
    """Edge-probability NI term (gnndelete.py:174-193, 239-241): the fused tile kernel against the
    oracle's explicit pair list in float64 - value and gradient - including a ragged last tile, a
    single node, no node at all, excluded (Df) pairs and a feature width off the MFMA path."""
    from gnndelete_amd import ops
    from oracle import gnndelete_ref as R
    g = torch.Generator().manual_seed(n + s + d)
    z = torch.randn(n, d, generator=g) * 0.4
    mask = torch.zeros(n, dtype=torch.bool)
    mask[torch.randperm(n, generator=g)[:s]] = True
    nodes = mask.nonzero().flatten()
    df = torch.stack([nodes[torch.randint(0, max(s, 1), (n_df,), generator=g)],
                      nodes[torch.randint(0, max(s, 1), (n_df,), generator=g)]]) if s else torch.zeros(2, 0, dtype=torch.long)
    ori = torch.randn(n, n, generator=g)
    pairs = R.sdf_pair_index(n, mask, df) if s else torch.zeros(2, 0, dtype=torch.long)
    count = pairs.shape[1]

    zd = z.double().requires_grad_(True)
    if count:
        want = ((zd[pairs[0]] * zd[pairs[1]]).sum(-1).sigmoid() - ori.double()[pairs[0], pairs[1]].sigmoid()).pow(2).mean()
        want.backward()

    pos = torch.full((n,), -1, dtype=torch.long)
    pos[nodes] = torch.arange(s)
    s_pad = (s + 3) // 4 * 4
    target = torch.full((max(s, 1), max(s_pad, 4)), -1.0)
    target[pos[pairs[0]], pos[pairs[1]]] = ori[pairs[0], pairs[1]].sigmoid()
    zg = z.cuda().requires_grad_(True)
    got = ops.pairs_sigmoid_mse(zg, nodes.int().cuda(), target.cuda(), count)
    (got * 3.0).backward()
    if count == 0:
        assert float(got) == 0.0 and float(zg.grad.abs().max()) == 0.0
        return
    assert abs(float(got) - float(want)) <= 1e-5 * abs(float(want))
    assert rel_l2(zg.grad.cpu() / 3.0, zd.grad) < TOL
    assert torch.all(zg.grad.cpu()[~mask] == 0)
    # deterministic (fixed-order split reduction)
    zg2 = z.cuda().requires_grad_(True)
    got2 = ops.pairs_sigmoid_mse(zg2, nodes.int().cuda(), target.cuda(), count)
    (got2 * 3.0).backward()
    assert torch.equal(got2, got) and torch.equal(zg2.grad, zg.grad)


@pytest.mark.parametrize('n,m,R,din,dout,nb', [(60, 500, 5, 128, 128, 4), (60, 500, 5, 128, 64, 4), (40, 300, 3, 24, 12, None),
                                               (50, 0, 4, 32, 32, 4), (300, 6000, 30, 128, 64, 4), (30, 200, 2, 104, 88, 4)])
def test_rgcn_conv_fused_matches_oracle(n, m, R, din, dout, nb):
    """Fused typed aggregate + (block-diagonal) transform vs the oracle's RGCNConv restatement in float64:
    forward and input gradient, incl. nodes with no in-edge, relations that never occur, an empty graph
    and widths off the 64-lane grid."""
    from gnndelete_amd import ops
    from gnndelete_amd.graph import TypedNodeCSR
    from oracle import pyg_semantics as pyg
    g = torch.Generator().manual_seed(n + m + R)
    ei = torch.randint(0, max(1, n - 3), (2, m), generator=g)
    et = torch.randint(0, max(1, R - 1), (m,), generator=g)
    x = torch.randn(n, din, generator=g, dtype=torch.float64)
    w = torch.randn((R, din, dout) if nb is None else (R, nb, din // nb, dout // nb), generator=g, dtype=torch.float64) * 0.2
    root = torch.randn(din, dout, generator=g, dtype=torch.float64) * 0.2
    bias = torch.randn(dout, generator=g, dtype=torch.float64)
    up = torch.randn(n, dout, generator=g, dtype=torch.float64)
    xr = x.clone().requires_grad_(True)
    want = pyg.rgcn_conv(xr, ei, et, w, root, bias, nb)
    want.backward(up)
    tg = TypedNodeCSR(ei.cuda(), et.cuda(), n, R)
    xg = x.float().cuda().requires_grad_(True)
    got = ops.rgcn_conv_frozen(xg, tg, w.float().cuda(), root.float().cuda(), bias.float().cuda(), 1 if nb is None else nb)
    got.backward(up.float().cuda())
    assert rel_l2(got.detach().cpu(), want.detach()) < TOL
    assert rel_l2(xg.grad.cpu(), xr.grad) < TOL


@pytest.mark.parametrize('n,m,R,din,dout,nb', [(200, 20000, 7, 128, 128, 4), (1000, 30000, 102, 128, 64, 4), (130, 9000, 5, 64, 64, 4),
                                               (257, 5000, 9, 128, 128, None), (64, 4000, 3, 64, 128, None), (500, 100, 4, 128, 64, None)])
def test_rgcn_tile_conv_matches_oracle_and_node_major(n, m, R, din, dout, nb, monkeypatch):
    """gd_rgcn_tile_conv_f32 ((64-node tile, relation) steps, accumulators in MFMA registers): forward and input
    gradient against the float64 oracle and against the node-major kernel; hub runs far beyond one 16-edge piece
    (several passes of one relation), relations that occur in no tile, a last tile with fewer than 64 nodes, dense
    and 4-block weights, every supported width pair; bit-reproducible."""
    from gnndelete_amd import _lib, ops
    from gnndelete_amd.graph import TypedNodeCSR
    from oracle import pyg_semantics as pyg
    monkeypatch.setenv('GD_RGCN_WAVE', '0')                # the 4-block cases would take the wave-private kernel (next test)
    g = torch.Generator().manual_seed(n + m + R)
    ei = torch.randint(0, n, (2, m), generator=g)
    et = torch.randint(0, max(1, R - 1), (m,), generator=g)
    ei[1, :m // 5] = 5                                     # a hub: one node with m / 5 in-edges ...
    et[:m // 10] = 2 % R                                   # ... half of them of one relation
    ei[0, m // 5:m // 4] = 9                               # and a hub source (long runs in the transposed graph)
    x = torch.randn(n, din, generator=g, dtype=torch.float64)
    w = torch.randn((R, din, dout) if nb is None else (R, nb, din // nb, dout // nb), generator=g, dtype=torch.float64) * 0.2
    root = torch.randn(din, dout, generator=g, dtype=torch.float64) * 0.2
    bias = torch.randn(dout, generator=g, dtype=torch.float64)
    up = torch.randn(n, dout, generator=g, dtype=torch.float64)
    xr = x.clone().requires_grad_(True)
    want = pyg.rgcn_conv(xr, ei, et, w, root, bias, nb)
    want.backward(up)
    nbk = 1 if nb is None else nb
    assert _lib.lib().gd_rgcn_tile_kl(din, dout, nbk, 0) > 0 and _lib.lib().gd_rgcn_tile_kl(dout, din, nbk, 1) > 0
    tg = TypedNodeCSR(ei.cuda(), et.cuda(), n, R)
    plan = tg.tile_plan(False)
    assert plan['n_pieces'] >= plan['n_steps'] > 0 and int(plan['piece'][:, 1].max()) >> 8 <= 16
    assert int((plan['step_piece_ptr'][1:] - plan['step_piece_ptr'][:-1]).max()) <= 32

    def run():
        xg = x.float().cuda().requires_grad_(True)
        got = ops.rgcn_conv_frozen(xg, tg, w.float().cuda(), root.float().cuda(), bias.float().cuda(), nbk)
        got.backward(up.float().cuda())
        return got.detach(), xg.grad
    got, dx = run()
    assert rel_l2(got.cpu(), want.detach()) < TOL
    assert rel_l2(dx.cpu(), xr.grad) < TOL
    got2, dx2 = run()
    assert torch.equal(got, got2) and torch.equal(dx, dx2)
    monkeypatch.setenv('GD_RGCN_NODE_MAJOR', '1')
    ref, dref = run()
    assert rel_l2(got.cpu(), ref.cpu()) < 1e-5 and rel_l2(dx.cpu(), dref.cpu()) < 1e-5


@pytest.mark.parametrize('n,m,R,din,dout', [(200, 20000, 7, 128, 128), (1000, 30000, 102, 128, 64), (130, 9000, 5, 64, 64),
                                            (300, 5000, 4, 64, 128), (64, 40, 3, 128, 128), (5000, 400000, 51, 128, 128)])
def test_rgcn_wave_conv_matches_oracle_and_tile_kernel(n, m, R, din, dout, monkeypatch):
    """gd_rgcn_wave_conv_f32 (one wave per (64-node tile, diagonal block), units of 16 slots x 4 edges, LDS accumulators):
    forward and input gradient against the float64 oracle and against the tile kernel; a hub whose runs span many slots of
    one unit and many units (the slots of one node are ADDED in the accumulator), relations that occur in no tile, a last
    tile with fewer than 64 nodes, tiles whose unit count is no multiple of the kernel's unroll (it runs on the plan's empty
    unit past their end), all four width pairs, both pipeline depths; bit-reproducible."""
    from gnndelete_amd import _lib, ops
    from gnndelete_amd.graph import TypedNodeCSR
    from oracle import pyg_semantics as pyg
    g = torch.Generator().manual_seed(n + m + R + din)
    ei = torch.randint(0, n, (2, m), generator=g)
    et = torch.randint(0, max(1, R - 1), (m,), generator=g)
    ei[1, :m // 5] = 5                                     # a hub: one node with m / 5 in-edges ...
    et[:m // 10] = 2 % R                                   # ... half of them of one relation
    ei[0, m // 5:m // 4] = 9                               # and a hub source (long runs in the transposed graph)
    nb = 4
    x = torch.randn(n, din, generator=g, dtype=torch.float64)
    w = torch.randn(R, nb, din // nb, dout // nb, generator=g, dtype=torch.float64) * 0.2
    root = torch.randn(din, dout, generator=g, dtype=torch.float64) * 0.2
    bias = torch.randn(dout, generator=g, dtype=torch.float64)
    up = torch.randn(n, dout, generator=g, dtype=torch.float64)
    xr = x.clone().requires_grad_(True)
    want = pyg.rgcn_conv(xr, ei, et, w, root, bias, nb)
    want.backward(up)
    assert _lib.lib().gd_rgcn_wave_covers(din, dout, nb) == 1 and _lib.lib().gd_rgcn_wave_covers(din, dout, 1) == 0
    tg = TypedNodeCSR(ei.cuda(), et.cuda(), n, R)
    for trans in (False, True):
        plan = tg.wave_plan(trans)
        assert plan['n_units'] == int(plan['tile_unit_ptr'][-1]) > 0 and plan['n_tiles'] == (n + 63) // 64
        assert plan['unit_row'].shape[0] == plan['unit_edges'].shape[0] == plan['unit_rel'].numel() == plan['n_units'] + 1
        assert int(plan['unit_row'][-1].abs().sum()) == 0 and bool((plan['unit_edges'][-1, :, :, 0] == n).all())   # the empty unit
        used = plan['unit_edges'][..., 0] != n
        assert int(used.sum()) == m                                          # every typed edge sits in exactly one pair
        assert int((plan['unit_row'] & 255).max()) < 64 and int((plan['unit_rel'] & 0xffff).max()) < R

    def run():
        xg = x.float().cuda().requires_grad_(True)
        got = ops.rgcn_conv_frozen(xg, tg, w.float().cuda(), root.float().cuda(), bias.float().cuda(), nb)
        got.backward(up.float().cuda())
        return got.detach(), xg.grad
    got, dx = run()
    assert rel_l2(got.cpu(), want.detach()) < TOL
    assert rel_l2(dx.cpu(), xr.grad) < TOL
    got2, dx2 = run()
    assert torch.equal(got, got2) and torch.equal(dx, dx2)
    monkeypatch.setenv('GD_RGCN_WAVE_DEPTH', '1')          # one unit of rows in flight instead of three: the same sums
    got1, dx1 = run()
    assert torch.equal(got, got1) and torch.equal(dx, dx1)
    monkeypatch.setenv('GD_RGCN_WAVE_BPW', '2')            # 16-wide blocks two per wave (opt-in form): the same sums
    got1, dx1 = run()
    assert torch.equal(got, got1) and torch.equal(dx, dx1)
    monkeypatch.delenv('GD_RGCN_WAVE_DEPTH')
    got1, dx1 = run()
    assert torch.equal(got, got1) and torch.equal(dx, dx1)
    # relu_in: the conv reads relu(x), formed where the gathered rows land - the same bits as the conv of a relu'd copy
    xc = x.float().cuda()
    wv = w.float().cuda()
    y_a = torch.zeros(n, dout, device='cuda')
    y_b = torch.zeros(n, dout, device='cuda')
    assert ops.rgcn_wave_relu_ok(tg, xc, y_a, nb)
    ops.rgcn_typed_accumulate(tg, xc, wv, nb, 0, y_a, relu_in=True)
    ops.rgcn_typed_accumulate(tg, xc.clamp(min=0), wv, nb, 0, y_b)
    assert torch.equal(y_a, y_b) and float(y_a.abs().max()) > 0
    monkeypatch.setenv('GD_RGCN_WAVE', '0')
    ref, dref = run()
    assert rel_l2(got.cpu(), ref.cpu()) < 1e-5 and rel_l2(dx.cpu(), dref.cpu()) < 1e-5


@pytest.mark.parametrize('n,d,frac,loss_frac', [(500, 64, 0.6, 0.8), (300, 32, 1.0, 1.0), (4000, 64, 0.9, 0.5), (70, 64, 0.3, 0.0),
                                                (300000, 64, 0.8, 0.7), (90000, 32, 0.5, 0.9)])
def test_del_loss_bwd_fused_matches_separate_steps(n, d, frac, loss_frac):
    """Fused last-layer kernel (Del forward + folded MSE terms + Del input gradient) vs the same three steps in
    float64: dz on the Del rows, dp scattered to the full matrix, the two loss sums; rows with no loss slot."""
    from gnndelete_amd import _lib
    from gnndelete_amd._lib import ptr, check
    L = _lib.lib()
    g = torch.Generator().manual_seed(n + d)
    p = torch.randn(n, d, generator=g)
    w = torch.randn(d, d, generator=g) * 0.2
    mask = torch.rand(n, generator=g) < frac
    idx = mask.nonzero().flatten()
    s = idx.numel()
    has = torch.rand(s, generator=g) < loss_frac
    slot = torch.full((s,), -1, dtype=torch.int32)
    n_slots = int(has.sum())
    slot[has] = torch.arange(n_slots, dtype=torch.int32)[torch.randperm(n_slots, generator=g)]
    tm = torch.randn(max(n_slots, 1), d, generator=g)
    coef = torch.rand(max(n_slots, 1), generator=g) + 0.1
    cnt = (torch.randint(1, 4, (max(n_slots, 1),), generator=g).float()) * torch.where(torch.rand(max(n_slots, 1), generator=g) < 0.5, -1.0, 1.0)
    z = p.double()[idx] @ w.double()
    dz_want = torch.zeros(s, d, dtype=torch.float64)
    sl = slot[has].long()
    df = z[has] - tm.double()[sl]
    dz_want[has] = coef.double()[sl][:, None] * df
    dp_want = torch.zeros(n, d, dtype=torch.float64)
    dp_want[idx] = dz_want @ w.double().t()
    sq = (df * df).sum(1) * cnt.double()[sl].abs()
    want_s = [float(sq[cnt[sl] >= 0].sum()), float(sq[cnt[sl] < 0].sum())]

    nb = L.gd_del_loss_bwd_blocks(s)
    dz = torch.full((max(s, 1), d), 7.0, device='cuda')
    dp = torch.zeros(n, d, device='cuda')
    parts = torch.zeros(2 * max(nb, 1), device='cuda')
    dev = lambda t_: t_.cuda()
    args = [dev(p), dev(idx.int()), dev(w), dev(slot), dev(tm), dev(coef), dev(cnt)]
    check(L.gd_del_loss_bwd_f32(ptr(args[0]), d, ptr(args[1]), s, ptr(args[2]), d, ptr(args[3]), ptr(args[4]), ptr(args[5]),
                                ptr(args[6]), ptr(dz), d, ptr(dp), d, ptr(parts), torch.cuda.current_stream().cuda_stream))
    assert rel_l2(dz.cpu()[:s], dz_want) < TOL or float(dz_want.abs().max()) == 0
    assert rel_l2(dp.cpu(), dp_want) < TOL or float(dp_want.abs().max()) == 0
    got_s = parts.view(-1, 2).double().sum(0).cpu()
    np.testing.assert_allclose(got_s.numpy(), want_s, rtol=1e-5, atol=1e-7)
    # the same pass + the Del weight's gradient (gd_del_loss_bwd_wgrad_f32): per-block partial sums of p[idx]^T dz in the layout
    # of gd_rows_gemm_wgrad_f32, finished by gd_rows_gemm_wgrad_reduce_f32; with and without the dz buffer; same dp and sums
    nbw = L.gd_rows_gemm_wgrad_blocks(s)
    assert L.gd_rows_gemm_wgrad_workspace(s, d, d) == max(nbw, 0) * d * d
    dw_want = p.double()[idx].t() @ dz_want
    for with_dz in (True, False):
        dz2 = torch.full((max(s, 1), d), 7.0, device='cuda')
        dp2 = torch.zeros(n, d, device='cuda')
        parts2 = torch.zeros(2 * max(nbw, 1), device='cuda')
        ws = torch.full((max(nbw, 1) * d * d,), 3.0, device='cuda')
        st = torch.cuda.current_stream().cuda_stream
        check(L.gd_del_loss_bwd_wgrad_f32(ptr(args[0]), d, ptr(args[1]), s, ptr(args[2]), d, ptr(args[3]), ptr(args[4]), ptr(args[5]),
                                          ptr(args[6]), ptr(dz2) if with_dz else None, d, ptr(dp2), d, ptr(parts2), ptr(ws), st))
        ws_form = d == 64 and s >= 65536 and os.environ.get('GD_DEL2_WS') != '0'       # the weight-stationary kernel (16-row units)
        if ws_form:                                                 # another k order of the same products: fp32 rounding apart
            assert rel_l2(dp2, dp) < 1e-6 and rel_l2(dp2.cpu(), dp_want) < TOL
            assert not with_dz or rel_l2(dz2[:s], dz[:s]) < 1e-6
        else:
            assert torch.equal(dp2, dp)                             # the same products in the same order
            if with_dz:
                assert torch.equal(dz2, dz)
        np.testing.assert_allclose(parts2.view(-1, 2).double().sum(0).cpu().numpy(), want_s, rtol=1e-5, atol=1e-7)
        dw = ws.view(max(nbw, 1), d, d).double().sum(0).cpu()
        assert rel_l2(dw, dw_want) < TOL or float(dw_want.abs().max()) == 0
        # against the separate weight-gradient launch on the same operands (another summation order: fp32 rounding apart)
        dw_sep = torch.zeros(d, d, device='cuda')
        ws_sep = torch.zeros(max(nbw, 1) * d * d, device='cuda')
        check(L.gd_rows_gemm_wgrad_f32(ptr(args[0]), d, ptr(args[1]), ptr(dz), d, None, None, None, s, d, d, ptr(dw_sep), 0, ptr(ws_sep), st))
        assert rel_l2(dw.float(), dw_sep.cpu()) < 1e-5 or float(dw_want.abs().max()) == 0


def _csr_reference(src, dst, n):
    """numpy restatement: rows = targets, sources ascending, ties in input order (stable)."""
    key = dst.astype(np.int64) * n + src.astype(np.int64)
    order = np.argsort(key, kind='stable')
    rowptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(rowptr, dst + 1, 1)
    return np.cumsum(rowptr), src[order], order


@pytest.mark.parametrize('n,m', [(1, 0), (1, 5), (7, 0), (50, 400), (1000, 20000), (235868, 2000000)])
def test_csr_from_coo_is_bit_exact(n, m):
    """gd_csr_from_coo (P1 layout step) against a stable numpy sort: multi-edges, self loops, isolated
    rows, both arrival orders give the same CSR."""
    from gnndelete_amd.graph import csr_from_coo
    rng = np.random.default_rng(n + m)
    src = rng.integers(0, n, m)
    dst = rng.integers(0, max(1, n - n // 5), m)          # the last fifth of the rows stays empty
    if m >= 100:
        src[:50], dst[:50] = src[50:100], dst[50:100]      # duplicates
    rp, col, order = csr_from_coo(torch.from_numpy(src).cuda(), torch.from_numpy(dst).cuda(), n)
    want_rp, want_col, want_order = _csr_reference(src, dst, n)
    assert rp.dtype == col.dtype == order.dtype == torch.int32
    assert np.array_equal(rp.cpu().numpy(), want_rp)
    assert np.array_equal(col.cpu().numpy(), want_col)
    assert np.array_equal(order.cpu().numpy(), want_order)
    if m:
        perm = rng.permutation(m)                          # arrival order must not matter for (rowptr, col)
        rp2, col2, _ = csr_from_coo(torch.from_numpy(src[perm]).cuda(), torch.from_numpy(dst[perm]).cuda(), n)
        assert torch.equal(rp2, rp) and torch.equal(col2, col)


def test_csr_from_coo_rejects_out_of_range_endpoints():
    from gnndelete_amd.graph import csr_from_coo
    src = torch.tensor([0, 1, 9], device='cuda')
    dst = torch.tensor([1, 2, 0], device='cuda')
    with pytest.raises(IndexError):
        csr_from_coo(src, dst, 5)
    with pytest.raises(IndexError):
        csr_from_coo(torch.tensor([0, -1], device='cuda'), torch.tensor([1, 2], device='cuda'), 5)


@pytest.mark.parametrize('n,d_in,d_out,select', [(500, 128, 64, False), (500, 128, 64, True), (77, 64, 32, True),
                                                 (300, 128, 128, False), (1, 32, 64, False)])
def test_rows_gemm_dots_matches_fp64(n, d_in, d_out, select, matrix_split):
    """gd_rows_gemm_dots_f32: h = act(x or x_alt) W^T and the two row dots <h, u1>, <h, u2> from its epilogue
    (GATConv lin_src + attention logits, framework/models/gat.py:11-12)."""
    from gnndelete_amd import ops
    g = torch.Generator().manual_seed(n + d_out)
    x = torch.randn(n, d_in, generator=g)
    xa = torch.randn(n, d_in, generator=g)
    sel = (torch.rand(n, generator=g) < 0.4).to(torch.uint8)
    w = torch.randn(d_out, d_in, generator=g) / d_in ** 0.5
    u1, u2 = torch.randn(d_out, generator=g), torch.randn(1, 1, d_out, generator=g)
    src = torch.where(sel.bool()[:, None], xa, x) if select else x
    if select:
        src = src.clamp(min=0)
    want = src.double() @ w.double().t()
    h, a1, a2 = ops.rows_gemm_dots(x.cuda(), w.cuda(), u1.cuda(), u2.cuda(), inp_alt=xa.cuda() if select else None,
                                   sel=sel.cuda() if select else None, relu_in=select)
    assert rel_l2(h.cpu(), want) < TOL
    assert rel_l2(a1.cpu(), want @ u1.double()) < TOL
    assert rel_l2(a2.cpu(), want @ u2.double().reshape(-1)) < TOL


@pytest.mark.parametrize('pieces', ['0', '1'])
@pytest.mark.parametrize('d,hub_degree', [(64, 1500), (16, 700), (128, 2500), (64, 5000)])
def test_gat_hub_row_with_more_pieces_than_feature_lanes(d, hub_degree, pieces, monkeypatch):
    """pieces = 0 (default): the one-launch form - the hub row is a GROUP of four member items, each walking its share of
    the in-edges 64 at a time with an online softmax, merged in LDS by the block (forward), its score gradients finished by
    the members once the row's t is summed in LDS (backward).  pieces = 1 (GD_GAT_PIECES=1): the piece form with fix-up launches:
    A hub row is cut into pieces of 64 in-edges that the fix-up merges flash-attention style.  With more pieces
    than d/4 (the lanes that carry features in the fix-up) the merge used to drop the tail pieces' rescale factors
    (LDS-crossbar shuffle from masked-off lanes reads 0); > 64 pieces take the serial branch.  Forward and backward
    of the balanced GAT kernels on a graph with one such hub vs the fp64 oracle."""
    from gnndelete_amd import ops
    from gnndelete_amd.graph import build_csr
    from oracle import pyg_semantics as pyg
    monkeypatch.setenv('GD_GAT_PIECES', pieces)
    n = hub_degree + 200
    g = torch.Generator().manual_seed(d + hub_degree)
    hub = 17
    src = torch.randperm(n, generator=g)[:hub_degree]
    ei = torch.cat([torch.stack([src, torch.full_like(src, hub)]), torch.randint(0, n, (2, 3 * n), generator=g)], 1)
    ei = ei[:, ei[0] != ei[1]]
    ei = torch.unique(ei[0] * n + ei[1])
    ei = torch.stack([ei // n, ei % n])
    x = torch.randn(n, 12, generator=g, dtype=torch.float64)
    w = torch.randn(d, 12, generator=g, dtype=torch.float64) * 0.3
    a_s, a_d = (torch.randn(1, 1, d, generator=g, dtype=torch.float64) for _ in range(2))
    b = torch.randn(d, generator=g, dtype=torch.float64)
    up = torch.randn(n, d, generator=g, dtype=torch.float64)
    xr = x.clone().requires_grad_(True)
    want = pyg.gat_conv(xr, ei, w, a_s, a_d, b)
    want.backward(up)
    gr = build_csr(ei.cuda(), n, 'gat')
    assert int(gr.plan.split[:, 2].max()) > d // 4, 'the hub must be cut into more pieces than feature lanes'
    xg = x.float().cuda().requires_grad_(True)
    h = xg @ w.float().cuda().t()
    out = ops.gat_aggregate(h, (h * a_s.float().cuda().view(1, -1)).sum(-1), (h * a_d.float().cuda().view(1, -1)).sum(-1),
                            gr, b.float().cuda())
    err = (out.detach().cpu().double() - want.detach()).norm(dim=1) / want.detach().norm(dim=1)
    assert float(err[hub]) < 1e-5, float(err[hub])
    assert rel_l2(out.detach().cpu(), want.detach()) < TOL
    out.backward(up.float().cuda())
    assert rel_l2(xg.grad.cpu(), xr.grad) < 5e-5


@pytest.mark.parametrize('d', [64, 128])
def test_gat_backward_without_the_transposition_pass(d, monkeypatch):
    """gd_spmm_csr_onepass_aux_f32: the source-major aggregation of GATConv's backward reads (attention weight, score
    gradient) through the transpose permutation itself and sums the score gradients per source row - against the form with
    the separate transposition pass (gd_gat_transpose_edges_f32 + gd_spmm_csr_onepass_f32): the message gradient bit for
    bit (same weights, same summation order), d a_src to fp32 rounding (64-lane against 8-lane sums), on a graph with hub
    rows in BOTH directions (group items), two-row items (d = 64), >= 8 k items, and on a row subset."""
    from gnndelete_amd import ops
    from gnndelete_amd.graph import SplitPlan, build_csr
    n = 30000
    g = torch.Generator().manual_seed(d)
    hubs = [(5, 3000), (77, 300), (29990, 70)]
    star_in = [torch.stack([torch.randint(0, n, (k,), generator=g), torch.full((k,), h)]) for h, k in hubs]
    star_out = [torch.stack([torch.full((k,), h + 1), torch.randint(0, n, (k,), generator=g)]) for h, k in hubs]
    ei = torch.cat([random_graph(n, 200000, seed=d)] + star_in + star_out, 1)
    gr = build_csr(ei.cuda(), n, 'gat')
    h = torch.randn(n, d, generator=g).cuda()
    a_src, a_dst = torch.randn(n, generator=g).cuda(), torch.randn(n, generator=g).cuda()
    dy = torch.randn(n, d, generator=g).cuda()
    _, rowmax, rowsum = ops.gat_forward_raw(gr, h, a_src, a_dst, None, 0.2)
    got = [t.clone() for t in ops.gat_backward_raw(gr, h, a_src, a_dst, rowmax, rowsum, dy, 0.2)]
    again = ops.gat_backward_raw(gr, h, a_src, a_dst, rowmax, rowsum, dy, 0.2)
    assert all(torch.equal(a, b) for a, b in zip(got, again))                      # bit-reproducible
    monkeypatch.setenv('GD_GAT_TRANSPOSE_PASS', '1')
    ref = [t.clone() for t in ops.gat_backward_raw(gr, h, a_src, a_dst, rowmax, rowsum, dy, 0.2)]
    monkeypatch.delenv('GD_GAT_TRANSPOSE_PASS')
    assert torch.equal(got[0], ref[0]) and torch.equal(got[2], ref[2])
    assert rel_l2(got[1].cpu(), ref[1].cpu()) < 1e-6 and float((got[1] - ref[1]).abs().max()) < 1e-4 * float(ref[1].abs().max())
    # source rows of a subset only (the rows a request can influence): the others keep a zero score gradient
    rows = torch.cat([torch.tensor([6, 78, 0]), torch.randperm(n, generator=g)[:12000]]).unique().cuda()
    plan_t = SplitPlan(gr.rowptr_t, rows=rows)
    sub = ops.gat_backward_raw(gr, h, a_src, a_dst, rowmax, rowsum, dy, 0.2, plan_t=plan_t)
    # (a subset pairs different light rows into two-row items at d = 64: another association of the same sums)
    assert rel_l2(sub[0][rows].cpu(), got[0][rows].cpu()) < 1e-6 and rel_l2(sub[1][rows].cpu(), got[1][rows].cpu()) < 1e-6
    other = torch.ones(n, dtype=torch.bool, device='cuda')
    other[rows] = False
    assert float(sub[1][other].abs().max()) == 0.0


@pytest.mark.parametrize('d', [64, 128, 16, 10])
def test_edge_dot_backward_kernel_is_exact_and_reproducible(d):
    """Decoder input gradients (dot product and DistMult) from gd_edge_dot_bwd_f32 - one deterministic pass over a
    node-major incidence list instead of autograd's two atomic scatter-adds - against fp64 autograd, with random
    upstream gradients, repeated edges, self pairs, a hub node (the cooperative path) and isolated nodes; two runs give
    the same bits.  d = 10 takes the scatter fallback (rows not 16-byte aligned)."""
    from gnndelete_amd import ops
    g = torch.Generator().manual_seed(d + 1)
    n, m, r = 90, 700, 5
    z = torch.randn(n, d, generator=g)
    e = torch.randint(0, n - 7, (2, m), generator=g)              # the last 7 nodes stay isolated
    e[:, :20] = e[:, 20:40]                                        # repeated edges
    e[1, 40:50] = e[0, 40:50]                                      # self pairs
    e[0, 100:420] = 3                                              # a hub: 320+ incidences, summed by the whole wave
    rel = torch.randn(r, d, generator=g)
    et = torch.randint(0, r, (m,), generator=g)
    up = torch.randn(m, generator=g)
    for use_rel in (False, True):
        zd, rd = z.double().requires_grad_(True), rel.double().requires_grad_(True)
        s = (zd[e[0]] * (rd[et] if use_rel else 1.0) * zd[e[1]]).sum(-1)
        s.backward(up.double())
        grads = []
        for _ in range(2):
            zg, rg = z.cuda().requires_grad_(True), rel.cuda().requires_grad_(True)
            out = ops.edge_dot(zg, e[0].cuda(), e[1].cuda(), rg if use_rel else None, et.cuda() if use_rel else None)
            out.backward(up.cuda())
            grads.append(zg.grad.clone())
            assert rel_l2(zg.grad.cpu(), zd.grad) < TOL
            if use_rel:
                assert rel_l2(rg.grad.cpu(), rd.grad) < TOL
        if d % 4 == 0:
            assert torch.equal(grads[0], grads[1])
        assert float(grads[0][n - 7:].abs().max()) == 0.0


def test_rbf_cka_on_the_device_matches_reference_golden():
    """RBFCKA with an explicit sigma (the form of gnndelete_nodeemb.py:38-66 that upstream's own code can run): value and
    gradient of the device path - Gram matrices on the HIP dense kernels (ops.gram), centering without H K H products -
    against the reference's function."""
    from helpers import load_golden, t
    from gnndelete_amd.framework.trainer.gnndelete_nodeemb import get_loss_fct
    fx = load_golden('losses.npz')
    a = t(fx['a']).cuda().requires_grad_(True)
    v = get_loss_fct('rbf_cka')(a, t(fx['b']).cuda(), sigma=2.0)
    v.backward()
    assert rel_l2(v.detach().cpu(), fx['rbf_cka_sigma2::value']) < 1e-5
    assert rel_l2(a.grad.cpu(), fx['rbf_cka_sigma2::grad']) < 1e-4


@pytest.mark.parametrize('n,d', [(29, 16), (300, 64), (515, 128), (130, 20)])
def test_gram_matches_fp64_with_autograd(n, d):
    """ops.gram: x x^T (and x y^T) 128 columns at a time on the kernels of ops.dense - the Gram matrices of the CKA losses and
    Trainer.test's all-pairs logits - value and both gradients against fp64."""
    from gnndelete_amd import ops
    g = torch.Generator().manual_seed(n + d)
    x64 = torch.randn(n, d, generator=g, dtype=torch.float64, requires_grad=True)
    y64 = torch.randn(n + 3, d, generator=g, dtype=torch.float64, requires_grad=True)
    up, up2 = torch.randn(n, n, generator=g, dtype=torch.float64), torch.randn(n, n + 3, generator=g, dtype=torch.float64)
    ((x64 @ x64.T) * up).sum().backward()
    want_gx = x64.grad.clone()
    x64.grad = None
    ((x64 @ y64.T) * up2).sum().backward()
    x = x64.detach().float().cuda().requires_grad_(True)
    y = y64.detach().float().cuda().requires_grad_(True)
    got = ops.gram(x)
    assert rel_l2(got.detach().cpu(), (x64 @ x64.T).detach()) < TOL
    (got * up.float().cuda()).sum().backward()
    assert rel_l2(x.grad.cpu(), want_gx) < 1e-5
    x.grad = None
    got2 = ops.gram(x, y)
    assert rel_l2(got2.detach().cpu(), (x64 @ y64.T).detach()) < TOL
    (got2 * up2.float().cuda()).sum().backward()
    assert rel_l2(x.grad.cpu(), x64.grad) < 1e-5 and rel_l2(y.grad.cpu(), y64.grad) < 1e-5


@pytest.mark.parametrize('name', ['mse_mean', 'mse_sum', 'kld_mean', 'kld_sum', 'cosine_mean', 'cosine_sum', 'linear_cka'])
def test_loss_zoo_on_the_device_matches_reference_golden(name):
    """get_loss_fct (framework/trainer/gnndelete_nodeemb.py:19-97) as the trainers use it on the GPU (the non-MSE
    forms run as device tensor ops on the generic autograd path): value and gradient against the reference's own
    functions (tests/golden/losses.npz)."""
    from helpers import load_golden, t
    from gnndelete_amd.framework.trainer.gnndelete_nodeemb import get_loss_fct
    fx = load_golden('losses.npz')
    a = t(fx['a']).cuda().requires_grad_(True)
    v = get_loss_fct(name)(a, t(fx['b']).cuda())
    v.backward()
    assert rel_l2(v.detach().cpu(), fx[f'{name}::value']) < 1e-5
    assert rel_l2(a.grad.cpu(), fx[f'{name}::grad']) < 1e-4


@pytest.mark.parametrize('m,k,n', [(300, 1639, 128), (1000, 96, 64), (77, 8710, 128), (2500, 333, 96), (64, 32, 32),
                                   (5000, 1664, 64), (3000, 1433, 128), (17716, 1639, 128), (20000, 224, 64)])
def test_gemm_wide_matches_fp64(m, k, n, matrix_split):
    """gd_gemm_f32 (K-tiled MFMA GEMM, W streamed through LDS in 32-row chunks, equal unit ranges per block whose
    pieces are added in k order): any reduction width (zero-padded to 32), every supported output width, with bias, on
    all rows and on a gathered row subset; ranges inside one row group, across groups, and whole groups per block."""
    from gnndelete_amd import ops
    g = torch.Generator().manual_seed(m + k)
    x = torch.randn(m, k, generator=g)
    w = torch.randn(k, n, generator=g) / k ** 0.5
    b = torch.randn(n, generator=g)
    want = x.double() @ w.double() + b.double()
    got = ops.gemm_wide(x.cuda(), w.cuda(), b.cuda())
    assert rel_l2(got.cpu(), want) < TOL
    assert torch.equal(got, ops.gemm_wide(x.cuda(), w.cuda(), b.cuda()))          # pieces added in fixed order
    idx = torch.randperm(m, generator=g)[:max(1, m // 3)].sort().values.to(torch.int32)
    out = torch.full((m, n), -7.0).cuda()
    ops.gemm_wide(x.cuda(), w.cuda(), None, idx=idx.cuda(), out=out)
    assert rel_l2(out.cpu()[idx.long()], (x.double() @ w.double())[idx.long()]) < TOL
    rest = torch.ones(m, dtype=torch.bool)
    rest[idx.long()] = False
    assert bool((out.cpu()[rest] == -7.0).all())


@pytest.mark.parametrize('m,in_f,out_f,bias', [(400, 1639, 128, False), (400, 128, 64, True), (90, 20, 4, True),
                                              (700, 200, 96, True), (50, 1639, 6, False)])
def test_dense_autograd_matches_fp64(m, in_f, out_f, bias):
    """ops.dense = torch.nn.functional.linear on the HIP kernels: forward, input gradient and weight gradient (for a
    wide input: the K-tiled kernel on a cached x^T) against fp64 autograd."""
    from gnndelete_amd import ops
    g = torch.Generator().manual_seed(in_f + out_f)
    x = torch.randn(m, in_f, generator=g)
    w = torch.randn(out_f, in_f, generator=g) / in_f ** 0.5
    b = torch.randn(out_f, generator=g) if bias else None
    up = torch.randn(m, out_f, generator=g)
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    bd = b.double().requires_grad_(True) if bias else None
    torch.nn.functional.linear(xd, wd, bd).backward(up.double())
    for const_x in (False, True):
        xg, wg = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True)
        bg = b.cuda().requires_grad_(True) if bias else None
        y = ops.dense(xg, wg, bg, const_x)
        assert rel_l2(y.detach().cpu(), torch.nn.functional.linear(x.double(), w.double(), b.double() if bias else None)) < TOL
        y.backward(up.cuda())
        assert rel_l2(xg.grad.cpu(), xd.grad) < TOL
        assert rel_l2(wg.grad.cpu(), wd.grad) < TOL
        if bias:
            assert rel_l2(bg.grad.cpu(), bd.grad) < TOL


def test_random_walk_kernel_and_device_sampler():
    """gd_random_walk (GraphSAINT's random walks on the device): every step goes to an out-neighbour of the previous node,
    a node without out-edges keeps the walker, the draw is uniform over the neighbours (chi-square on a 6-neighbour hub),
    the same seed gives the same walks and another seed different ones; the device sampler's induced subgraph equals
    the host sampler's for the same node set, attribute slicing included."""
    from gnndelete_amd.framework.data import Data
    from gnndelete_amd.framework.trainer.sampler import RandomWalkSubgraphSampler, make_sampler
    g = torch.Generator().manual_seed(5)
    n, m = 400, 3000
    ei = torch.randint(0, n - 10, (2, m), generator=g)                     # nodes n-10 .. n-1: no edges at all
    ei = torch.cat([ei, torch.tensor([[7] * 6, [11, 12, 13, 14, 15, 16]])], 1)
    ei = ei[:, ei[0] != 7] if False else ei
    data = Data(num_nodes=n, edge_index=ei, x=torch.randn(n, 3, generator=g), edge_flag=torch.rand(ei.shape[1], generator=g) < 0.3,
                note='kept')
    dev_sampler = RandomWalkSubgraphSampler(data.clone().to('cuda'), batch_size=5000, walk_length=3, num_steps=2)
    torch.manual_seed(123)
    nodes = dev_sampler._walk()
    walks = dev_sampler.last_walks.cpu()
    assert walks.shape == (4, 5000) and torch.equal(nodes.cpu(), walks.flatten().unique())
    adj = torch.zeros(n, n, dtype=torch.bool)
    adj[ei[0], ei[1]] = True
    has_out = adj.any(1)
    for s in range(3):
        a, b = walks[s], walks[s + 1]
        assert bool((adj[a, b] | (~has_out[a] & (a == b))).all())
    torch.manual_seed(123)
    dev_sampler._walk()
    assert torch.equal(dev_sampler.last_walks.cpu(), walks)
    torch.manual_seed(124)
    dev_sampler._walk()
    assert not torch.equal(dev_sampler.last_walks.cpu(), walks)
    # uniformity: 60,000 walkers on node 7's out-neighbours (its only out-edges are the 6 added ones, plus random ones)
    from gnndelete_amd import _lib
    from gnndelete_amd._lib import check, ptr
    nb = ei[1][ei[0] == 7]
    start = torch.full((60000,), 7, dtype=torch.long, device='cuda')
    out = torch.empty(2, 60000, dtype=torch.long, device='cuda')
    check(_lib.lib().gd_random_walk(ptr(dev_sampler._rowptr32), ptr(dev_sampler._col32), n, ptr(start), 60000, 1, 99, ptr(out), None), 'walk')
    torch.cuda.synchronize()
    cnt = torch.bincount(out[1].cpu(), minlength=n).double()
    expect = torch.bincount(nb, minlength=n).double() / nb.numel() * 60000
    chi2 = float((((cnt - expect) ** 2)[expect > 0] / expect[expect > 0]).sum())
    assert float(cnt[expect == 0].sum()) == 0 and chi2 < 3 * int((expect > 0).sum()), chi2
    # induced subgraph: device == host for the same node set
    host = RandomWalkSubgraphSampler(data, batch_size=10, walk_length=2, num_steps=1)
    hb, db = host.subgraph(nodes.cpu()), dev_sampler.subgraph(nodes)
    assert db.edge_index.is_cuda and hb.note == db.note == 'kept' and hb.num_nodes == db.num_nodes
    for k in ('edge_index', 'x', 'edge_flag'):
        assert torch.equal(hb[k], db[k].cpu()), k
    assert make_sampler(data, 10, 1).dev.type == 'cuda'


@pytest.mark.parametrize('n,d_in,d_out', [(500, 64, 128), (77, 128, 64), (1000, 32, 32)])
def test_rows_gemm_gated_rank1_matches_fp64(n, d_in, d_out, matrix_split):
    """gd_rows_gemm_gated_rank1_f32: out[r] = gate(in[r] @ W + a[r] p + b[r] q) on a row subset - the GAT input gradient's
    two rank-1 terms moved to the output side of the product that consumes it; rows outside the subset untouched."""
    from gnndelete_amd import ops
    g = torch.Generator().manual_seed(n + d_in)
    x = torch.randn(n, d_in, generator=g)
    w = torch.randn(d_in, d_out, generator=g) / d_in ** 0.5
    a, b = torch.randn(n, generator=g), torch.randn(n, generator=g)
    p, q = torch.randn(d_out, generator=g), torch.randn(d_out, generator=g)
    idx = torch.randperm(n, generator=g)[:n // 2].sort().values.to(torch.int32)
    z = torch.randn(idx.numel(), d_out, generator=g)                  # the forward activations whose sign gates the product
    words = (d_out + 31) // 32
    bits = torch.zeros(idx.numel(), words, dtype=torch.int64)
    for c in range(d_out):
        bits[:, c // 32] |= (z[:, c] > 0).long() << (c % 32)
    bits = torch.where(bits >= 2 ** 31, bits - 2 ** 32, bits).to(torch.int32)
    out = torch.full((n, d_out), -3.0).cuda()
    ops.rows_gemm(x.cuda(), idx.cuda(), w.cuda(), trans_w=False, out=out, gate_bits=bits.cuda().contiguous(),
                  rank1=(a.cuda(), p.cuda(), b.cuda(), q.cuda()))
    rows = idx.long()
    want = (x.double()[rows] @ w.double() + a.double()[rows, None] * p.double() + b.double()[rows, None] * q.double()) * (z > 0)
    assert rel_l2(out.cpu()[rows], want) < TOL
    rest = torch.ones(n, dtype=torch.bool)
    rest[rows] = False
    assert bool((out.cpu()[rest] == -3.0).all())


@pytest.mark.parametrize('n,d', [(300, 64), (77, 128), (1000, 10), (5, 1000), (64, 3)])
def test_rowpair_loss_kernels_match_torch_forms(n, d):
    """gd_rowpair_loss_f32: cosine distance and KL(softmax(b) || softmax(a)) per row with the gradient w.r.t. a, against the
    reference's torch expressions in float64 (framework/trainer/gnndelete_nodeemb.py:18-28), widths on and off the lane
    grid, a zero row (clamped cosine denominator)."""
    import torch.nn.functional as F
    from gnndelete_amd import ops
    g = torch.Generator().manual_seed(n + d)
    a, b = torch.randn(n, d, generator=g) * 2, torch.randn(n, d, generator=g) * 2
    a[0] = 0
    up = torch.rand(n, generator=g) + 0.5
    for kind in ('cosine', 'kld'):
        ad = a.double().requires_grad_(True)
        if kind == 'cosine':
            want = 1 - F.cosine_similarity(ad, b.double())
        else:
            want = F.kl_div(F.log_softmax(ad, -1), b.double().softmax(-1), reduction='none').sum(-1)
        want.backward(up.double())
        ag = a.cuda().requires_grad_(True)
        got = ops.rowpair_loss(ag, b.cuda(), kind)
        got.backward(up.cuda())
        assert rel_l2(got.detach().cpu(), want.detach()) < TOL, kind
        assert rel_l2(ag.grad.cpu(), ad.grad) < 10 * TOL, kind


@pytest.mark.parametrize('n,m,R,d', [(60, 500, 5, 32), (300, 4000, 21, 128), (40, 0, 3, 8), (50, 700, 4, 10)])
def test_segment_softmax_and_typed_weighted_sum_match_autograd(n, m, R, d):
    """The RGAT attention pieces (framework/models/rgat.py:322-337): softmax over the in-edges of a target node across
    relations (gd_segment_softmax_f32 fwd / bwd) and the alpha-weighted typed aggregation with gradients to x and to
    alpha (SpMM over the relation-major CSR, its transpose, gd_rowpair_dot_f32) against dense fp64 autograd."""
    from gnndelete_amd import nn as gnn, ops
    g = torch.Generator().manual_seed(n + d)
    src, dst = torch.randint(0, n, (m,), generator=g), torch.randint(0, n, (m,), generator=g)
    et = torch.randint(0, R, (m,), generator=g)
    x = torch.randn(n, d, generator=g)
    e = torch.randn(m, generator=g)
    up_a, up_m = torch.randn(m, generator=g), torch.randn(R * n, d, generator=g)
    # fp64 reference
    e64 = e.double().requires_grad_(True)
    x64 = x.double().requires_grad_(True)
    mx = torch.full((n,), float('-inf'), dtype=torch.float64).index_reduce(0, dst, e64.detach(), 'amax') if m else None
    if m:
        ex = torch.exp(e64 - mx[dst])
        alpha64 = ex / (torch.zeros(n, dtype=torch.float64).index_add(0, dst, ex)[dst] + 1e-16)
    else:
        alpha64 = e64
    m64 = torch.zeros(R * n, d, dtype=torch.float64).index_add(0, et * n + dst, alpha64[:, None] * x64[src])
    ((alpha64 * up_a.double()).sum() + (m64 * up_m.double()).sum()).backward()
    if m == 0:
        return
    conv = gnn.RGATConv(d, d, R)
    ei, etc = torch.stack([src, dst]).cuda(), et.cuda()
    tc = conv._edge_orders(ei, etc, n)
    eg = e.cuda().requires_grad_(True)
    xg = x.cuda().requires_grad_(True)
    alpha = ops.segment_softmax(eg[tc['ord_d']], tc['rowptr_d'])[tc['inv_d']]
    mm = ops.typed_weighted_sum(xg, alpha[tc['ord_v']], tc)
    ((alpha * up_a.cuda()).sum() + (mm * up_m.cuda()).sum()).backward()
    assert rel_l2(alpha.detach().cpu(), alpha64.detach()) < TOL
    assert rel_l2(mm.detach().cpu(), m64.detach()) < TOL
    assert rel_l2(xg.grad.cpu(), x64.grad) < TOL
    assert rel_l2(eg.grad.cpu(), e64.grad) < 2e-5


@pytest.mark.parametrize('n,m,R,din,dout,nb,coef', [(60, 500, 5, 128, 64, 4, False), (300, 4000, 21, 128, 128, 4, True), (40, 0, 3, 8, 8, None, True),
                                                    (50, 700, 4, 24, 12, None, False), (1000, 30000, 102, 128, 64, 4, True)])
def test_typed_conv_with_trainable_relation_weights_matches_fp64_autograd(n, m, R, din, dout, nb, coef):
    """ops.typed_conv: y_i = sum_e c_e x_j W_r with TRAINABLE relation weights (block-diagonal or dense) - c_e the mean weights
    of RGCNConv or given per-edge coefficients (RGATConv's attention) - forward, input gradient (the typed conv kernels),
    relation-weight gradient (gd_typed_wgrad_f32) and coefficient gradient (gd_typed_edge_dot_f32) against fp64 autograd over
    the plain per-edge formula; relations without edges get a zero gradient; bit-reproducible."""
    from gnndelete_amd import ops
    from gnndelete_amd.graph import TypedNodeCSR
    g = torch.Generator().manual_seed(n + R)
    ei = torch.randint(0, n, (2, m), generator=g)
    et = torch.randint(0, max(R - 1, 1), (m,), generator=g)               # the last relation has no edge
    x = torch.randn(n, din, generator=g)
    blocks = nb or 1
    w = torch.randn(R, blocks, din // blocks, dout // blocks, generator=g) * 0.2
    c = torch.rand(m, generator=g) + 0.1 if coef else None
    gy = torch.randn(n, dout, generator=g)
    # ---- fp64 reference
    x64, w64 = x.double().requires_grad_(), w.double().requires_grad_()
    c64 = c.double().requires_grad_() if coef else None
    if m:
        if coef:
            ce = c64
        else:
            cnt = torch.zeros(R * n, dtype=torch.float64).index_add_(0, et * n + ei[1], torch.ones(m, dtype=torch.float64))
            ce = 1.0 / cnt[et * n + ei[1]]
        msg = torch.einsum('ebi,ebio->ebo', x64[ei[0]].view(m, blocks, -1), w64[et]).reshape(m, dout) * ce[:, None]
        y64 = torch.zeros(n, dout, dtype=torch.float64).index_add(0, ei[1], msg)
    else:
        y64 = (x64.sum() + w64.sum()) * 0.0 + torch.zeros(n, dout, dtype=torch.float64)
    (y64 * gy.double()).sum().backward()
    # ---- kernels
    tg = TypedNodeCSR(ei.cuda(), et.cuda(), n, R)
    wk = (w if nb else w.view(R, din, dout)).cuda().requires_grad_()
    xk = x.cuda().requires_grad_()
    ck = c.cuda().requires_grad_() if coef else None
    y = ops.typed_conv(xk, tg, wk, blocks, ck)
    (y * gy.cuda()).sum().backward()
    tol = 2e-5
    assert rel_l2(y.detach().cpu(), y64.detach()) < tol or m == 0
    assert rel_l2(xk.grad.cpu(), x64.grad) < tol or m == 0
    assert rel_l2(wk.grad.cpu().view_as(w), w64.grad) < tol or m == 0
    assert float(wk.grad.view(R, -1)[R - 1].abs().max()) == 0.0
    if coef and m:
        assert rel_l2(ck.grad.cpu(), c64.grad) < tol
    if m:
        xk2, wk2 = x.cuda().requires_grad_(), wk.detach().clone().requires_grad_()
        y2 = ops.typed_conv(xk2, tg, wk2, blocks, c.cuda() if coef else None)
        (y2 * gy.cuda()).sum().backward()
        assert torch.equal(y2, y) and torch.equal(wk2.grad, wk.grad) and torch.equal(xk2.grad, xk.grad)


@pytest.mark.parametrize('form', ['wave', 'tile'])
@pytest.mark.parametrize('n,m,R,world', [(4000, 60000, 25, 2), (1000, 30000, 51, 3), (300, 2000, 25, 2)])
def test_typed_conv_on_a_row_partition_matches_the_whole_graph(n, m, R, world, form, monkeypatch):
    """TypedNodeCSR(row_range=...): a rank's share of the typed graph - in-edges of its target rows forward, out-edges of
    its source rows (with the GLOBAL mean weights) for the input gradient - gives, on the rank's rows, what the whole
    graph gives: exactly through the (tile, relation) kernel (same summation order per row); through the wave-private
    kernel to fp32 rounding (a run of several slots is summed by a scan tree that depends on where the slots fall in their
    unit, and a tile cut by the partition holds other slots in front of them)."""
    from gnndelete_amd import ops
    from gnndelete_amd.graph import TypedNodeCSR
    monkeypatch.setenv('GD_RGCN_WAVE', '1' if form == 'wave' else '0')
    g = torch.Generator().manual_seed(n + R)
    src, dst = torch.randint(0, n, (m,), generator=g), torch.randint(0, n, (m,), generator=g)
    et = torch.randint(0, R, (m,), generator=g)
    ei = torch.stack([torch.cat([src, dst]), torch.cat([dst, src])]).cuda()
    ety = torch.cat([et, et + R]).cuda()
    nr, nb = 2 * R, 4
    x = torch.randn(n, 128, generator=g).cuda()
    dy = torch.randn(n, 64, generator=g).cuda()
    w2 = (torch.randn(nr, nb, 32, 16, generator=g) * 0.2).cuda()
    full = TypedNodeCSR(ei, ety, n, nr)
    y = torch.zeros(n, 64, device='cuda')
    ops.rgcn_typed_accumulate(full, x, w2, nb, 0, y)
    dx = torch.zeros(n, 128, device='cuda')
    ops.rgcn_typed_accumulate(full, dy, w2, nb, 1, dx)
    chunk = (n + world - 1) // world
    for rank in range(world):
        lo, hi = min(n, rank * chunk), min(n, (rank + 1) * chunk)
        part = TypedNodeCSR(ei, ety, n, nr, row_range=(lo, hi))
        yr = torch.zeros(n, 64, device='cuda')
        ops.rgcn_typed_accumulate(part, x, w2, nb, 0, yr)
        dxr = torch.zeros(n, 128, device='cuda')
        ops.rgcn_typed_accumulate(part, dy, w2, nb, 1, dxr)
        torch.cuda.synchronize()
        if form == 'tile':
            assert torch.equal(yr[lo:hi], y[lo:hi]) and torch.equal(dxr[lo:hi], dx[lo:hi]), rank
        else:
            assert rel_l2(yr[lo:hi].cpu(), y[lo:hi].cpu()) < 1e-6 and rel_l2(dxr[lo:hi].cpu(), dx[lo:hi].cpu()) < 1e-6, rank
        assert float(yr[:lo].abs().sum()) == 0 and float(yr[hi:].abs().sum()) == 0


@pytest.mark.parametrize('compact', [False, True])
@pytest.mark.parametrize('with_add', [True, False])
@pytest.mark.parametrize('n_sel', [37, 1000, 66000, 70001])
def test_del1_forward_loss_and_weight_gradient_in_one_pass(n_sel, with_add, compact):
    """gd_del1_loss_wgrad_f32 (first-layer Del at 128 features + folded loss + weight-gradient partials) against fp64: z and its
    packed sign pattern, the two loss sums, dW after the fixed-order reduction; and against the two launches it replaces."""
    from gnndelete_amd import _lib, ops
    from gnndelete_amd._lib import ptr, check, stream_ptr
    torch.manual_seed(n_sel)
    dev, d, n = 'cuda', 128, 80000
    p = torch.randn(n, d, device=dev)
    w = (torch.eye(d, device=dev) + 0.05 * torch.randn(d, d, device=dev)).contiguous()
    idx = torch.sort(torch.randperm(n, device=dev)[:n_sel]).values.to(torch.int32)
    n_slots = max(1, n_sel - max(3, n_sel // 13))             # (some selected rows carry no loss term)
    slot = torch.full((n_sel,), -1, dtype=torch.int32, device=dev)
    has = torch.randperm(n_sel, device=dev)[:n_slots]
    slot[has] = torch.randperm(n_slots, device=dev).to(torch.int32)
    tm = torch.randn(n_slots, d, device=dev)
    coef = torch.rand(n_slots, device=dev) * 1e-3
    cnt = torch.randint(1, 4, (n_slots,), device=dev).float() * torch.where(torch.rand(n_slots, device=dev) < 0.3, -1.0, 1.0)
    g_add = torch.randn(n, d, device=dev) * 1e-3 if with_add else None
    lib = _lib.lib()
    # compact: one partial matrix per block of the launch (what gd_step_tail_parts_f32 is told); else as many slots as the
    # weight-gradient entries leave, the ones beyond the launch's blocks zeroed (what gd_rows_gemm_wgrad_reduce_f32 counts)
    nb = lib.gd_del1_loss_wgrad_parts(n_sel) if compact else lib.gd_rows_gemm_wgrad_blocks(n_sel)
    assert 1 <= lib.gd_del1_loss_wgrad_parts(n_sel) <= lib.gd_rows_gemm_wgrad_blocks(n_sel)
    z = torch.zeros(n, d, device=dev)
    bits = torch.zeros(n_sel, 4, dtype=torch.int32, device=dev)
    lp = torch.full((2 * nb,), float('nan'), device=dev)
    ws = torch.full((max(1, lib.gd_rows_gemm_wgrad_workspace(n_sel, d, d)),), float('nan'), device=dev)
    check(lib.gd_del1_loss_wgrad_f32(ptr(p), p.stride(0), ptr(idx), n_sel, ptr(w), d, ptr(z), z.stride(0), ptr(bits), ptr(slot), ptr(tm),
                                     ptr(coef), ptr(cnt), ptr(g_add), d if with_add else 0, ptr(lp), ptr(ws), nb, stream_ptr(p.device)),
          'gd_del1_loss_wgrad_f32')
    if compact:
        dw = ws[:nb * d * d].view(nb, d, d).double().sum(0).float()
    else:
        dw = torch.zeros(d, d, device=dev)
        check(lib.gd_rows_gemm_wgrad_reduce_f32(ptr(ws), n_sel, d, d, ptr(dw), 0, None, None, None, None, 0.0, 0.0, 0.0, 0.0,
                                                stream_ptr(p.device)), 'gd_rows_gemm_wgrad_reduce_f32')
    li = idx.long()
    p64, w64 = p.double()[li], w.double()
    z64 = p64 @ w64
    assert float((z[li].double() - z64).norm() / z64.norm()) < 1e-6
    rest = torch.ones(n, dtype=torch.bool, device=dev)
    rest[li] = False
    assert float(z[rest].abs().max()) == 0.0                      # rows outside idx untouched
    zk = z[li]
    want_bits = ((zk > 0).view(n_sel, 4, 32).long() << torch.arange(32, device=dev)).sum(-1)
    want_bits = torch.where(want_bits >= 2 ** 31, want_bits - 2 ** 32, want_bits).to(torch.int32)
    assert torch.equal(bits, want_bits)                            # the pattern of the z the kernel stored
    u = slot.long().clamp(min=0)
    live = (slot >= 0).double()[:, None]
    diff = (z64 - tm.double()[u]) * live
    g64 = coef.double()[u][:, None] * diff
    if with_add:
        g64 = g64 + g_add.double()[li]
    dw64 = p64.t() @ g64
    assert float((dw.double() - dw64).norm() / dw64.norm()) < 2e-6
    sq = (diff * diff).sum(1) * cnt.double()[u].abs()
    neg = (cnt[u] < 0) & (slot >= 0)
    sums = lp.view(-1, 2).double().sum(0)
    assert abs(float(sums[0]) - float(sq[~neg].sum())) <= 1e-5 * float(sq[~neg].sum())
    assert abs(float(sums[1]) - float(sq[neg].sum())) <= 1e-5 * float(sq[neg].sum())
    # the two launches it replaces: same z up to the summation order of another matrix instruction, same sign bits where z is not at 0
    z2 = torch.zeros(n, d, device=dev)
    bits2 = torch.zeros(n_sel, 4, dtype=torch.int32, device=dev)
    ops.rows_gemm(p, idx, w, out=z2, sign_bits=bits2)
    assert float((z2 - z).abs().max()) < 1e-4


def test_two_row_target_loss_jobs_in_one_launch():
    """gd_rowtarget_mse_pair_f32 = two gd_rowtarget_mse_f32 launches (128-wide loss sums only + 64-wide with gradient rows): same
    per-block partials bit for bit, same gradient rows."""
    from gnndelete_amd import _lib
    from gnndelete_amd._lib import ptr, check, stream_ptr
    torch.manual_seed(3)
    dev, n = 'cuda', 9000
    lib = _lib.lib()
    jobs = []
    for d, rows in ((128, 7001), (64, 6500)):
        z = torch.randn(n, d, device=dev)
        ridx = torch.randperm(n, device=dev)[:rows].to(torch.int32)
        tm = torch.randn(rows, d, device=dev)
        coef, cnt = torch.rand(rows, device=dev), torch.randint(1, 5, (rows,), device=dev).float()
        kind = (torch.rand(rows, device=dev) < 0.3).to(torch.int32)
        jobs.append((z, ridx, tm, coef, cnt, kind, rows, d))
    def single(j, dz):
        z, ridx, tm, coef, cnt, kind, rows, d = j
        part = torch.zeros(2 * lib.gd_rowtarget_mse_blocks(rows), device=dev)
        check(lib.gd_rowtarget_mse_f32(ptr(z), z.stride(0), ptr(tm), d, ptr(ridx), ptr(coef), ptr(cnt), ptr(kind), rows, ptr(dz),
                                       dz.stride(0) if dz is not None else 0, None, ptr(part), stream_ptr(z.device)), 'single')
        return part
    dz_b = torch.zeros(n, 64, device=dev)
    pa, pb = single(jobs[0], None), single(jobs[1], dz_b)
    dz_b2 = torch.zeros(n, 64, device=dev)
    qa, qb = torch.zeros_like(pa), torch.zeros_like(pb)
    def args(j, dz, part):
        z, ridx, tm, coef, cnt, kind, rows, d = j
        return (ptr(z), z.stride(0), ptr(tm), d, ptr(ridx), ptr(coef), ptr(cnt), ptr(kind), rows, ptr(dz), dz.stride(0) if dz is not None else 0, ptr(part))
    assert lib.gd_rowtarget_mse_pair_covers(128, 64) == 1 and lib.gd_rowtarget_mse_pair_covers(96, 64) == 0
    check(lib.gd_rowtarget_mse_pair_f32(*args(jobs[0], None, qa), *args(jobs[1], dz_b2, qb), stream_ptr(torch.device(dev))), 'pair')
    assert torch.equal(pa, qa) and torch.equal(pb, qb) and torch.equal(dz_b, dz_b2)
    want = ((jobs[0][0][jobs[0][1].long()].double() - jobs[0][2].double()) ** 2).sum(1) * jobs[0][4].double()
    got = qa.view(-1, 2).double().sum(0)
    k0 = jobs[0][5] == 0
    assert abs(float(got[0]) - float(want[k0].sum())) <= 1e-5 * float(want[k0].sum())
    assert abs(float(got[1]) - float(want[~k0].sum())) <= 1e-5 * float(want[~k0].sum())


@pytest.mark.parametrize('rank1', [False, True])
@pytest.mark.parametrize('n_sel', [1000, 66000])
def test_del1_pass_forms_the_previous_input_gradient_itself(n_sel, rank1):
    """gd_del1_chain_loss_wgrad_f32 = gd_rows_gemm_gated_f32 (dh = (dt[idx] @ W_next) gated by the stored sign pattern) followed by
    gd_del1_loss_wgrad_f32 with g_add = dh: same z, the new sign pattern of that z, same loss sums, same dW - up to the summation
    order of the products."""
    from gnndelete_amd import _lib, ops
    from gnndelete_amd._lib import ptr, check, stream_ptr
    torch.manual_seed(n_sel + 1)
    dev, d, o, n = 'cuda', 128, 64, 80000
    p = torch.randn(n, d, device=dev)
    w = (torch.eye(d, device=dev) + 0.05 * torch.randn(d, d, device=dev)).contiguous()
    w_next = (torch.randn(o, d, device=dev) / 8).contiguous()
    dt = torch.randn(n, o, device=dev) * 1e-3
    idx = torch.sort(torch.randperm(n, device=dev)[:n_sel]).values.to(torch.int32)
    n_slots = max(1, n_sel - n_sel // 9)
    slot = torch.full((n_sel,), -1, dtype=torch.int32, device=dev)
    slot[torch.randperm(n_sel, device=dev)[:n_slots]] = torch.randperm(n_slots, device=dev).to(torch.int32)
    tm = torch.randn(n_slots, d, device=dev)
    coef = torch.rand(n_slots, device=dev) * 1e-3
    cnt = torch.randint(1, 4, (n_slots,), device=dev).float() * torch.where(torch.rand(n_slots, device=dev) < 0.3, -1.0, 1.0)
    prev = torch.randint(-2 ** 31, 2 ** 31 - 1, (n_sel, 4), device=dev, dtype=torch.int64).to(torch.int32)      # the stored pattern
    lib = _lib.lib()
    nb = lib.gd_del1_loss_wgrad_parts(n_sel)
    ws_n = max(1, lib.gd_rows_gemm_wgrad_workspace(n_sel, d, d))

    def reduce_(ws):
        return ws[:nb * d * d].view(nb, d, d).double().sum(0)
    # reference: the two entries it replaces
    dh = torch.zeros(n, d, device=dev)
    r1 = None
    if rank1:      # GATConv's two rank-1 terms in front of the gate (gd_rows_gemm_gated_rank1_f32's epilogue)
        r1 = (torch.randn(n, device=dev) * 1e-3, torch.randn(d, device=dev), torch.randn(n, device=dev) * 1e-3, torch.randn(d, device=dev))
    r1p = tuple(ptr(t) for t in r1) if rank1 else (None, None, None, None)
    ops.rows_gemm(dt, idx, w_next, out=dh, gate_bits=prev.clone(), rank1=r1)
    z_r, bits_r = torch.zeros(n, d, device=dev), prev.clone()
    lp_r, ws_r = torch.zeros(2 * nb, device=dev), torch.zeros(ws_n, device=dev)
    check(lib.gd_del1_loss_wgrad_f32(ptr(p), p.stride(0), ptr(idx), n_sel, ptr(w), d, ptr(z_r), z_r.stride(0), ptr(bits_r), ptr(slot), ptr(tm),
                                     ptr(coef), ptr(cnt), ptr(dh), d, ptr(lp_r), ptr(ws_r), nb, stream_ptr(p.device)), 'ref')
    z_c, bits_c = torch.zeros(n, d, device=dev), prev.clone()
    lp_c, ws_c = torch.zeros(2 * nb, device=dev), torch.zeros(ws_n, device=dev)
    check(lib.gd_del1_chain_loss_wgrad_f32(ptr(p), p.stride(0), ptr(idx), n_sel, ptr(w), d, ptr(z_c), z_c.stride(0), ptr(bits_c), ptr(slot),
                                           ptr(tm), ptr(coef), ptr(cnt), ptr(dt), dt.stride(0), o, ptr(w_next), *r1p, ptr(lp_c), ptr(ws_c), nb,
                                           stream_ptr(p.device)), 'chain')
    assert float((z_c - z_r).abs().max()) < 1e-4
    li = idx.long()
    zk = z_c[li]
    want_bits = ((zk > 0).view(n_sel, 4, 32).long() << torch.arange(32, device=dev)).sum(-1)
    want_bits = torch.where(want_bits >= 2 ** 31, want_bits - 2 ** 32, want_bits).to(torch.int32)
    assert torch.equal(bits_c, want_bits)
    dw_r, dw_c = reduce_(ws_r), reduce_(ws_c)
    assert float((dw_c - dw_r).norm() / dw_r.norm()) < 2e-6
    assert torch.allclose(lp_c.view(-1, 2).double().sum(0), lp_r.view(-1, 2).double().sum(0), rtol=1e-5)
    # the gate really is the STORED pattern: with an all-zero pattern the second stream vanishes
    z_0, bits_0 = torch.zeros(n, d, device=dev), torch.zeros(n_sel, 4, dtype=torch.int32, device=dev)
    ws_0 = torch.zeros(ws_n, device=dev)
    check(lib.gd_del1_chain_loss_wgrad_f32(ptr(p), p.stride(0), ptr(idx), n_sel, ptr(w), d, ptr(z_0), z_0.stride(0), ptr(bits_0), ptr(slot),
                                           ptr(tm), ptr(coef), ptr(cnt), ptr(dt), dt.stride(0), o, ptr(w_next), *r1p, ptr(lp_c), ptr(ws_0), nb,
                                           stream_ptr(p.device)), 'chain0')
    ws_n0 = torch.zeros(ws_n, device=dev)
    check(lib.gd_del1_loss_wgrad_f32(ptr(p), p.stride(0), ptr(idx), n_sel, ptr(w), d, ptr(z_r), z_r.stride(0), ptr(bits_r), ptr(slot), ptr(tm),
                                     ptr(coef), ptr(cnt), None, 0, ptr(lp_r), ptr(ws_n0), nb, stream_ptr(p.device)), 'ref0')
    assert float((reduce_(ws_0) - reduce_(ws_n0)).norm() / reduce_(ws_n0).norm()) < 2e-6
