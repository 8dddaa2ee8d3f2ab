"""torch.autograd.Function wrappers around the C ABI (include/gnndelete_hip.h).

PyTorch only owns memory, streams and the autograd tape here; all arithmetic on the hot path is
the HIP library's.  Every wrapper raises when handed CPU tensors - there is no fallback."""
import os

import torch

from . import _lib
from ._lib import check, ptr, stream_ptr


def matrix_split():
    """How the row GEMMs form their fp32 products: 0 = fp32 matrix instruction, 6 / 9 = split arithmetic on the bf16
    matrix instruction (include/gnndelete_hip.h: gd_set_matrix_split)."""
    return _lib.lib().gd_matrix_split()


def set_matrix_split(n_products):
    _lib.check(_lib.lib().gd_set_matrix_split(int(n_products)), 'gd_set_matrix_split')


def _f32_rows(t):
    """fp32, unit column stride, 16-byte aligned rows (what the kernels require)."""
    if t.device.type != 'cuda':
        raise _lib.GnnDeleteHipError('gnndelete_amd ops need CUDA(HIP) tensors; got a CPU tensor (no CPU fallback)')
    if t.dtype != torch.float32:
        t = t.float()
    if t.dim() != 2 or t.stride(1) != 1 or t.stride(0) < t.shape[1] or (t.stride(0) % 4 and t.shape[1] % 4 == 0):
        t = t.contiguous()
    return t




def _spmm_raw(rowptr, col, val, x, bias, self_coef, n_rows, plan=None, out=None, x_self=None):
    """y = self_coef * x_self + A x + bias (x_self = x unless given; same row pitch).  With a SplitPlan
    (and a float4-able width) the load-balanced kernel is used, otherwise the one-wave-per-row kernel."""
    d = x.shape[1]
    if x_self is not None:
        assert plan is not None and x_self.stride(0) == x.stride(0) and x_self.shape[1] == d
    y = out if out is not None else torch.empty(n_rows, d, dtype=torch.float32, device=x.device)
    if (plan is not None and d % 4 == 0 and d <= 1024 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0
            and os.environ.get('GD_SPMM_TWO_LAUNCH') != '1'):
        # hub rows summed by the four waves of a block inside the same launch (no scratch rows, no fix-up kernel)
        items, n_items, bounds = plan.onepass(d)
        check(_lib.lib().gd_spmm_csr_onepass_f32(ptr(items), n_items, ptr(col), ptr(val), ptr(x), x.stride(0), ptr(y),
                                                 y.stride(0), ptr(bias), float(self_coef), ptr(x_self), d, int(col.shape[0]),
                                                 max(int(x.shape[0]), int(y.shape[0])), ptr(bounds), stream_ptr(x.device)),
              'gd_spmm_csr_onepass_f32')
        return y
    if plan is not None and d % 4 == 0 and d <= 1024 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0:
        scratch = plan.scratch(d, x.device)
        check(_lib.lib().gd_spmm_csr_balanced_f32(ptr(plan.items), plan.n_items, ptr(plan.split), plan.n_split,
                                                  ptr(col), ptr(val), ptr(x), x.stride(0), ptr(y), y.stride(0),
                                                  ptr(bias), float(self_coef), ptr(x_self), ptr(scratch), d, int(col.shape[0]),
                                                  max(int(x.shape[0]), int(y.shape[0])), ptr(plan.xcd_bounds(d)),
                                                  stream_ptr(x.device)),
              'gd_spmm_csr_balanced_f32')
        return y
    assert x_self is None, 'x_self needs the balanced kernel (d % 4 == 0, 16-byte aligned rows)'
    check(_lib.lib().gd_spmm_csr_f32(ptr(rowptr), ptr(col), ptr(val), ptr(x), x.stride(0), ptr(y), y.stride(0),
                                     ptr(bias), float(self_coef), n_rows, d, stream_ptr(x.device)),
          'gd_spmm_csr_f32')
    return y


class _SpMM(torch.autograd.Function):
    """y = self_coef * x + A x + bias over a CSRGraph (A = val or all-ones)."""

    @staticmethod
    def forward(ctx, x, bias, graph, self_coef):
        x = _f32_rows(x)
        ctx.graph, ctx.self_coef = graph, self_coef
        ctx.has_bias = bias is not None
        return _spmm_raw(graph.rowptr, graph.col, graph.val, x, bias, self_coef, graph.n, graph.plan)

    @staticmethod
    def backward(ctx, dy):
        g = ctx.graph
        dx = db = None
        if ctx.needs_input_grad[0]:
            dx = _spmm_raw(g.rowptr_t, g.col_t, g.val_t, _f32_rows(dy), None, ctx.self_coef, g.n, g.plan_t)
        if ctx.has_bias and ctx.needs_input_grad[1]:
            db = dy.sum(0)
        return dx, db, None, None


def spmm(x, graph, bias=None, self_coef=0.0):
    return _SpMM.apply(x, bias, graph, self_coef)


# ------------------------------------------------------------------------------ constant-operand caches
# Derived copies of operands a caller declares constant (transposed frozen weights, zero-padded node features).  The
# caches evict ONE least-recently-used entry at a time, and every tensor they hand out is also reported to the active
# `keep_constants` stores: an engine whose hipGraph has the address of such a copy baked in holds its own reference, so
# an eviction (or a clear) can never free memory a captured graph still replays from.
_SINKS = []


class keep_constants:
    """with keep_constants(store): every cached constant operand handed out inside is added to `store` (a dict)."""

    def __init__(self, store):
        self.store = store

    def __enter__(self):
        _SINKS.append(self.store)
        return self.store

    def __exit__(self, *exc):
        _SINKS.pop()


def _note_constant(t):
    for store in _SINKS:
        store[id(t)] = t
    return t


def _lru_get(cache, key, capacity, build):
    hit = cache.pop(key, None)
    if hit is None:
        while len(cache) >= capacity:
            cache.pop(next(iter(cache)))            # oldest entry only
        hit = build()
    cache[key] = hit                                # (re-)insert as most recently used
    return hit


_WT_CACHE = {}


def _const_weight(w, trans_w):
    """-> (weight, trans_w) for a weight the caller declares constant (frozen backbone): an [out, in] weight is replaced
    by a cached [in, out] copy so that the row kernels fill their LDS weight image with coalesced loads (the [n][k]
    fill is strided).  Keyed on storage address + shape + torch's version counter: an in-place update of the weight
    (an optimizer step) invalidates the copy; the entry keeps the weight alive so the address cannot be recycled."""
    if not trans_w:
        return w, trans_w
    key = (w.data_ptr(), tuple(w.shape), w._version)
    hit = _lru_get(_WT_CACHE, key, 64, lambda: (w.detach().t().contiguous(), w))
    return _note_constant(hit[0]), False


def rows_gemm(inp, idx, w, trans_w=False, bias=None, relu_in=False, out=None, save_in=None, gate_bits=None,
              sign_bits=None, const_w=False, rank1=None):
    """out[idx] = act(inp[idx]) @ (w or w^T) - raw call (no autograd).
    sign_bits (int32 [n_sel, ceil(d_out/32)], written): packed [out > 0] of the rows just produced;
    gate_bits (same layout, read): zero the product where the bit is clear (ReLU backward in the epilogue);
    rank1 = (row_a [N], col_p [d_out], row_b [N], col_q [d_out]), with gate_bits: the product gets
    row_a[r] col_p + row_b[r] col_q added before the gate (MFMA widths only);
    const_w: the weight is constant across calls (see _const_weight)."""
    inp = _f32_rows(inp)
    if const_w:
        w, trans_w = _const_weight(w, trans_w)
    n_sel = inp.shape[0] if idx is None else int(idx.shape[0])
    d_in = inp.shape[1]
    d_out = w.shape[0] if trans_w else w.shape[1]
    assert (w.shape[1] if trans_w else w.shape[0]) == d_in
    if out is None:
        out = torch.empty(inp.shape[0], d_out, dtype=torch.float32, device=inp.device)
    w = w.contiguous()
    n_words = (d_out + 31) // 32
    if gate_bits is not None:
        assert bias is None and not relu_in and save_in is None and sign_bits is None
        assert gate_bits.dtype == torch.int32 and gate_bits.is_contiguous() and gate_bits.numel() >= n_sel * n_words
        if rank1 is not None:
            ra, cp, rb, cq = rank1
            check(_lib.lib().gd_rows_gemm_gated_rank1_f32(ptr(inp), inp.stride(0), ptr(idx), n_sel, ptr(w), d_in, d_out,
                                                          int(trans_w), ptr(gate_bits), ptr(ra), ptr(cp), ptr(rb), ptr(cq),
                                                          ptr(out), out.stride(0), stream_ptr(inp.device)),
                  'gd_rows_gemm_gated_rank1_f32')
            return out
        check(_lib.lib().gd_rows_gemm_gated_f32(ptr(inp), inp.stride(0), ptr(idx), n_sel, ptr(w), d_in, d_out,
                                                int(trans_w), ptr(gate_bits), ptr(out), out.stride(0),
                                                stream_ptr(inp.device)), 'gd_rows_gemm_gated_f32')
        return out
    if sign_bits is not None:
        assert sign_bits.dtype == torch.int32 and sign_bits.is_contiguous() and sign_bits.numel() >= n_sel * n_words
        check(_lib.lib().gd_rows_gemm_signs_f32(ptr(inp), inp.stride(0), ptr(idx), n_sel, ptr(w), d_in, d_out,
                                                int(trans_w), ptr(bias), int(relu_in), ptr(out), out.stride(0),
                                                ptr(save_in), ptr(sign_bits), stream_ptr(inp.device)),
              'gd_rows_gemm_signs_f32')
        return out
    check(_lib.lib().gd_rows_gemm_f32(ptr(inp), inp.stride(0), ptr(idx), n_sel, ptr(w), d_in, d_out, int(trans_w),
                                      ptr(bias), int(relu_in), ptr(out), out.stride(0), ptr(save_in),
                                      stream_ptr(inp.device)), 'gd_rows_gemm_f32')
    return out


def gate_rows(src, idx, sign_bits, out):
    """out[idx[s], :] = src[idx[s], :] where bit c of sign_bits[s] is set, else 0 (ReLU backward from the packed
    [z > 0] pattern the Del-1 kernel emitted): gd_gate_rows_f32, one launch on static buffers (graph-capturable)."""
    n_sel, d = int(idx.shape[0]), src.shape[1]
    if n_sel == 0:
        return out
    if d % 4 == 0 and src.stride(0) % 4 == 0 and out.stride(0) % 4 == 0 and src.is_cuda:
        check(_lib.lib().gd_gate_rows_f32(ptr(src), src.stride(0), ptr(idx), n_sel, ptr(sign_bits), d, ptr(out), out.stride(0),
                                          stream_ptr(src.device)), 'gd_gate_rows_f32')
        return out
    words = (d + 31) // 32
    shifts = torch.arange(32, device=src.device, dtype=torch.int32)
    bits = ((sign_bits[:n_sel, :words, None] >> shifts) & 1).reshape(n_sel, words * 32)[:, :d]
    rows = idx.long()
    out[rows] = src[rows] * bits.to(src.dtype)
    return out


def rows_gemm_select(inp, inp_alt, sel, w, trans_w=False, bias=None, relu_in=False, out=None, idx=None, const_w=False):
    """rows_gemm over a matrix split across two buffers: row r comes from inp_alt where sel[r] (uint8); all rows,
    or the rows listed in idx (int32, written to the same rows of out)."""
    inp, inp_alt = _f32_rows(inp), _f32_rows(inp_alt)
    if const_w:
        w, trans_w = _const_weight(w, trans_w)
    assert inp.shape == inp_alt.shape and inp.stride(0) == inp_alt.stride(0) and sel.dtype == torch.uint8
    n, d_in = inp.shape
    d_out = w.shape[0] if trans_w else w.shape[1]
    if out is None:
        out = torch.empty(n, d_out, dtype=torch.float32, device=inp.device)
    w = w.contiguous()
    if idx is not None:
        n = int(idx.shape[0])
    check(_lib.lib().gd_rows_gemm_select_f32(ptr(inp), ptr(inp_alt), ptr(sel), inp.stride(0), ptr(idx), n, ptr(w), d_in,
                                             d_out, int(trans_w), ptr(bias), int(relu_in), ptr(out), out.stride(0),
                                             stream_ptr(inp.device)), 'gd_rows_gemm_select_f32')
    return out


def rows_gemm_accumulate_ok(n_rows, d_in, d_out):
    """Does gd_rows_gemm_accumulate_f32 take this shape (the weight-stationary form's: widths in {64, 128}, >= 65,536 rows)?"""
    return matrix_split() == 0 and bool(_lib.lib().gd_rows_gemm_ws_covers(int(n_rows), int(d_in), int(d_out)))


def rows_gemm_accumulate_(out, inp, idx, w, trans_w=False, const_w=False):
    """out[rows] += inp[rows] @ w (w^T with trans_w), in place (raw, no autograd): gd_rows_gemm_accumulate_f32."""
    inp = _f32_rows(inp)
    if const_w:
        w, trans_w = _const_weight(w, trans_w)
    n, d_in = inp.shape
    d_out = w.shape[0] if trans_w else w.shape[1]
    assert out.shape[1] == d_out and out.dtype == torch.float32 and out.stride(1) == 1
    if idx is not None:
        n = int(idx.shape[0])
    w = w.contiguous()
    check(_lib.lib().gd_rows_gemm_accumulate_f32(ptr(inp), inp.stride(0), ptr(idx), n, ptr(w), d_in, d_out, int(trans_w), ptr(out),
                                                 out.stride(0), stream_ptr(inp.device)), 'gd_rows_gemm_accumulate_f32')
    return out


def rows_gemm_dots_ok(d_in, d_out, n_rows=0, selected=False):
    """Whether fusing the row dots into the GEMM epilogue pays (the engine's choice; the entry itself takes 128 too).
    n_rows / selected (rows from two buffers AND an index list): what decides whether the weight-stationary kernel takes the
    call - it carries the dots at 128 outputs as well (csrc/rows_gemm_ws.hip, MODE 3)."""
    if (d_in == 128 and d_out in (64, 128) and os.environ.get('GD_ROWS_GEMM_WS_EPI', '1') != '0'
            and _lib.lib().gd_rows_gemm_ws_covers(int(n_rows), d_in, d_out)):
        return True                       # (round 6: also with a selector AND an index list - n_rows = the listed rows)
    # d_out <= 64: the 128-wide variant of the LDS-operand kernel runs out of registers (spills: measured 7 % slower than
    # the separate pass)
    return d_in % 32 == 0 and d_out % 32 == 0 and d_out <= 64 and d_in * d_out * 4 <= 64 * 1024


def rows_gemm_dots(inp, w, u1, u2, inp_alt=None, sel=None, relu_in=False, out=None, idx=None, dots_out=None,
                   const_w=False):
    """out = act(inp or inp_alt) @ w^T (w [d_out, d_in]) and, from the same pass, a1 = out @ u1, a2 = out @ u2
    (gd_rows_gemm_dots_f32; MFMA widths only - see rows_gemm_dots_ok)."""
    inp = _f32_rows(inp)
    n, d_in = inp.shape
    d_out = w.shape[0]
    assert d_in % 32 == 0 and d_out % 32 == 0 and d_out <= 128 and d_in * d_out * 4 <= 64 * 1024 and w.shape[1] == d_in
    if inp_alt is not None:
        inp_alt = _f32_rows(inp_alt)
        assert inp_alt.shape == inp.shape and inp_alt.stride(0) == inp.stride(0) and sel.dtype == torch.uint8
    if out is None:
        out = torch.empty(n, d_out, dtype=torch.float32, device=inp.device)
    if dots_out is not None:                      # caller-owned (e.g. zero-initialised, for a row subset)
        a1, a2 = dots_out
    else:
        a1 = torch.empty(n, dtype=torch.float32, device=inp.device)
        a2 = torch.empty(n, dtype=torch.float32, device=inp.device)
    w, trans = _const_weight(w, True) if const_w else (w.contiguous(), True)
    u1, u2 = u1.reshape(-1).contiguous(), u2.reshape(-1).contiguous()
    n_rows = n if idx is None else int(idx.shape[0])
    check(_lib.lib().gd_rows_gemm_dots_f32(ptr(inp), ptr(inp_alt), ptr(sel), inp.stride(0), ptr(w), d_in, d_out, int(trans), None,
                                           int(relu_in), ptr(out), out.stride(0), ptr(idx), n_rows, ptr(u1), ptr(u2),
                                           ptr(a1), ptr(a2), stream_ptr(inp.device)), 'gd_rows_gemm_dots_f32')
    return out, a1, a2


_PAD_CACHE = {}          # copies derived from the node features (large: a handful of entries)
_WPAD_CACHE = {}         # copies derived from weights (small, many)


def _cached(tag, t, build):
    """build(t) cached per tensor content identity (storage address, shape, strides, torch's version counter); the
    entry keeps `t` alive so the address cannot be recycled.  For operands that stay constant across calls (the
    node features, frozen weights): padded / transposed copies are made once.  Feature-derived and weight-derived
    copies live in separate LRU caches, so a stream of weight entries cannot push out a 100 MB padded feature matrix."""
    key = (tag, t.data_ptr(), tuple(t.shape), tuple(t.stride()), t._version)
    cache, cap = (_WPAD_CACHE, 64) if tag.startswith(('padw', 'wT')) else (_PAD_CACHE, 8)
    hit = _lru_get(cache, key, cap, lambda: (build(t), t))
    return _note_constant(hit[0])


def _pad_cols32(t, mult=32):
    k = t.shape[1]
    kp = (k + mult - 1) // mult * mult
    if kp == k and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0 and t.stride(1) == 1:
        return t
    out = torch.zeros(t.shape[0], kp, dtype=torch.float32, device=t.device)
    out[:, :k] = t
    return out


def mfma_out_width(n):
    return n % 32 == 0 and 32 <= n <= 128


def gemm_wide(x, w_kn, bias=None, idx=None, out=None, const_x=False, const_w=False):
    """out[rows] = x[rows] @ w_kn (+ bias) with a reduction dimension of any width (gd_gemm_f32): x [M, K], w_kn [K, N]
    (the transpose of a torch Linear weight), N in {32, 64, 96, 128}.  K is zero-padded to a multiple of 32 (128 with split
    arithmetic on; x: a
    padded copy, cached when const_x says the matrix does not change between calls - the node features; const_w likewise
    for the zero-padded weight)."""
    assert mfma_out_width(w_kn.shape[1]) and w_kn.shape[0] == x.shape[1]
    x = _f32_rows(x)
    mult = 128 if matrix_split() == 6 else 32            # the split form walks K in macro chunks of 128
    xp = _cached(f'padx{mult}', x, lambda t: _pad_cols32(t, mult)) if const_x else _pad_cols32(x, mult)
    k = xp.shape[1]
    if w_kn.shape[0] != k or not w_kn.is_contiguous():
        def pad_rows(t):
            wp = torch.zeros(k, t.shape[1], dtype=torch.float32, device=x.device)
            wp[:t.shape[0]] = t
            return wp
        # (a frozen weight is padded once, not with a fill + copy in front of every product)
        w_kn = _cached(f'padw{k}', w_kn, pad_rows) if const_w else pad_rows(w_kn)
    n = w_kn.shape[1]
    n_rows = x.shape[0] if idx is None else int(idx.shape[0])
    if out is None:
        out = torch.empty(x.shape[0], n, dtype=torch.float32, device=x.device)
    ws_n = _lib.lib().gd_gemm_f32_workspace(n_rows, k, n)
    ws = torch.empty(max(int(ws_n), 4), dtype=torch.float32, device=x.device)
    check(_lib.lib().gd_gemm_f32(ptr(xp), xp.stride(0), ptr(idx), n_rows, ptr(w_kn), k, n, ptr(bias), ptr(out), out.stride(0),
                                 ptr(ws), stream_ptr(x.device)), 'gd_gemm_f32')
    return out


def _small_weight(d_in, d_out):
    """Does [d_in, d_out] fit the whole-weight-in-LDS row kernel (rows_gemm_mfma_kernel)?"""
    return d_in % 32 == 0 and mfma_out_width(d_out) and d_in * 32 * (4 if d_out == 96 else d_out // 32) * 4 <= 64 * 1024


class _Dense(torch.autograd.Function):
    """y = x @ weight^T (+ bias), weight [out, in] as torch.nn.Linear / torch_geometric's Linear store it
    (framework/models/gcn.py:11-12 ...): every product on the HIP matrix-core kernels where the widths allow -
    forward (whole weight in LDS for in <= 128, K-tiled for wider inputs), input gradient, weight gradient (row
    reduction kernel; for a wide input the K-tiled kernel on a cached x^T); odd widths run the generic kernels, and a shape
    without a kernel raises - this op never falls back to torch's matmul.  (What of the package still calls torch / rocBLAS
    products is listed in DESIGN.md section 4: two [1, 64] x [64, 128] constants folded at engine set-up, and RGCNConv / RGATConv's
    einsum over relation weights whose block widths the typed kernels do not take - odd or above 128 - none of them on a
    BASELINE configuration.)"""

    @staticmethod
    def forward(ctx, x, weight, bias, const_x):
        x = _f32_rows(x)
        out_f, in_f = weight.shape
        ctx.const_x = const_x
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        w = weight.detach()
        b = bias.detach().contiguous() if bias is not None else None
        if _small_weight(in_f, out_f) and x.stride(0) % 4 == 0:
            return rows_gemm(x, None, w, trans_w=True, bias=b)
        if mfma_out_width(out_f):
            # a trainable weight changes every optimizer step: caching its transpose would miss every time and only
            # churn the cache - transpose on the fly; a frozen weight's transpose is made once
            wt = w.t().contiguous() if weight.requires_grad else _cached('wT', w, lambda t: t.t().contiguous())
            return gemm_wide(x, wt, b, const_x=const_x)
        if in_f <= 1024:
            return rows_gemm(x, None, w, trans_w=True, bias=b)            # any widths: the one-wave-per-row kernel
        # a wide input with an output width the K-tiled kernel has no tile for (a class count, ...): blocks of <= 128 output
        # columns, each zero-padded to a multiple of 32
        y = torch.empty(x.shape[0], out_f, dtype=torch.float32, device=x.device)
        for j in range(0, out_f, 128):
            nb = min(128, out_f - j)
            npad = (nb + 31) // 32 * 32
            wt = torch.zeros(in_f, npad, dtype=torch.float32, device=x.device)
            wt[:, :nb] = w[j:j + nb].t()
            bp = None
            if b is not None:
                bp = torch.zeros(npad, dtype=torch.float32, device=x.device)
                bp[:nb] = b[j:j + nb]
            y[:, j:j + nb] = gemm_wide(x, wt, bp, const_x=const_x)[:, :nb]
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = _f32_rows(dy)
        out_f, in_f = weight.shape
        dx = dw = db = None
        w = weight.detach()
        if ctx.needs_input_grad[0]:
            if _small_weight(out_f, in_f) or (out_f <= 1024 and in_f <= 1024):
                dx = rows_gemm(dy, None, w, trans_w=False)                  # dy [*, out] @ W [out, in]
            elif mfma_out_width(in_f):
                dx = gemm_wide(dy, w)                                       # [*, out] @ [out, in], the K-tiled kernel
            elif out_f <= 1024:
                # a wide, odd input width (the reference never asks for this gradient: its wide inputs are the node features):
                # 128 columns of dx at a time - dy [*, out] @ W[:, j : j + 128] - into a padded buffer (16-byte row pitch)
                wide = (in_f + 127) // 128 * 128
                dxp = torch.empty(dy.shape[0], wide, dtype=torch.float32, device=dy.device)
                wp = torch.zeros(out_f, wide, dtype=torch.float32, device=dy.device)
                wp[:, :in_f] = w
                for j in range(0, wide, 128):
                    rows_gemm(dy, None, wp[:, j:j + 128].contiguous(), trans_w=False, out=dxp[:, j:j + 128])
                dx = dxp[:, :in_f].contiguous()
            else:
                raise NotImplementedError(f'ops.dense: the input gradient of a {in_f} -> {out_f} product has no kernel; there is no '
                                          'vendor-BLAS fallback (the reference never asks for it: its wide inputs are the node features)')
        if ctx.needs_input_grad[1]:
            m = x.shape[0]
            if out_f % 32 == 0 and in_f % 32 == 0 and out_f <= 128 and in_f <= 128 and out_f != 96 and in_f != 96:
                dw = rows_gemm_wgrad(dy, None, x, None, m)                  # dy^T x  [out, in]
            elif mfma_out_width(out_f) and m >= 32:
                # dW^T [in, out] = x^T dy: the K-tiled kernel with the ROWS of x as the reduction dimension
                def transpose_pad(t):
                    mp = (t.shape[0] + 31) // 32 * 32
                    xt = torch.zeros(t.shape[1], mp, dtype=torch.float32, device=t.device)
                    xt[:, :t.shape[0]] = t.t()
                    return xt
                xt = _cached('xT', x, transpose_pad) if ctx.const_x else transpose_pad(x)
                dyp = torch.zeros(xt.shape[1], out_f, dtype=torch.float32, device=dy.device)
                dyp[:m] = dy
                dw = gemm_wide(xt, dyp).t()
            else:
                dw = rows_gemm_wgrad(dy, None, x, None, m)                  # any widths (one thread per element of dW, split over row blocks)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.sum(0)
        return dx, dw, db, None


def dense(x, weight, bias=None, const_x=False):
    """torch.nn.functional.linear on the HIP kernels (see _Dense); const_x: x is the same matrix on every call (the
    node features) - padded / transposed copies of it are cached."""
    return _Dense.apply(x, weight, bias, const_x)


def gram(x, y=None):
    """x @ y^T (y = x unless given) for device tensors, 128 output columns per product on the kernels of `dense` (matrix-core
    forms where the widths allow, the generic row kernel otherwise; autograd through both operands): the Gram matrices of the
    CKA losses (framework/trainer/gnndelete_nodeemb.py:30-66) and the all-pairs logits z z^T of Trainer.test (base.py:288) -
    the last two products of the path that went through torch's matmul (rocBLAS).  Host tensors (the CPU-side loss-zoo checks)
    take the torch expression."""
    y = x if y is None else y
    if not x.is_cuda:
        return x @ y.t()
    return torch.cat([dense(x, y[j:j + 128].contiguous()) for j in range(0, y.shape[0], 128)], 1)


def rows_gemm_wgrad(a, a_idx, g, g_idx, n_sel, relu_mask=None, out=None, accumulate=False, g_add=None):
    """dW[d_a, d_b] (+)= sum_s a[a_idx[s]]^T (mask(g) + g_add)[g_idx[s]] - raw call."""
    a, g = _f32_rows(a), _f32_rows(g)
    d_a, d_b = a.shape[1], g.shape[1]
    if out is None:
        out = torch.zeros(d_a, d_b, dtype=torch.float32, device=a.device)
        accumulate = False
    ws = torch.empty(max(1, _lib.lib().gd_rows_gemm_wgrad_workspace(n_sel, d_a, d_b)), dtype=torch.float32,
                     device=a.device)
    if relu_mask is not None:
        relu_mask = _f32_rows(relu_mask)
        assert relu_mask.stride(0) == g.stride(0)
    if g_add is not None:
        g_add = _f32_rows(g_add)
        assert g_add.stride(0) == g.stride(0) and g_add.shape == g.shape
    check(_lib.lib().gd_rows_gemm_wgrad_f32(ptr(a), a.stride(0), ptr(a_idx), ptr(g), g.stride(0), ptr(g_idx),
                                            ptr(relu_mask), ptr(g_add), n_sel, d_a, d_b, ptr(out), int(accumulate), ptr(ws),
                                            stream_ptr(a.device)), 'gd_rows_gemm_wgrad_f32')
    return out


class _DelRows(torch.autograd.Function):
    """DeletionLayer.forward: copy of x whose rows ``idx`` are multiplied by W."""

    @staticmethod
    def forward(ctx, x, w, idx):
        x = _f32_rows(x)
        z = x.clone()
        rows_gemm(x, idx, w, out=z)
        ctx.idx = idx
        # W is only needed for the input gradient; not saving it otherwise lets the layer-wise
        # trainer step W in place between the two backward passes (frozen-backbone semantics)
        ctx.save_for_backward(x, w if ctx.needs_input_grad[0] else None)
        return z

    @staticmethod
    def backward(ctx, dz):
        x, w = ctx.saved_tensors
        idx = ctx.idx
        dz = _f32_rows(dz)
        dx = dw = None
        if ctx.needs_input_grad[1]:
            dw = rows_gemm_wgrad(x, idx, dz, idx, int(idx.shape[0]))
        if ctx.needs_input_grad[0]:
            dx = dz.clone()
            rows_gemm(dz, idx, w, trans_w=True, out=dx)
        return dx, dw, None


def del_rows(x, w, idx):
    return _DelRows.apply(x, w, idx)


# ------------------------------------------------------------------------------ GAT
def _pow2(d):
    return d >= 4 and (d & (d - 1)) == 0 and d <= 1024


def _ws_buf(bufs, key, shape, device, zero=False):
    """A work buffer of a raw op: fresh when bufs is None, else the caller's persistent one under `key` (made on first use).
    Engines whose step is cut into several hipGraphs pass a dict they own, so that what one captured segment writes and a
    later one reads lives in memory the ENGINE holds (not in a capture-time allocation only a graph pool keeps alive)."""
    if bufs is None:
        return (torch.zeros if zero else torch.empty)(shape, dtype=torch.float32, device=device)
    t = bufs.get(key)
    if t is None or tuple(t.shape) != tuple(shape if isinstance(shape, (tuple, list)) else (shape,)):
        t = bufs[key] = (torch.zeros if zero else torch.empty)(shape, dtype=torch.float32, device=device)
    return t


def _gat_onepass():
    """GATConv's edge-gradient pass on the one-launch items (hub rows as groups inside the launch); GD_GAT_PIECES=1: the piece form
    with its second edge pass and fix-up launches (A/B switch, read per call).  The forward keeps the piece form."""
    return os.environ.get('GD_GAT_PIECES') != '1'


def gat_forward_raw(graph, h, a_src, a_dst, bias, slope, out=None, plan=None, bufs=None):
    """Fused edge-softmax aggregation; returns (y, rowmax, rowsum) - balanced kernels when the
    width allows, else the one-wave-per-row kernel (then rowmax is the saved alpha, rowsum None).
    bufs: see _ws_buf (row statistics / saved alpha land in the caller's persistent buffers)."""
    n, d = graph.n, h.shape[1]
    y = out if out is not None else torch.empty(n, d, dtype=torch.float32, device=h.device)
    plan = plan or graph.plan               # a plan over a row subset: only those rows of y / rowmax / rowsum are written
    if _pow2(d) and h.stride(0) % 4 == 0:
        rowmax = _ws_buf(bufs, 'rowmax', (n,), h.device)
        rowsum = _ws_buf(bufs, 'rowsum', (n,), h.device)
        scratch = plan.scratch_flat('gat', _lib.lib().gd_gat_balanced_scratch(plan.n_slots, d), h.device)
        check(_lib.lib().gd_gat_aggregate_balanced_f32(
            ptr(plan.items), plan.n_items, ptr(plan.split), plan.n_split, plan.n_slots, ptr(graph.col), ptr(a_src),
            ptr(a_dst), ptr(h), h.stride(0), ptr(y), y.stride(0), ptr(bias), ptr(rowmax), ptr(rowsum), ptr(scratch),
            float(slope), d, graph.nnz, int(h.shape[0]), stream_ptr(h.device)), 'gd_gat_aggregate_balanced_f32')
        return y, rowmax, rowsum
    alpha = _ws_buf(bufs, 'alpha', (graph.nnz,), h.device)
    check(_lib.lib().gd_gat_aggregate_f32(ptr(graph.rowptr), ptr(graph.col), ptr(a_src), ptr(a_dst), ptr(h),
                                          h.stride(0), ptr(y), y.stride(0), ptr(bias), ptr(alpha), float(slope), n, d,
                                          stream_ptr(h.device)), 'gd_gat_aggregate_f32')
    return y, alpha, None


def gat_backward_raw(graph, h, a_src, a_dst, rowmax, rowsum, dy, slope, plan=None, plan_t=None, bufs=None):
    """-> (dh message path, da_src, da_dst).  plan / plan_t: work items of a row subset for the target-major edge
    gradients and the source-major aggregation (rows outside are not produced).  bufs: see _ws_buf."""
    g = graph
    n, d = g.n, h.shape[1]
    dev = h.device
    da_dst = _ws_buf(bufs, 'da_dst', (n,), dev)
    if rowsum is None:                     # saved alpha from the row kernel
        alpha = rowmax
        de = _ws_buf(bufs, 'de', (g.nnz,), dev)
        dh = _ws_buf(bufs, 'dh', tuple(h.shape), dev)
        da_src = _ws_buf(bufs, 'da_src', (n,), dev)
        check(_lib.lib().gd_gat_aggregate_bwd_f32(
            ptr(g.rowptr), ptr(g.col), ptr(alpha), ptr(g.rowptr_t), ptr(g.col_t), ptr(g.perm_t), ptr(a_src),
            ptr(a_dst), ptr(h), h.stride(0), ptr(dy), dy.stride(0), ptr(dh), dh.stride(0), ptr(da_src), ptr(da_dst),
            ptr(de), float(slope), n, d, stream_ptr(dev)), 'gd_gat_aggregate_bwd_f32')
        return dh, da_src, da_dst
    subset = plan is not None
    plan = plan or g.plan
    # per edge (alpha, score gradient) interleaved; edges of rows outside a subset are never computed: keep them finite
    # (zero) for the transposition pass
    ade = _ws_buf(bufs, 'ade', (g.nnz, 2), dev, zero=subset)
    t_row = _ws_buf(bufs, 't_row', (n,), dev)
    if _gat_onepass():
        # one launch: a hub row's four members sum t in LDS and finish the score gradients of their own edges (no second edge
        # pass, no fix-up launches)
        items, n_items, bounds = plan.onepass(d, multirow=1)
        check(_lib.lib().gd_gat_edge_grads_balanced_f32(
            ptr(items), n_items, None, 0, ptr(g.col), ptr(a_src), ptr(a_dst), ptr(rowmax), ptr(rowsum), ptr(h), h.stride(0),
            ptr(dy), dy.stride(0), ptr(ade), ptr(da_dst), ptr(t_row), None, float(slope), d, g.nnz, ptr(bounds), stream_ptr(dev)),
            'gd_gat_edge_grads_balanced_f32')
    else:
        scratch = plan.scratch_flat('gat_bwd', max(4, plan.n_slots), dev)
        check(_lib.lib().gd_gat_edge_grads_balanced_f32(
            ptr(plan.items), plan.n_items, ptr(plan.split), plan.n_split, ptr(g.col), ptr(a_src), ptr(a_dst), ptr(rowmax),
            ptr(rowsum), ptr(h), h.stride(0), ptr(dy), dy.stride(0), ptr(ade), ptr(da_dst), ptr(t_row),
            ptr(scratch), float(slope), d, g.nnz, None, stream_ptr(dev)), 'gd_gat_edge_grads_balanced_f32')
    da_src = _ws_buf(bufs, 'da_src', (n,), dev)
    pt = plan_t or g.plan_t
    dh = None if bufs is None else _ws_buf(bufs, 'dh', (n, dy.shape[1]), dev)
    if (d in (64, 128) and dy.stride(0) % 4 == 0 and dy.data_ptr() % 16 == 0 and n < 2 ** 24 and dy.stride(0) * 4 < 2 ** 24
            and n * dy.stride(0) * 4 < 2 ** 32 and os.environ.get('GD_GAT_TRANSPOSE_PASS') != '1'):
        # the source-major aggregation reads (alpha, score gradient) through the transpose permutation itself and sums the score
        # gradients per source row: no transposition pass (gd_spmm_csr_onepass_aux_f32)
        if dh is None:
            dh = torch.empty(n, d, dtype=torch.float32, device=dev)
        if plan_t is not None:
            da_src.zero_()                  # (rows outside a subset are not visited)
        items, n_items, bounds = pt.onepass(d)
        check(_lib.lib().gd_spmm_csr_onepass_aux_f32(ptr(items), n_items, ptr(g.col_t), ptr(g.perm_t), ptr(ade), ptr(dy), dy.stride(0),
                                                     ptr(dh), dh.stride(0), ptr(da_src), d, g.nnz, n, ptr(bounds), stream_ptr(dev)),
              'gd_spmm_csr_onepass_aux_f32')
        return dh, da_src, da_dst
    alpha_t = _ws_buf(bufs, 'alpha_t', (g.nnz,), dev)
    check(_lib.lib().gd_gat_transpose_edges_f32(ptr(g.rowptr_t), ptr(g.perm_t), ptr(ade), n, ptr(alpha_t),
                                                ptr(da_src), stream_ptr(dev)), 'gd_gat_transpose_edges_f32')
    dh = _spmm_raw(g.rowptr_t, g.col_t, alpha_t, dy, None, 0.0, n, pt, out=dh)
    return dh, da_src, da_dst


def rank1_add2_(y, a, u, b, v):
    """y[i,:] += a[i] * u + b[i] * v, in place (raw, no autograd)."""
    n, d = y.shape
    u, v = u.reshape(-1).contiguous(), v.reshape(-1).contiguous()
    if d % 4 or y.stride(0) % 4 or y.stride(1) != 1:
        return y.addcmul_(a[:, None], u.view(1, -1)).addcmul_(b[:, None], v.view(1, -1))
    check(_lib.lib().gd_rank1_add2_f32(ptr(y), y.stride(0), n, d, ptr(a), ptr(u), ptr(b), ptr(v),
                                       stream_ptr(y.device)), 'gd_rank1_add2_f32')
    return y


def row_dots(h, v1, v2, bufs=None):
    """(h @ v1, h @ v2) in one pass (raw, no autograd).  bufs: see _ws_buf."""
    h = _f32_rows(h)
    n, d = h.shape
    a1 = _ws_buf(bufs, 'a1', (n,), h.device)
    a2 = _ws_buf(bufs, 'a2', (n,), h.device)
    v1, v2 = v1.reshape(-1).contiguous(), v2.reshape(-1).contiguous()
    if d % 4 or h.stride(0) % 4:
        # odd widths: the two dots as a [n, d] x [d, 2] product on the one-wave-per-row kernel
        both = rows_gemm(h, None, torch.stack([v1, v2]), trans_w=True)
        a1.copy_(both[:, 0])
        a2.copy_(both[:, 1])
        return a1, a2
    check(_lib.lib().gd_row_dots_f32(ptr(h), h.stride(0), n, d, ptr(v1), ptr(v2), ptr(a1), ptr(a2),
                                     stream_ptr(h.device)), 'gd_row_dots_f32')
    return a1, a2


class _GatAggregate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, a_src, a_dst, bias, graph, slope):
        h = _f32_rows(h)
        a_src, a_dst = a_src.contiguous().float(), a_dst.contiguous().float()
        y, rowmax, rowsum = gat_forward_raw(graph, h, a_src, a_dst, bias, slope)
        ctx.graph, ctx.slope, ctx.has_bias, ctx.no_sum = graph, slope, bias is not None, rowsum is None
        if any(ctx.needs_input_grad[:3]):
            ctx.save_for_backward(h, a_src, a_dst, rowmax, rowsum if rowsum is not None else rowmax)
        return y

    @staticmethod
    def backward(ctx, dy):
        h, a_src, a_dst, rowmax, rowsum = ctx.saved_tensors
        dh, da_src, da_dst = gat_backward_raw(ctx.graph, h, a_src, a_dst, rowmax, None if ctx.no_sum else rowsum,
                                              _f32_rows(dy), ctx.slope)
        db = dy.sum(0) if ctx.has_bias and ctx.needs_input_grad[3] else None
        return dh, da_src, da_dst, db, None, None


def gat_aggregate(h, a_src, a_dst, graph, bias=None, slope=0.2):
    return _GatAggregate.apply(h, a_src, a_dst, bias, graph, slope)


# ------------------------------------------------------------------------------ R-GCN mean
class _RgcnMean(torch.autograd.Function):
    """[R, n, d] per-relation mean aggregates of x over a typed CSR (graph.build_typed_csr)."""

    @staticmethod
    def forward(ctx, x, typed, num_rel, n):
        x = _f32_rows(x)
        rowptr, col, rowptr_t, col_t, inv_t = typed
        y = torch.empty(num_rel, n, x.shape[1], dtype=torch.float32, device=x.device)
        check(_lib.lib().gd_rgcn_mean_f32(ptr(rowptr), ptr(col), ptr(x), x.stride(0), ptr(y), y.stride(1), num_rel, n,
                                          x.shape[1], stream_ptr(x.device)), 'gd_rgcn_mean_f32')
        ctx.typed, ctx.n = typed, n
        return y

    @staticmethod
    def backward(ctx, dy):
        _, _, rowptr_t, col_t, inv_t = ctx.typed
        dy2 = _f32_rows(dy.reshape(-1, dy.shape[-1]))
        dx = _spmm_raw(rowptr_t, col_t, inv_t, dy2, None, 0.0, ctx.n)
        return dx, None, None, None


def rgcn_mean(x, typed, num_rel, n):
    return _RgcnMean.apply(x, typed, num_rel, n)


_RGCN_PACK_CACHE = {}


def rgcn_packed_weight(weight, n_blocks, d_in, d_out, trans):
    """The relation weights [R, n_blocks, ib, ob] in the MFMA lane order of gd_rgcn_tile_conv_f32 for one direction,
    cached per (storage, shape, version): a frozen backbone packs once, an updated weight repacks."""
    kl = int(_lib.lib().gd_rgcn_tile_kl(d_in, d_out, n_blocks, int(trans)))
    assert kl > 0
    key = (weight.data_ptr(), tuple(weight.shape), weight._version, int(trans))
    def build():
        wc = weight.detach().contiguous()
        packed = torch.empty(wc.shape[0] * (d_out // 16) * (kl // 16) * 256, dtype=torch.float32, device=wc.device)
        check(_lib.lib().gd_rgcn_pack_weight_f32(ptr(wc), wc.shape[0], n_blocks, d_in, d_out, int(trans), ptr(packed),
                                                 stream_ptr(wc.device)), 'gd_rgcn_pack_weight_f32')
        return (packed, weight)
    return _note_constant(_lru_get(_RGCN_PACK_CACHE, key, 16, build)[0])


def rgcn_wave_form(d_in, d_out, n_blocks, n_nodes=None, ldx=None):
    """Whether the typed conv of these widths runs on the wave-private kernel (GD_RGCN_WAVE=0: the tile kernel instead).
    n_nodes / ldx (elements): the kernel addresses source rows with 24-bit row ids and pitches inside a 4 GiB buffer
    (gd_rgcn_wave_conv_f32 returns GD_E_DIM beyond that, e.g. > 8.3 M entities at 128 floats) - a graph that exceeds them
    falls through to the tile / node-major kernels instead of raising."""
    if os.environ.get('GD_RGCN_WAVE', '1') == '0' or int(_lib.lib().gd_rgcn_wave_covers(d_in, d_out, n_blocks)) <= 0:
        return False
    if n_nodes is not None and (n_nodes >= 2 ** 24 or (ldx is not None and (ldx * 4 >= 2 ** 24 or (n_nodes + 1) * ldx * 4 >= 2 ** 32))):
        return False
    return True


def rgcn_wave_relu_ok(tg, x, y, n_blocks, trans=0):
    """Whether rgcn_typed_accumulate(..., relu_in=True) is available for this call (the wave-private kernel takes it)."""
    d_in, d_out = x.shape[1], y.shape[1]
    return (x.stride(0) % 4 == 0 and y.stride(0) % 4 == 0 and os.environ.get('GD_RGCN_NODE_MAJOR') != '1'
            and int(_lib.lib().gd_rgcn_tile_kl(d_in, d_out, n_blocks, int(trans))) > 0
            and rgcn_wave_form(d_in, d_out, n_blocks, tg.n, x.stride(0)) and tg.num_relations < 65536 and tg.fwd[3].numel() > 0)


def rgcn_typed_accumulate(tg, x, weight, n_blocks, trans, y, edge_w=None, relu_in=False):
    """y += sum_r (weighted mean over the relation-r in-edges of x) @ W_r (trans: the input gradient on the transposed
    graph with W_r^T) - raw, no autograd.  The (tile, relation) kernel where the widths allow, the node-major one
    otherwise or when GD_RGCN_NODE_MAJOR=1; edge_w (per edge of tg.fwd, in its order) replaces the mean weights;
    relu_in: the conv reads relu(x) - only the wave-private kernel forms it on the fly (rgcn_wave_relu_ok)."""
    d_in, d_out = x.shape[1], y.shape[1]
    arrays = tg.bwd if trans else tg.fwd
    if arrays[3].numel() == 0:
        return y
    tiled = (edge_w is None and x.stride(0) % 4 == 0 and y.stride(0) % 4 == 0 and os.environ.get('GD_RGCN_NODE_MAJOR') != '1'
             and int(_lib.lib().gd_rgcn_tile_kl(d_in, d_out, n_blocks, int(trans))) > 0)
    if tiled and rgcn_wave_form(d_in, d_out, n_blocks, tg.n, x.stride(0)) and tg.num_relations < 65536:      # (relation | scan steps << 16 per unit)
        # four diagonal blocks: the wave-private kernel (one wave per (64-node tile, block), csrc/rgcn_wave.hip)
        p = tg.wave_plan(bool(trans))
        packed = rgcn_packed_weight(weight, n_blocks, d_in, d_out, trans)
        check(_lib.lib().gd_rgcn_wave_conv_f32(ptr(p['job_tile']), p['n_tiles'], p['tile'], ptr(p['tile_unit_ptr']), p['n_units'], ptr(p['unit_rel']),
                                               ptr(p['unit_edges']), ptr(p['unit_row']), ptr(x), x.stride(0), d_in, ptr(packed),
                                               n_blocks, ptr(y), y.stride(0), d_out, tg.n, int(bool(relu_in)), stream_ptr(x.device)),
              'gd_rgcn_wave_conv_f32')
        return y
    if relu_in:
        raise _lib.GnnDeleteHipError('rgcn_typed_accumulate(relu_in=True) needs the wave-private kernel (check rgcn_wave_relu_ok)')
    if tiled:
        p = tg.tile_plan(bool(trans))
        packed = rgcn_packed_weight(weight, n_blocks, d_in, d_out, trans)
        y_ext = None
        if p['n_hubs']:
            bufs = p.setdefault('_y_ext', {})
            y_ext = bufs.get(d_out)
            if y_ext is None:
                y_ext = bufs[d_out] = torch.zeros(p['n_slice_rows'], d_out, dtype=torch.float32, device=x.device)
        check(_lib.lib().gd_rgcn_tile_conv_f32(ptr(p['tile_order']), ptr(p['tile_step_ptr']), ptr(p['step_rel']),
                                               ptr(p['step_piece_ptr']), ptr(p['piece']), ptr(p['col']),
                                               ptr(p['w']), p['n_tiles'], ptr(x), x.stride(0), d_in, ptr(packed), n_blocks,
                                               int(trans), ptr(y), y.stride(0), d_out, tg.n, ptr(p['hub_node']) if p['n_hubs'] else None,
                                               ptr(p['hub_ptr']) if p['n_hubs'] else None, p['n_hubs'], ptr(y_ext),
                                               stream_ptr(x.device)), 'gd_rgcn_tile_conv_f32')
        return y
    node_ptr, seg_ptr, seg_rel, col, w = arrays
    if edge_w is not None:
        w = edge_w
    weight = weight.detach().contiguous()
    check(_lib.lib().gd_rgcn_conv_f32(ptr(node_ptr), ptr(seg_ptr), ptr(seg_rel), ptr(col), ptr(w), ptr(x), x.stride(0), d_in,
                                      ptr(weight), n_blocks, int(trans), ptr(y), y.stride(0), d_out, tg.n,
                                      stream_ptr(x.device)), 'gd_rgcn_conv_f32')
    return y


class _RgcnConvFrozen(torch.autograd.Function):
    """y = sum_r mean_{j in N_r(i)} x_j W_r + x_i root + bias with CONSTANT relation weights (frozen
    backbone / evaluation): gd_rgcn_conv_f32 forms no [R, n, d] tensor; backward = input gradient only."""

    @staticmethod
    def forward(ctx, x, tg, weight, root, bias, n_blocks):
        x = _f32_rows(x)
        weight = weight.detach().contiguous()
        # root is [in, out]: the row kernel takes it as is (any widths up to 1024: MFMA where they allow)
        if x.shape[1] > 1024:
            raise NotImplementedError('RGCNConv: inputs above 1,024 floats have no kernel (no vendor-BLAS fallback)')
        y = rows_gemm(x, None, root.detach(), trans_w=False, bias=bias.detach() if bias is not None else None)
        rgcn_typed_accumulate(tg, x, weight, n_blocks, 0, y)
        ctx.tg, ctx.n_blocks = tg, n_blocks
        ctx.save_for_backward(weight, root.detach())
        return y

    @staticmethod
    def backward(ctx, dy):
        weight, root = ctx.saved_tensors
        dy = _f32_rows(dy)
        dx = rows_gemm(dy, None, root, trans_w=True)
        rgcn_typed_accumulate(ctx.tg, dy, weight, ctx.n_blocks, 1, dx)
        return dx, None, None, None, None, None


def rgat_aggregate_nograd(x, tg, alpha, weight, bias, n_blocks, out_dim):
    """y_i = sum_e alpha_e (x_j W_r) + bias for RGATConv under no_grad: the typed conv kernel with the attention
    coefficients as per-edge weights (alpha in input edge order)."""
    x = _f32_rows(x)
    n = x.shape[0]
    y = (bias.detach().to(torch.float32).expand(n, out_dim).contiguous() if bias is not None
         else torch.zeros(n, out_dim, dtype=torch.float32, device=x.device))
    node_ptr, seg_ptr, seg_rel, col, _ = tg.fwd
    if col.numel():
        w = alpha.detach().to(torch.float32)[tg.fwd_order].contiguous()
        weight = weight.detach().contiguous()
        check(_lib.lib().gd_rgcn_conv_f32(ptr(node_ptr), ptr(seg_ptr), ptr(seg_rel), ptr(col), ptr(w), ptr(x),
                                          x.stride(0), x.shape[1], ptr(weight), n_blocks, 0, ptr(y), y.stride(0),
                                          out_dim, n, stream_ptr(x.device)), 'gd_rgcn_conv_f32')
    return y


class _SegmentSoftmax(torch.autograd.Function):
    """alpha = softmax of e over the rows of a CSR (gd_segment_softmax_f32), differentiable in e."""

    @staticmethod
    def forward(ctx, e, rowptr):
        e = e.contiguous().float()
        alpha = torch.zeros_like(e)
        check(_lib.lib().gd_segment_softmax_f32(ptr(rowptr), ptr(e), int(rowptr.numel()) - 1, ptr(alpha), stream_ptr(e.device)),
              'gd_segment_softmax_f32')
        ctx.save_for_backward(alpha, rowptr)
        return alpha

    @staticmethod
    def backward(ctx, dalpha):
        alpha, rowptr = ctx.saved_tensors
        dalpha = dalpha.contiguous().float()
        de = torch.zeros_like(alpha)
        check(_lib.lib().gd_segment_softmax_bwd_f32(ptr(rowptr), ptr(alpha), ptr(dalpha), int(rowptr.numel()) - 1, ptr(de),
                                                    stream_ptr(alpha.device)), 'gd_segment_softmax_bwd_f32')
        return de, None


def segment_softmax(e, rowptr):
    """e [nnz] in CSR order, rowptr int32 [n + 1] -> alpha [nnz]."""
    return _SegmentSoftmax.apply(e, rowptr)


class _TypedWeightedSum(torch.autograd.Function):
    """m[(r, i), :] = sum over the type-r in-edges e of i of alpha_e x[src_e, :] - the weighted typed aggregation of
    RGATConv's messages (rgat.py:322-337) over a relation-major CSR (rows r * n + i), differentiable in x AND in the
    edge weights: dx through the transposed CSR with the same weights, d alpha_e = <dm[row(e)], x[src(e)]> (gd_rowpair_dot_f32)."""

    @staticmethod
    def forward(ctx, x, alpha_v, tc):
        x = _f32_rows(x)
        alpha_v = alpha_v.contiguous().float()
        m = _spmm_raw(tc['rowptr_v'], tc['col_v'], alpha_v, x, None, 0.0, tc['n_vrows'])
        ctx.tc = tc
        ctx.save_for_backward(x, alpha_v)
        return m

    @staticmethod
    def backward(ctx, dm):
        x, alpha_v = ctx.saved_tensors
        tc = ctx.tc
        dm = _f32_rows(dm)
        dx = dalpha = None
        if ctx.needs_input_grad[0]:
            dx = _spmm_raw(tc['rowptr_t'], tc['col_t'], alpha_v[tc['v_of_t']].contiguous(), dm, None, 0.0, tc['n'])
        if ctx.needs_input_grad[1]:
            dalpha = torch.empty_like(alpha_v)
            nnz = int(alpha_v.numel())
            check(_lib.lib().gd_rowpair_dot_f32(ptr(dm), dm.stride(0), ptr(tc['vrow_v']), ptr(x), x.stride(0), ptr(tc['col_v']), nnz,
                                                x.shape[1], ptr(dalpha), stream_ptr(x.device)), 'gd_rowpair_dot_f32')
        return dx, dalpha, None


def typed_weighted_sum(x, alpha_v, tc):
    return _TypedWeightedSum.apply(x, alpha_v, tc)


class _TypedConv(torch.autograd.Function):
    """y_i = sum over the in-edges e = (j -> i, r) of  c_e x_j W_r  with TRAINABLE relation weights: c_e = the mean weight
    1 / |N_r(i)| (RGCNConv, rgcn.py:17-38 when the conv trains: base.py:394-493, retrain.py:235-339) or a given per-edge
    coefficient in input edge order (RGATConv's attention, rgat.py:322-337).  Forward and input gradient = the typed conv
    kernels the frozen path runs; the weight gradient and the coefficients' gradient come straight from the edge lists
    (gd_typed_wgrad_f32, gd_typed_edge_dot_f32) - no [R, N, d] tensor of per-relation aggregates, no torch.einsum."""

    @staticmethod
    def forward(ctx, x, tg, weight, n_blocks, coef):
        x = _f32_rows(x)
        w = weight.detach().contiguous().float()
        n, d_out = x.shape[0], int(w.shape[-1]) * n_blocks
        y = torch.zeros(n, d_out, dtype=torch.float32, device=x.device)
        cf = None if coef is None else coef.detach().float().contiguous()
        rgcn_typed_accumulate(tg, x, w, n_blocks, 0, y, edge_w=None if cf is None else cf[tg.fwd_order].contiguous())
        ctx.tg, ctx.n_blocks, ctx.has_coef = tg, n_blocks, cf is not None
        ctx.save_for_backward(x, w, cf if cf is not None else x.new_zeros(0))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, cf = ctx.saved_tensors
        tg, nb = ctx.tg, ctx.n_blocks
        dy = _f32_rows(dy)
        d_in, d_out = x.shape[1], dy.shape[1]
        dx = dw = dcoef = None
        if ctx.needs_input_grad[0]:
            dx = torch.zeros_like(x)
            if ctx.has_coef and tg.bwd_order is None:
                raise _lib.GnnDeleteHipError('per-edge coefficients on a row partition of the typed graph are not supported')
            rgcn_typed_accumulate(tg, dy, w, nb, 1, dx, edge_w=cf[tg.bwd_order].contiguous() if ctx.has_coef else None)
        if ctx.needs_input_grad[2]:
            rm = tg.rel_major()
            wts = rm['w'] if not ctx.has_coef else cf[tg.fwd_order][rm['from_fwd']].contiguous()
            dw = torch.empty_like(w)
            check(_lib.lib().gd_typed_wgrad_f32(ptr(rm['rel_ptr']), tg.num_relations, ptr(rm['src']), ptr(rm['dst']), ptr(wts), ptr(x),
                                                x.stride(0), ptr(dy), dy.stride(0), nb, d_in, d_out, ptr(dw), stream_ptr(x.device)),
                  'gd_typed_wgrad_f32')
        if ctx.has_coef and ctx.needs_input_grad[4]:
            src, dst, rel = tg.fwd_edges()
            g_fwd = torch.empty(src.numel(), dtype=torch.float32, device=x.device)
            check(_lib.lib().gd_typed_edge_dot_f32(ptr(src), ptr(dst), ptr(rel), int(src.numel()), ptr(x), x.stride(0), ptr(dy),
                                                   dy.stride(0), ptr(w), nb, d_in, d_out, ptr(g_fwd), stream_ptr(x.device)),
                  'gd_typed_edge_dot_f32')
            dcoef = torch.empty_like(g_fwd)
            dcoef[tg.fwd_order] = g_fwd
        return dx, None, dw, None, dcoef


def typed_conv(x, tg, weight, n_blocks, coef=None):
    return _TypedConv.apply(x, tg, weight, n_blocks, coef)


def rgcn_conv_frozen(x, tg, weight, root, bias, n_blocks):
    return _RgcnConvFrozen.apply(x, tg, weight, root, bias, n_blocks)


class _RowPairLoss(torch.autograd.Function):
    """Per-row cosine distance / KL(softmax(b) || softmax(a)) with the gradient w.r.t. a formed in the same pass
    (gd_rowpair_loss_f32); b is a constant (the reference compares against embeddings computed under no_grad)."""

    @staticmethod
    def forward(ctx, a, b, kind):
        a, b = _f32_rows(a), _f32_rows(b.detach())
        n, d = a.shape
        val = torch.empty(n, dtype=torch.float32, device=a.device)
        grad = torch.empty(n, d, dtype=torch.float32, device=a.device)
        check(_lib.lib().gd_rowpair_loss_f32(kind, ptr(a), a.stride(0), None, ptr(b), b.stride(0), None, n, d, ptr(val), ptr(grad),
                                             grad.stride(0), stream_ptr(a.device)), 'gd_rowpair_loss_f32')
        ctx.save_for_backward(grad)
        return val

    @staticmethod
    def backward(ctx, dval):
        (grad,) = ctx.saved_tensors
        return grad * dval[:, None], None, None


def rowpair_loss_ok(a, b):
    return a.is_cuda and a.dim() == 2 and a.shape == b.shape and 0 < a.shape[1] <= 1024 and not b.requires_grad and a.shape[0] > 0


def rowpair_loss(a, b, kind):
    """kind 'cosine': 1 - cos(a_r, b_r) per row; 'kld': KL(softmax(b_r) || softmax(a_r)) per row.  Differentiable in a."""
    return _RowPairLoss.apply(a, b, {'cosine': 0, 'kld': 1}[kind])


# ------------------------------------------------------------------------------ decoders
class _EdgeDot(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, rel, e0, e1, etype):
        z = _f32_rows(z)
        e0, e1 = e0.contiguous().long(), e1.contiguous().long()
        if rel is not None:
            rel, etype = _f32_rows(rel), etype.contiguous().long()
        out = torch.empty(e0.shape[0], dtype=torch.float32, device=z.device)
        check(_lib.lib().gd_edge_dot_f32(ptr(z), z.stride(0), z.shape[1], ptr(e0), ptr(e1), ptr(rel),
                                         rel.stride(0) if rel is not None else 0, ptr(etype), e0.shape[0], ptr(out),
                                         stream_ptr(z.device)), 'gd_edge_dot_f32')
        ctx.save_for_backward(z, rel, e0, e1, etype)
        return out

    @staticmethod
    def backward(ctx, dout):
        z, rel, e0, e1, etype = ctx.saved_tensors
        dz = drel = None
        dout = dout.contiguous().float()
        if ctx.needs_input_grad[0]:
            n, d = z.shape
            if d % 4 == 0 and z.stride(0) % 4 == 0 and (rel is None or rel.stride(0) % 4 == 0):
                # node-major incidence list of the decoded edges (one stable sort) carrying, per incidence, the other
                # endpoint, the edge's upstream gradient and its relation; then one deterministic kernel
                m = e0.shape[0]
                ends = torch.cat([e0, e1])
                ends_sorted, order = torch.sort(ends, stable=True)
                edge = order % m
                other = torch.where(order >= m, e0[edge], e1[edge]).to(torch.int32)
                inc_ptr = torch.searchsorted(ends_sorted, torch.arange(n + 1, device=z.device))     # (a histogram of 8 M keys costs 0.4 ms)
                dz = torch.empty(n, d, dtype=torch.float32, device=z.device)
                et_inc = etype[edge].to(torch.int32) if rel is not None else None
                check(_lib.lib().gd_edge_dot_bwd_f32(ptr(z), z.stride(0), d, ptr(other), ptr(dout[edge]), ptr(rel),
                                                     rel.stride(0) if rel is not None else 0, ptr(et_inc),
                                                     ptr(inc_ptr), n, ptr(dz), dz.stride(0), stream_ptr(z.device)),
                      'gd_edge_dot_bwd_f32')
            else:
                a, b = z[e0], z[e1]
                r = rel[etype] if rel is not None else None
                g = dout[:, None]
                dz = torch.zeros_like(z)
                dz.index_add_(0, e0, g * (b * r if r is not None else b))
                dz.index_add_(0, e1, g * (a * r if r is not None else a))
        if rel is not None and ctx.needs_input_grad[1]:
            drel = torch.zeros_like(rel).index_add_(0, etype, dout[:, None] * z[e0] * z[e1])
        return dz, drel, None, None, None


def edge_dot(z, e0, e1, rel=None, etype=None):
    return _EdgeDot.apply(z, rel, e0, e1, etype)


# ------------------------------------------------------------------------------ edge-probability NI term
class _PairsSigmoidMse(torch.autograd.Function):
    """mean over the included pairs i > j of (sigmoid(z[nodes[i]] . z[nodes[j]]) - target[i, j])^2,
    value and gradient from one fused pass (gd_pairs_sigmoid_mse_f32); nothing |S| x |S| is written."""

    @staticmethod
    def forward(ctx, z, nodes, target, count):
        z = _f32_rows(z)
        n_s, d = int(nodes.shape[0]), z.shape[1]
        assert nodes.dtype == torch.int32 and nodes.is_contiguous()
        assert target.dtype == torch.float32 and target.dim() == 2 and target.stride(1) == 1
        assert target.shape[0] >= n_s and target.shape[1] >= n_s
        loss = torch.zeros((), dtype=torch.float32, device=z.device)
        dz = torch.empty(max(1, n_s), d, dtype=torch.float32, device=z.device)
        if count > 0 and n_s > 0:
            ws = torch.empty(_lib.lib().gd_pairs_sigmoid_mse_workspace(n_s, d), dtype=torch.float32, device=z.device)
            check(_lib.lib().gd_pairs_sigmoid_mse_f32(ptr(z), z.stride(0), ptr(nodes), n_s, d, ptr(target),
                                                      target.stride(0), 1.0 / count, ptr(loss), ptr(dz), ptr(ws),
                                                      stream_ptr(z.device)), 'gd_pairs_sigmoid_mse_f32')
        else:
            dz.zero_()
        ctx.save_for_backward(dz, nodes)
        ctx.shape = z.shape
        return loss

    @staticmethod
    def backward(ctx, dloss):
        dz, nodes = ctx.saved_tensors
        out = torch.zeros(ctx.shape, dtype=torch.float32, device=dz.device)
        if nodes.numel():
            out.index_copy_(0, nodes.long(), dz[:nodes.numel()] * dloss)
        return out, None, None, None


def pairs_sigmoid_mse(z, nodes, target, count):
    """`target`: dense [|S|, |S|] float32, entry (i, j) with i > j = target probability of the pair
    (nodes[i], nodes[j]), negative = pair excluded; `count` = number of included pairs."""
    return _PairsSigmoidMse.apply(z, nodes, target, count)
