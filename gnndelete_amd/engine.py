"""Fused Del-operator training step (the reference's hot loop body,
framework/trainer/gnndelete_nodeemb.py:188-299) for MSE losses.

The reference runs this iteration as ~60 autograd-dispatched PyTorch/PyG ops with three
``.item()`` syncs.  Here the iteration is an explicit forward + hand-derived backward over the
HIP kernels - no autograd tape - laid out once over static buffers and captured into a hipGraph
(torch.cuda.CUDAGraph), so a step is a single graph launch:

  forward   p1 = conv1(x)                      frozen backbone, written straight into z1
            z1[S1] = p1[S1] @ W_D1             Del-1 in place, MFMA; the gathered p1[S1] is kept
            p2 = conv2(relu(z1)) -> z2         relu fused into the W2 GEMM (MFMA), SpMM
            z2[S2] = p2[S2] @ W_D2             Del-2 in place
  losses    DEC + NI of both layers            one fused value+gradient kernel per layer
  backward  dW_D2 = p2[S2]^T dz2[S2]
            dz2[S2] <- dz2[S2] @ W_D2^T ; dt2 = A^T dz2 ; dh[S1] = dt2[S1] @ W2
            dW_D1  = p1[S1]^T (dz1 [+ relu'(z1) * dh])[S1]
  update    Adam on W_D1 / W_D2 with the zero_grad placement of the chosen --loss_type
            (including the gradient carry-over of both_layerwise and the never-zeroed grads of
            both_all).

Per-step loss sums are appended to a device-side history without a host sync.
"""
import os

import torch

from . import _lib, ops
from ._lib import check, ptr, stream_ptr
from .graph import graph_for
from .nn import GATConv, GCNConv, GINConv, RGCNConv, SAGEConv

LOSS_TYPES = ('both_all', 'both_layerwise', 'only2_layerwise', 'only2_all', 'only1')


def _loss_coefficients(loss_type, alpha):
    """(coef_r, coef_l) multiplying the DEC / NI terms inside the differentiated loss."""
    if loss_type in ('both_all', 'both_layerwise', 'only2_layerwise'):
        return alpha, 1.0 - alpha
    return alpha, 1.0                      # only2_all / only1: loss_l + alpha * loss_r


class _LayerTerms:
    """DEC + NI terms of one layer, folded per touched row (targets are fixed for the run).

    Every term is (row of z, row of z_ori, weight, kind).  All terms of a row are merged into the
    mean target ``tm[u]``, the count ``cnt[u]`` and a constant, using
        sum_t |z - o_t|^2 = c |z - mean_t o_t|^2 + (sum_t |o_t|^2 - c |mean_t o_t|^2)
    (folded once in fp64).  If some row carries both kinds (never the case for the reference's
    masks, where NI rows exclude the Df endpoints) the general segmented kernel is used instead."""

    def __init__(self, pos_edge, neg_edge, ni_mask, z_ori, coef_r, coef_l, reduction, row_range=None, d_norm=None):
        """row_range=(lo, hi): keep only the terms whose z row lies in [lo, hi) (1-D row
        partition); the mean normalisers stay the GLOBAL term counts.  d_norm: the width the 'mean' reduction divides
        by when z_ori carries zero padding columns behind it (NodeembEngine's padded class dimension)."""
        device = z_ori.device
        d = z_ori.shape[1]
        dn = d if d_norm is None else int(d_norm)
        pos, neg = pos_edge.to(device).long(), neg_edge.to(device).long()
        ni_rows = ni_mask.to(device).nonzero().flatten()
        m = pos.shape[1]
        self.n_r = (2 * m * dn) if reduction == 'mean' else 1
        self.n_l = (ni_rows.numel() * dn) if reduction == 'mean' else 1
        self.count_r, self.count_l = 2 * m, int(ni_rows.numel())
        w_r, w_l = coef_r / max(self.n_r, 1), coef_l / max(self.n_l, 1)
        rows = torch.cat([pos[0], pos[1], ni_rows])
        tgt = torch.cat([neg[0], neg[1], ni_rows])
        kind = torch.cat([torch.zeros(2 * m, dtype=torch.int32, device=device),
                          torch.ones(ni_rows.numel(), dtype=torch.int32, device=device)])
        mixed = bool(ni_mask.to(device)[pos.flatten()].any()) if m else False
        if row_range is not None:
            mine = (rows >= row_range[0]) & (rows < row_range[1])
            rows, tgt, kind = rows[mine], tgt[mine], kind[mine]
        self.k_const = [0.0, 0.0]
        self.folded = not mixed
        if self.folded:
            uniq, inv, c = torch.unique(rows, return_inverse=True, return_counts=True)
            # per-row sums over the row's loss terms in a FIXED order (terms sorted by row, original order inside a row; one
            # sequential sum per row): index_add_ adds with atomics in arrival order, and the constants of the loss LOG below
            # (q - c |tbar|^2, a cancellation) then change in their last bits from one engine to the next - found in round 4
            # as a one-ulp difference between the logs of two engines that ran identical kernels
            order = torch.argsort(inv, stable=True)
            o64 = z_ori[tgt[order]].double()
            if uniq.numel():
                tbar = torch.segment_reduce(o64, 'sum', lengths=c, axis=0)
                tbar /= c[:, None].double()
                q = torch.segment_reduce((o64 * o64).sum(1), 'sum', lengths=c, axis=0)
            else:                                   # (a rank of a row partition without loss terms)
                tbar = torch.zeros(0, d, dtype=torch.float64, device=device)
                q = torch.zeros(0, dtype=torch.float64, device=device)
            k_row = (q - c.double() * (tbar * tbar).sum(1)).clamp_(min=0.0)
            kind_u = torch.zeros(uniq.numel(), dtype=torch.int32, device=device)
            kind_u[inv] = kind
            self.k_const = [float(k_row[kind_u == 0].sum()), float(k_row[kind_u == 1].sum())]
            self.n_rows = int(uniq.numel())
            self.row_idx = uniq.to(torch.int32)
            self.tm = tbar.float().contiguous()
            self.cnt = c.float()
            self.coef = (2.0 * torch.where(kind_u == 0, w_r, w_l) * c.double()).float()
            self.kind = kind_u
            ws = _lib.lib().gd_rowtarget_mse_workspace(self.n_rows)
        else:
            w = torch.where(kind == 0, w_r, w_l).float()
            order = torch.argsort(rows, stable=True)
            rows, tgt, kind, w = rows[order], tgt[order], kind[order], w[order]
            seg_row, counts = torch.unique_consecutive(rows, return_counts=True)
            seg_ptr = torch.zeros(seg_row.numel() + 1, dtype=torch.int32, device=device)
            seg_ptr[1:] = torch.cumsum(counts, 0)
            self.n_seg = int(seg_row.numel())
            self.seg_ptr, self.seg_row = seg_ptr, seg_row.to(torch.int32)
            self.term_o, self.term_w, self.term_kind = tgt.to(torch.int32), w, kind
            self.z_ori = z_ori
            ws = _lib.lib().gd_rowpair_mse_workspace(self.n_seg)
        self.partials = torch.zeros(max(2, ws), dtype=torch.float32, device=device)

    def n_partial_blocks(self):
        """Blocks of per-block loss partials the folded kernel writes (for gd_loss_finalize_f32)."""
        return _lib.lib().gd_rowtarget_mse_blocks(self.n_rows) if self.folded and self.n_rows else 0

    def order_inside_first(self, idx, n_sel):
        """Reorder the folded loss rows: members of the sorted Del row list idx first (in their order), the others behind them.
        Sets n_in (their number) and returns it; the suffix [n_in:] is then what launch_outside() walks."""
        assert self.folded
        self.n_in = self.n_rows
        if self.n_rows == 0 or n_sel == 0:
            self.n_in = 0 if n_sel == 0 else self.n_rows
            return self.n_in
        pos = torch.searchsorted(idx, self.row_idx)
        inside = (pos < n_sel) & (idx[pos.clamp(max=n_sel - 1)] == self.row_idx)
        order = torch.argsort((~inside).to(torch.int8), stable=True)
        self.row_idx, self.cnt = self.row_idx[order].contiguous(), self.cnt[order].contiguous()
        self.coef, self.kind = self.coef[order].contiguous(), self.kind[order].contiguous()
        self.tm = self.tm[order].contiguous()
        self.n_in = int(inside.sum())
        return self.n_in

    def outside_blocks(self):
        n_out = self.n_rows - getattr(self, 'n_in', self.n_rows)
        return _lib.lib().gd_rowtarget_mse_blocks(n_out) if n_out else 0

    def launch_outside(self, z, dz, partials):
        """The loss rows outside the Del rows (after order_inside_first): z there is the conv output itself.  dz = None: the
        loss sums only.  partials: 2 floats per block, outside_blocks() of them."""
        k, n_out = self.n_in, self.n_rows - self.n_in
        if n_out == 0:
            return
        d = z.shape[1]
        check(_lib.lib().gd_rowtarget_mse_f32(ptr(z), z.stride(0), ptr(self.tm[k:]), d, ptr(self.row_idx[k:]), ptr(self.coef[k:]),
                                              ptr(self.cnt[k:]), ptr(self.kind[k:]), n_out, ptr(dz) if dz is not None else None,
                                              dz.stride(0) if dz is not None else 0, None, ptr(partials), stream_ptr(z.device)),
              'gd_rowtarget_mse_f32')

    def outside_args(self, z, dz, partials):
        """The argument run of one job of gd_rowtarget_mse_pair_f32 for the rows launch_outside() walks."""
        k, n_out = self.n_in, self.n_rows - self.n_in
        return (ptr(z), z.stride(0), ptr(self.tm[k:]), z.shape[1], ptr(self.row_idx[k:]), ptr(self.coef[k:]), ptr(self.cnt[k:]),
                ptr(self.kind[k:]), n_out, ptr(dz) if dz is not None else None, dz.stride(0) if dz is not None else 0, ptr(partials))

    def launch(self, z, dz, sums):
        """sums = None (folded form only): leave the partials for gd_loss_finalize_f32."""
        d = z.shape[1]
        if self.folded:
            check(_lib.lib().gd_rowtarget_mse_f32(ptr(z), z.stride(0), ptr(self.tm), d, ptr(self.row_idx),
                                                  ptr(self.coef), ptr(self.cnt), ptr(self.kind), self.n_rows, ptr(dz),
                                                  dz.stride(0), ptr(sums), ptr(self.partials), stream_ptr(z.device)),
                  'gd_rowtarget_mse_f32')
        else:
            zo = self.z_ori
            check(_lib.lib().gd_rowpair_mse_f32(ptr(z), z.stride(0), ptr(zo), zo.stride(0), d, ptr(self.seg_ptr),
                                                ptr(self.seg_row), self.n_seg, ptr(self.term_o), ptr(self.term_w),
                                                ptr(self.term_kind), ptr(dz), dz.stride(0), 0, ptr(sums),
                                                ptr(self.partials), stream_ptr(z.device)), 'gd_rowpair_mse_f32')


class _Adam:
    """torch.optim.Adam state for one Del weight.  With a shared iteration counter (`iter_ctr`, a
    device int32 the step's finalize kernel advances) the step number is read from it and the
    update can be fused into the weight-gradient reduction; without one it keeps its own counter."""

    def __init__(self, param, lr, betas=(0.9, 0.999), eps=1e-8, iter_ctr=None):
        self.param, self.lr, self.betas, self.eps = param, lr, betas, eps
        self.m = torch.zeros_like(param)
        self.v = torch.zeros_like(param)
        self.iter_ctr = iter_ctr
        self.step = torch.zeros(1, dtype=torch.int32, device=param.device) if iter_ctr is None else iter_ctr
        self.applied = 0                  # host-side count of updates (for state export)

    def apply(self, grad):
        self.applied += 1
        if self.iter_ctr is not None:
            check(_lib.lib().gd_adam_at_f32(ptr(self.param), ptr(grad), ptr(self.m), ptr(self.v), ptr(self.iter_ctr),
                                            self.param.numel(), self.lr, self.betas[0], self.betas[1], self.eps,
                                            stream_ptr(self.param.device)), 'gd_adam_at_f32')
            return
        check(_lib.lib().gd_adam_f32(ptr(self.param), ptr(grad), ptr(self.m), ptr(self.v), ptr(self.step),
                                     self.param.numel(), self.lr, self.betas[0], self.betas[1], self.eps,
                                     stream_ptr(self.param.device)), 'gd_adam_f32')

    def state_dict(self):
        return {'exp_avg': self.m.clone(), 'exp_avg_sq': self.v.clone(), 'step': int(self.step)}


def _rows_inside(terms, idx, n_sel):
    """Are all loss rows of `terms` (folded form) members of the sorted Del row list idx?"""
    if not terms.folded or n_sel == 0:
        return False
    if terms.n_rows == 0:
        return True
    pos = torch.searchsorted(idx, terms.row_idx)
    return bool(((pos < n_sel) & (idx[pos.clamp(max=n_sel - 1)] == terms.row_idx)).all())


def _loss_slots(terms, idx, n_sel, device):
    """(slot[n_sel] int32: position of each Del row in the folded loss rows or -1, cnt with NI negative)."""
    slot = torch.full((n_sel,), -1, dtype=torch.int32, device=device)
    if terms.n_rows:
        pos = torch.searchsorted(idx, terms.row_idx)
        inside = (pos < n_sel) & (idx[pos.clamp(max=n_sel - 1)] == terms.row_idx)      # (loss rows outside the Del rows: no slot)
        slot[pos[inside]] = torch.arange(terms.n_rows, dtype=torch.int32, device=device)[inside]
    return slot, torch.where(terms.kind == 1, -terms.cnt, terms.cnt).contiguous()


def _padded_out_shadow(model, o_pad):
    """The model with layer 2 widened to o_pad output columns by ZERO columns (conv2's weights, bias and attention vectors, W_D2):
    a node-classification request has out_dim = #classes (4 on the DBLP / collab stand-ins, delete_node.py:63-64), a width none of
    the matrix-core / weight-stationary / fused forms is built for.  The padding columns of every layer-2 quantity are exactly zero
    in every iteration - t2 = relu(z1) W2^T has zero columns where W2 has zero rows, the aggregation and the bias keep them zero,
    W_D2's padding rows / columns start at zero and their gradient p2^T dz2 is zero (p2 = 0 there, dz2 = coef (z2 - 0) = 0), and
    Adam on an all-zero gradient history moves nothing - so the top-left block follows the unpadded trajectory with the same
    products added in the same order plus exact zeros.  conv1 / deletion1 are the caller's own modules."""
    from types import SimpleNamespace
    import torch.nn as nn
    c2 = model.conv2
    wd2 = model.deletion2.deletion_weight
    o = wd2.shape[0]
    dev = wd2.device

    def rows(w):                                   # [o, in] -> [o_pad, in]
        out = torch.zeros(o_pad, *w.shape[1:], dtype=w.dtype, device=w.device)
        out[:o] = w.detach()
        return nn.Parameter(out, requires_grad=False)

    def last(v):                                   # [..., o] -> [..., o_pad]
        out = torch.zeros(*v.shape[:-1], o_pad, dtype=v.dtype, device=v.device)
        out[..., :o] = v.detach()
        return nn.Parameter(out, requires_grad=False)
    if isinstance(c2, GCNConv):
        s2 = GCNConv(c2.in_channels, o_pad)
        s2.lin.weight, s2.bias = rows(c2.lin.weight), last(c2.bias)
    elif isinstance(c2, GATConv):
        s2 = GATConv(c2.in_channels, o_pad, c2.negative_slope)
        s2.lin_src.weight, s2.bias = rows(c2.lin_src.weight), last(c2.bias)
        s2.att_src, s2.att_dst = last(c2.att_src), last(c2.att_dst)
    elif isinstance(c2, GINConv):
        lin = nn.Linear(c2.nn.in_features, o_pad, bias=c2.nn.bias is not None)
        lin.weight = rows(c2.nn.weight)
        if c2.nn.bias is not None:
            lin.bias = last(c2.nn.bias)
        s2 = GINConv(lin, c2.eps)
    elif isinstance(c2, SAGEConv):
        s2 = SAGEConv(c2.in_channels, o_pad)
        s2.lin_l.weight, s2.lin_l.bias, s2.lin_r.weight = rows(c2.lin_l.weight), last(c2.lin_l.bias), rows(c2.lin_r.weight)
    else:
        raise NotImplementedError(type(c2).__name__)
    w = torch.zeros(o_pad, o_pad, dtype=wd2.dtype, device=dev)
    w[:o, :o] = wd2.detach()
    del2 = SimpleNamespace(deletion_weight=nn.Parameter(w, requires_grad=False), mask=model.deletion2.mask)
    return SimpleNamespace(conv1=model.conv1, conv2=s2.to(dev), deletion1=model.deletion1, deletion2=del2)


class NodeembEngine:
    """One object per unlearning request (fixed graph, fixed Df, fixed negatives)."""

    def __init__(self, model, x, edge_index, z1_ori, z2_ori, pos_edge, neg_edge, ni_mask1, ni_mask2,
                 loss_type='both_layerwise', alpha=0.5, lr=1e-3, reduction='mean', mask_1hop=None, mask_2hop=None,
                 use_graph=True, history=4096, reorder=True, cache_layer1=False, affected_rows_only=False,
                 edge_type=None):
        """edge_type (R-GCN only): relation type per column of edge_index (reverse edges included, as
        delete_gnn.py:158-171 builds them); x is then the entity id vector the embedding table is indexed with."""
        assert loss_type in LOSS_TYPES, loss_type
        conv1, conv2 = model.conv1, model.conv2
        if not isinstance(conv2, (GCNConv, GINConv, GATConv, SAGEConv, RGCNConv)):
            raise NotImplementedError(f'NodeembEngine: unsupported conv {type(conv2).__name__}')
        if isinstance(conv2, RGCNConv):
            assert edge_type is not None, 'R-GCN needs edge_type'
            nb = conv2.num_blocks or 1
            if any((c.in_channels // (c.num_blocks or 1)) % 2 or (c.out_channels // (c.num_blocks or 1)) % 2
                   or c.in_channels > 128 or c.out_channels > 128 for c in (conv1, conv2)):
                raise NotImplementedError('NodeembEngine: R-GCN widths outside the typed conv kernel (<= 128, even blocks)')
            # Locality order (GD_RGCN_REORDER=0 turns it off): the degree-weighted label propagation of reorder.py cuts the typed
            # conv's fabric reads by ~10 % (FETCH_SIZE 4.41 -> 3.95 GB per layer-1 launch, profiles/r03_rgcn_reorder_ab.txt).
            # With the tile kernel that bought nothing (bound by its per-step dependent chain); the wave-private kernel runs
            # at the fabric rate, where bytes are time: 2.01 -> 1.89 ms per step on the synth-biokg request (round 4).
            # (cache_layer1: as for the other backbones - conv1's output is loop-invariant; affected_rows_only: conv2's INPUT
            #  GRADIENT is formed for the Del-1 rows only, the only rows that read it - see _conv2_backward_to_s1; the forward
            #  stays whole: the DEC rows of a knowledge-graph request are the Df endpoints, most of the graph)
            self._rgcn_rows_only = bool(affected_rows_only)
            affected_rows_only = False
            reorder = os.environ.get('GD_RGCN_REORDER', '1') != '0'
            x = model.node_emb.weight.detach()[x.to(model.node_emb.weight.device)]      # frozen embedding lookup, once
        dev = x.device
        if dev.type != 'cuda':
            raise _lib.GnnDeleteHipError('NodeembEngine needs CUDA(HIP) tensors (no CPU fallback)')
        # (round 6) A class dimension below 32 (node classification: out_dim = #classes) is padded with zero columns to the
        # width the fused layer-2 forms are built for (GD_PAD_OUT = 32 | 64, 0 = off): same trajectory (see _padded_out_shadow),
        # W_D2's top-left block is copied back into the caller's parameter at the end of every iteration.
        self._o_true = int(model.deletion2.deletion_weight.shape[0])
        self._user_wd2 = None
        pad_to = int(os.environ.get('GD_PAD_OUT', '64'))
        if (not isinstance(conv2, RGCNConv) and self._o_true < 32 and pad_to in (32, 64)
                and model.deletion1.deletion_weight.shape[0] == 128):
            self._user_wd2 = model.deletion2.deletion_weight
            model = _padded_out_shadow(model, pad_to)
            conv2 = model.conv2
            z2_ori = torch.nn.functional.pad(z2_ori.float(), (0, pad_to - self._o_true))
        self.model, self.loss_type, self.alpha = model, loss_type, alpha
        self.n = n = x.shape[0]
        none = torch.zeros(n, dtype=torch.bool)           # a model built without masks: Del = identity
        m1 = model.deletion1.mask if mask_1hop is None else mask_1hop
        m2 = model.deletion2.mask if mask_2hop is None else mask_2hop
        m1 = (none if m1 is None else m1).to(dev)
        m2 = (none if m2 is None else m2).to(dev)
        ni_mask1, ni_mask2 = ni_mask1.to(dev), ni_mask2.to(dev)
        pos_edge, neg_edge = pos_edge.to(dev), neg_edge.to(dev)
        self.perm = None
        if reorder and edge_index.shape[1] > 0 and n > 4096:
            # internal locality-preserving numbering (see reorder.py); everything below lives in it
            from .reorder import locality_order
            perm, inv = locality_order(edge_index, n)
            self.perm = perm
            x = x[perm].contiguous()
            edge_index = inv[edge_index]
            z1_ori, z2_ori = z1_ori[perm], z2_ori[perm]
            m1, m2, ni_mask1, ni_mask2 = m1[perm], m2[perm], ni_mask1[perm], ni_mask2[perm]
            pos_edge, neg_edge = inv[pos_edge], inv[neg_edge]
        self.x, self.edge_index = x, edge_index
        self.wd1, self.wd2 = model.deletion1.deletion_weight, model.deletion2.deletion_weight
        self.h, self.o = self.wd1.shape[0], self.wd2.shape[0]
        self.idx1 = m1.nonzero().flatten().to(device=dev, dtype=torch.int32)
        self.idx2 = m2.nonzero().flatten().to(device=dev, dtype=torch.int32)
        self.s1, self.s2 = int(self.idx1.numel()), int(self.idx2.numel())
        self.z1_ori, self.z2_ori = ops._f32_rows(z1_ori), ops._f32_rows(z2_ori)
        coef_r, coef_l = _loss_coefficients(loss_type, alpha)
        self.t1 = _LayerTerms(pos_edge, neg_edge, ni_mask1, self.z1_ori, coef_r, coef_l, reduction)
        self.t2 = _LayerTerms(pos_edge, neg_edge, ni_mask2, self.z2_ori, coef_r, coef_l, reduction, d_norm=self._o_true)
        self.uses_l1 = loss_type in ('both_all', 'both_layerwise', 'only1')
        self.uses_l2 = loss_type != 'only1'
        # does a gradient of loss-2 w.r.t. W_D1 (through conv2) ever reach an optimizer step?
        self.needs_l2_to_w1 = loss_type in ('both_all', 'both_layerwise', 'only2_all')

        f32 = dict(dtype=torch.float32, device=dev)
        self.z1 = torch.empty(n, self.h, **f32)
        self.z2 = torch.empty(n, self.o, **f32)
        self.xs1 = torch.empty(max(1, self.s1), self.h, **f32)       # p1[S1] (input rows of Del-1)
        # Del-2 out of place: conv2 writes p2, Del-2 writes z2[S2] from p2[S2] and the weight gradient gathers its
        # operand from p2 - no [S2, O] copy of the Del input per step.  z2 rows outside S2 are then never
        # formed, which training only tolerates when every layer-2 loss row lies in S2 (always so for the
        # reference's masks); otherwise the in-place form with its saved input is used.
        # (round 5) Knowledge-graph requests: upstream hands the NI masks to the Del operators (S minus the Df endpoints), so the
        # DEC rows lie OUTSIDE the Del rows.  The fused forms then take the loss rows inside the Del rows (the NI rows) and one
        # stand-alone loss launch per layer the rest, reading the conv output (Del is the identity there): layer 2 writes their
        # dz2 rows (= dp2, conv2's backward reads them), layer 1 the loss sums only (no trainable weight lies upstream of them).
        self._out1 = self._out2 = 0
        mixed_ok = (self.t1.folded and self.t2.folded and os.environ.get('GD_NO_SPLIT_LOSS') != '1' and os.environ.get('GD_NO_SPLIT') != '1')
        inside2 = _rows_inside(self.t2, self.idx2, self.s2)
        if (not inside2 and mixed_ok and self.s2 > 0 and self.o in (32, 64) and self.t2.n_rows > 0
                and os.environ.get('GD_NO_FUSED_L2') != '1'):
            self._out2 = self.t2.n_rows - self.t2.order_inside_first(self.idx2, self.s2)
            inside2 = True
        self._split2 = inside2 and os.environ.get('GD_NO_SPLIT') != '1'
        self.p2 = torch.empty(n, self.o, **f32) if self._split2 else self.z2
        self.xs2 = None if self._split2 else torch.empty(max(1, self.s2), self.o, **f32)
        # ... and with that, Del-2 forward + layer-2 loss + Del-2 input gradient are ONE kernel (csrc/del_fused.hip)
        self._fuse_l2 = (self._split2 and self.t1.folded and self.o in (32, 64) and self.t2.n_rows > 0
                         and os.environ.get('GD_NO_FUSED_L2') != '1')
        if self._fuse_l2:
            self._slot2, self._cnt_signed2 = _loss_slots(self.t2, self.idx2, self.s2, dev)
            self._lp2_blocks = _lib.lib().gd_del_loss_bwd_blocks(self.s2)
            self._lp2 = torch.zeros(2 * max(1, self._lp2_blocks + self.t2.outside_blocks()), **f32)
            self.dz2c = torch.zeros(self.s2, self.o, **f32)          # dz2 on the S2 rows (compact)
        # Same for layer 1: conv1 writes pre1, Del-1 writes z1[S1] from pre1[S1]; conv2's Linear reads row r from z1 if r is in
        # S1, else from pre1 (gd_rows_gemm_select_f32).  Upstream clones the whole [N, H] matrix for this (deletion.py:24).
        # (round 6) Also with the cached layer-1 output - pre1 is then that fixed buffer, filled once - so that the trainer's
        # default step runs the same fused / chained Del-1 pass as the full step instead of three launches (Del-1, the loss-fused
        # weight gradient, the gated product); GD_CACHE_SPLIT=0: the round-5 form (z1 = a copy of the cached output whose S1 rows
        # are overwritten every step).
        self._split1 = ((not cache_layer1 or os.environ.get('GD_CACHE_SPLIT', '1') != '0')
                        and _rows_inside(self.t1, self.idx1, self.s1) and os.environ.get('GD_NO_SPLIT') != '1')
        self.pre1 = torch.empty(n, self.h, **f32) if self._split1 else self.z1
        self._sel1 = m1.to(torch.uint8).contiguous() if self._split1 else None
        if self._split1:
            self.xs1 = None
        self.dz1 = torch.zeros(n, self.h, **f32)                     # only loss rows are ever written
        if isinstance(conv2, SAGEConv):
            # [ dt2 = A_mean^T dp2 | dp2 ]: the two halves of conv2's input gradient operand side by side, so
            # that dh = dt2 W_l + dp2 W_r is ONE K = 2*O product; dz2 is the right half (a strided view)
            self.dcat = torch.zeros(n, 2 * self.o, **f32)
            self.dz2 = self.dcat[:, self.o:]
        else:
            self.dz2 = torch.zeros(n, self.o, **f32)
        self.dh = torch.zeros(n, self.h, **f32)                      # only S1 rows are ever written
        self.z1_pos = torch.zeros(max(1, self.s1), (self.h + 31) // 32, dtype=torch.int32, device=dev)   # [z1[S1] > 0]
        self.g1 = torch.zeros_like(self.wd1)                         # the .grad of W_D1 / W_D2
        self.g2 = torch.zeros_like(self.wd2)
        self.sums = torch.zeros(4, **f32)                            # r1, l1, r2, l2 (sums of squares)
        self.ws1 = torch.empty(max(1, _lib.lib().gd_rows_gemm_wgrad_workspace(self.s1, self.h, self.h)), **f32)
        self.ws2 = torch.empty(max(1, _lib.lib().gd_rows_gemm_wgrad_workspace(self.s2, self.o, self.o)), **f32)
        self.iter_ctr = torch.zeros(1, dtype=torch.int32, device=dev)       # advanced once per step
        self.adam1, self.adam2 = _Adam(self.wd1, lr, iter_ctr=self.iter_ctr), _Adam(self.wd2, lr, iter_ctr=self.iter_ctr)
        self.hist = torch.zeros(history, 4, **f32)
        self.hist_pos = torch.zeros(1, dtype=torch.int32, device=dev)
        self.steps_done = 0
        self._graph = None
        self._use_graph = use_graph
        self._const_refs = {}
        # OPT-IN (GD_SIDE_STREAM=1), measured and NOT kept as the default: the split-K reductions (+ Adam) and the loss
        # finalize are launch-sized and nothing later in the iteration reads what they write, so they can run on a side
        # stream, forked behind the kernel that feeds them and joined at the end of the iteration.  On this part the
        # cross-queue dependencies of the replayed hipGraph cost more than the three links they remove from the chain:
        # 683 -> 715 us per step (profiles/r03_ab.txt).
        self._overlap = os.environ.get('GD_SIDE_STREAM') == '1'
        self._side = torch.cuda.Stream(device=dev) if self._overlap else None
        # layer-1 loss folded into the W_D1 weight-gradient fetch (see _wgrad1): needs the folded loss form on
        # both layers, MFMA-able widths, a loss type whose layer-1 gradient feeds W_D1, every loss row inside S1
        self._fuse_loss1 = False
        t1 = self.t1
        if (t1.folded and self.t2.folded and loss_type in ('both_layerwise', 'both_all', 'only1') and self.s1 > 0
                and self.h in (32, 64, 128) and t1.n_rows > 0 and os.environ.get('GD_NO_FUSED_LOSS1') != '1'):
            pos = torch.searchsorted(self.idx1, t1.row_idx)
            inside = (pos < self.s1) & (self.idx1[pos.clamp(max=self.s1 - 1)] == t1.row_idx)
            if not bool(inside.all()) and mixed_ok and not self._split1:
                self._out1 = t1.n_rows - t1.order_inside_first(self.idx1, self.s1)
            if bool(inside.all()) or self._out1:
                self._slot1, self._cnt_signed1 = _loss_slots(t1, self.idx1, self.s1, dev)
                self._lp1_blocks = _lib.lib().gd_rows_gemm_wgrad_blocks(self.s1)
                self._lp1 = torch.zeros(2 * max(1, self._lp1_blocks + t1.outside_blocks()), **f32)
                self._fuse_loss1 = True
        # The launch-sized tail of the step (two split-K reductions with Adam + the loss finalize) as ONE launch
        # (gd_step_tail_f32) where both optimizers step in every iteration and both layers run their fused loss forms
        # (round 5: also where a layer runs its UNFUSED loss / weight-gradient stages - the knowledge-graph request, whose
        #  DEC rows lie outside the Del masks: the weight-gradient launches then leave their partial products, the loss kernels
        #  their per-block partials, and four launch-sized kernels - two reductions, Adam, finalize - become this one)
        self._tail = (t1.folded and self.t2.folded and self.s1 > 0 and self.s2 > 0
                      and loss_type in ('both_layerwise', 'both_all') and not self._overlap and self.h % 2 == 0 and self.o % 2 == 0
                      and os.environ.get('GD_NO_STEP_TAIL') != '1'
                      and ((self._fuse_loss1 and self._fuse_l2)
                           # (unfused stages: only where the weight-gradient launches are the split-K matrix kernels, whose partial
                           #  products the tail reduces - widths of 32 / 64 / 128)
                           or (self.h in (32, 64, 128) and self.o in (32, 64, 128) and os.environ.get('GD_TAIL_FUSED_ONLY') != '1')))
        self._tail_acc = [0, 0]
        self._arrive = torch.zeros(2, dtype=torch.int32, device=dev)     # step_tail's check-in counter (+ the word its address trick may name)
        # ... and with the tail launch doing the reduction, the W_D2 weight gradient's partial sums come out of the fused Del-2
        # kernel itself (gd_del_loss_bwd_wgrad_f32: p2 and dz2 are in its registers) - no weight-gradient launch, no dz2 buffer
        self._fuse_wg2 = self._tail and self._fuse_l2 and os.environ.get('GD_NO_FUSED_WGRAD2') != '1'
        if self._fuse_wg2:
            self._lp2 = torch.zeros(2 * max(1, _lib.lib().gd_rows_gemm_wgrad_blocks(self.s2) + self.t2.outside_blocks()), **f32)
            self._lp2_blocks = _lib.lib().gd_del_loss_bwd_wgrad_parts(self.s2, self.o)      # (one per compute unit in the step-sized form)
        # (round 5) ... and where the W_D1 step of an iteration only waits for gradients that exist when Del-1 runs (the
        # layer-wise types: this iteration's layer-1 loss + the gradient conv2 sent back in the PREVIOUS one), Del-1 itself, the
        # layer-1 loss and the W_D1 weight gradient are ONE pass over the S1 rows (gd_del1_loss_wgrad_f32: pre1 read once, z1
        # only written - the weight-gradient launch re-read both)
        self._fuse_del1 = bool(self._tail and self._fuse_loss1 and self._split1 and loss_type in ('both_layerwise', 'only1')
                               and _lib.lib().gd_del1_loss_wgrad_covers(self.s1, self.h))
        if self._fuse_del1:       # ... which leaves one partial matrix per compute unit (the tail launch is told how many)
            # (the buffer keeps the size of the two-launch form: bench.py's stand-alone timing of the weight-gradient kernel
            #  writes gd_rows_gemm_wgrad_blocks(s1) block sums into it)
            self._lp1_blocks = _lib.lib().gd_del1_loss_wgrad_parts(self.s1)
        # ... and for GCN / GIN the gradient conv2 sent back in the previous iteration is FORMED in that pass too, from the aggregated
        # layer-2 gradient the previous iteration left and the sign pattern its Del-1 stored (gd_del1_chain_loss_wgrad_f32): the
        # gated [S1, 64] x [64, 128] product at the end of the step, its [S1, 128] write and the read-back are gone
        self._chain1 = bool(self._fuse_del1 and loss_type == 'both_layerwise' and isinstance(conv2, (GCNConv, GINConv))
                            and self.h == 128 and self.o == 64 and os.environ.get('GD_DEL1_CHAIN') != '0')
        # (GATConv: the same with its two rank-1 terms added in front of the gate; the message gradient and the two logit gradients the
        #  previous iteration's backward left live in buffers this engine owns)
        self._gat_bufs = None
        if (self._fuse_del1 and loss_type == 'both_layerwise' and isinstance(conv2, GATConv) and self.h == 128 and self.o == 64
                and self._mfma_weight(conv2.lin_src.weight) and os.environ.get('GD_NO_GAT_RANK1_EPILOGUE') != '1'
                and os.environ.get('GD_DEL1_CHAIN') != '0'):
            self._chain1 = True
            self._gat_bufs = {'dh': torch.zeros(n, self.o, **f32), 'da_src': torch.zeros(n, **f32), 'da_dst': torch.zeros(n, **f32)}
        self._dt2_keep = torch.zeros(n, self.o, **f32) if (self._chain1 and self._gat_bufs is None) else None
        # OPT-IN (GD_SMALL_SIDE_W1=1), measured in round 6 and NOT the default: on SMALL requests (below the row count of the fused
        # Del-1 pass) the loss-fused W_D1 weight-gradient launch reads only what exists once Del-1 has run and nothing reads its
        # partial sums before the tail, and at a few thousand rows neither it nor layer 2's launches fill the chip - so it can run
        # on a side stream next to layer 2's forward, joined before the iteration overwrites dh.  synth-dblp 0.1834 -> 0.192 ms,
        # synth-cora 0.4765 -> 0.484 ms (3 alternating pairs, profiles/r06_side_w1_ab.txt): the cross-queue dependencies of the
        # replayed hipGraph cost more than the 17 us launch they hide - the same answer as GD_SIDE_STREAM gave at step size in round 3.
        self._side_w1 = bool(self._tail and self._fuse_loss1 and not self._fuse_del1 and loss_type == 'both_layerwise'
                             and self.s1 > 0 and not self._overlap and os.environ.get('GD_SMALL_SIDE_W1') == '1')
        if self._side_w1 and self._side is None:
            self._side = torch.cuda.Stream(device=dev)
        # ... and the two stand-alone loss launches of a knowledge-graph step (DEC rows outside the Del rows, both layers) are one
        self._out_pair = bool(self._out1 and self._out2 and self._fuse_loss1 and self._fuse_l2 and loss_type != 'only2_all'
                              and _lib.lib().gd_rowtarget_mse_pair_covers(self.h, self.o) and os.environ.get('GD_NO_LOSS_PAIR') != '1')
        self._mode = {GCNConv: 'gcn', GINConv: 'gin', GATConv: 'gat', SAGEConv: 'sage', RGCNConv: 'rgcn'}[type(conv2)]
        self._gat_dots = os.environ.get('GD_NO_GAT_DOTS') != '1'      # attention logits from the GEMM epilogue
        self._gat_r1 = None                                            # (att_src W2, att_dst W2): constants of the frozen conv2
        if self._mode == 'gat':
            w2_ = conv2.lin_src.weight.detach()
            self._gat_r1 = ((conv2.att_src.detach().reshape(1, -1) @ w2_).reshape(-1).float().contiguous(),
                            (conv2.att_dst.detach().reshape(1, -1) @ w2_).reshape(-1).float().contiguous())
        if self._mode == 'rgcn':
            from .graph import TypedNodeCSR
            self.typed = TypedNodeCSR(edge_index, edge_type.to(dev), n, conv2.num_relations)
            self.typed_s1 = None
            if getattr(self, '_rgcn_rows_only', False) and self.s1 > 0:
                in1 = torch.zeros(n, dtype=torch.bool, device=dev)
                in1[self.idx1.long()] = True
                self.typed_s1 = self.typed.restrict_bwd(in1)
            self.graph = None
            self._hbuf = torch.zeros(n, self.h, **f32)                # relu(z1) as the typed conv reads it
            self._dxbuf = torch.zeros(n, self.h, **f32)               # conv2's input gradient
            # tile plans and packed relation weights are made here, outside any graph capture
            for c_, din, dout, tr in ((conv1, self.x.shape[1], self.h, 0), (conv2, self.h, self.o, 0), (conv2, self.o, self.h, 1)):
                if int(_lib.lib().gd_rgcn_tile_kl(din, dout, c_.num_blocks or 1, tr)) > 0 and os.environ.get('GD_RGCN_NODE_MAJOR') != '1':
                    tg_ = self.typed_s1 if (tr and self.typed_s1 is not None) else self.typed
                    if ops.rgcn_wave_form(din, dout, c_.num_blocks or 1, n, din) and self.typed.num_relations < 65536:
                        tg_.wave_plan(bool(tr))
                    else:
                        tg_.tile_plan(bool(tr))
                    ops.rgcn_packed_weight(c_.weight.detach(), c_.num_blocks or 1, din, dout, tr)
        else:
            gmode = {'gcn': 'gcn', 'gin': 'sum', 'gat': 'gat', 'sage': 'mean'}[self._mode]
            self.graph = graph_for(edge_index, n, gmode)
        if self._mode == 'sage':
            # frozen: [W_l ; W_r] of conv2 stacked once -> one GEMM gives (t2_l | t2_r), and its transpose
            # side gives dh from (dt2 | dp2)
            self._w2cat = torch.cat([conv2.lin_l.weight.detach(), conv2.lin_r.weight.detach()], 0).contiguous()
        # Optional: the frozen layer-1 output p1 = conv1(x) is loop-invariant (fixed x, edges and
        # weights), so it can be computed once.  OFF by default: upstream recomputes it every epoch and
        # the benchmark's `value` is measured that way; the trainer turns it on (identical results).
        # Affected rows only (GCN, GIN, GraphSAGE; GAT: its aggregation kernels): the training graph holds the S_Df edges plus self loops, so a row outside the
        # 2-hop set S2 neither reads nor feeds a row inside it and no loss term sees it - its transforms and
        # aggregates influence nothing the iteration produces.  With this option every N-row kernel runs on the S2
        # rows (the transposed aggregation on S1): identical Del weights and losses, cost proportional to the
        # affected subgraph instead of the graph.  OFF by default (upstream computes every row and the benchmark's
        # `value` is measured that way); the trainer turns it on.  Verified closed under the graph first.
        self._rows_only = False
        gin_ok = self._mode == 'gin' and conv1.nn.out_features <= conv1.nn.in_features and conv2.nn.out_features <= conv2.nn.in_features
        if (affected_rows_only and (self._mode in ('gcn', 'gat', 'sage') or gin_ok) and self._split2 and self.s2 > 0
                and not self._out1 and not self._out2):        # (loss rows outside the Del rows read conv outputs of rows outside S2)
            g = self.graph
            in2 = torch.zeros(n, dtype=torch.bool, device=dev)
            in2[self.idx2.long()] = True
            in1_rows = torch.zeros(n, dtype=torch.bool, device=dev)
            in1_rows[self.idx1.long()] = True
            deg = (g.rowptr[1:] - g.rowptr[:-1]).long()
            erow = torch.repeat_interleave(torch.arange(n, device=dev), deg)
            closed = bool(in2[g.col.long()[in2[erow]]].all()) and bool(in2[self.idx1.long()].all())
            deg_t = (g.rowptr_t[1:] - g.rowptr_t[:-1]).long()
            erow_t = torch.repeat_interleave(torch.arange(n, device=dev), deg_t)
            closed = closed and bool(in2[g.col_t.long()[in1_rows[erow_t]]].all())
            if closed:
                from .graph import SplitPlan
                self._plan2 = SplitPlan(g.rowptr, rows=self.idx2)
                self._plan_t1 = SplitPlan(g.rowptr_t, rows=self.idx1)
                self._t1buf = torch.zeros(n, self.h, **f32)
                self._t1rbuf = torch.zeros(n, self.h, **f32) if self._mode == 'sage' else None
                self._t2buf = torch.zeros(n, self.o * (2 if self._mode == 'sage' else 1), **f32)
                self._dt2buf = torch.zeros(n, self.o, **f32)
                self._adots = torch.zeros(2, n, **f32)                     # GAT logits of a row subset
                if self._split1:
                    self.pre1.zero_()
                self.p2.zero_()
                self._rows_only = True
        if self._rows_only and self._chain1 and self._dt2_keep is not None:
            # (round 6) the rows-only step is chained as well: its transposed aggregation writes the S1 rows of the buffer the next
            # iteration's Del-1 pass reads (all it reads); before, `_chain1` stayed set while the step ran the unchained form
            self._dt2buf = self._dt2_keep
        self.cache_layer1 = cache_layer1
        if cache_layer1:
            with torch.no_grad():
                self._conv1_forward()                  # (writes pre1: its own buffer in the split form, else z1)
                if self._split1:
                    self.p1 = self.pre1
                else:
                    self.p1 = self.z1.clone()
                    if self.s1:
                        self.xs1.copy_(self.p1[self.idx1.long()])

    # ------------------------------------------------------------------ pieces
    @staticmethod
    def _mfma_weight(weight):
        out_f, in_f = weight.shape
        return in_f % 32 == 0 and out_f % 32 == 0 and out_f <= 128 and in_f * out_f * 4 <= 64 * 1024

    def _linear(self, x, weight, relu_in=False):
        """x @ weight^T on the matrix-core kernels: the whole-weight-in-LDS row kernel when the weight fits its 64 KB
        image, else the K-tiled kernel (wide bag-of-words inputs: the padded copy of x is made once)."""
        out_f, in_f = weight.shape
        if in_f % 32 == 0 and out_f % 32 == 0 and out_f <= 128 and in_f * out_f * 4 <= 64 * 1024:
            return ops.rows_gemm(x, None, weight, trans_w=True, const_w=True, relu_in=relu_in)
        if ops.mfma_out_width(out_f):
            xin = torch.relu(x) if relu_in else x
            return ops.gemm_wide(xin, ops._const_weight(weight, True)[0], const_x=not relu_in, const_w=True)
        if in_f <= 1024:
            return ops.rows_gemm(x, None, weight, trans_w=True, const_w=True, relu_in=relu_in)      # any widths: one wave per row
        raise NotImplementedError(f'NodeembEngine: a {in_f} -> {out_f} Linear has no kernel (inputs above 1,024 floats need an output '
                                  'width in {32, 64, 96, 128}); there is no vendor-BLAS fallback')

    def _linear_relu_z1(self, weight):
        """relu(z1) @ weight^T where z1 = Del-1 output on the S1 rows and conv1 output elsewhere."""
        if not self._split1:
            return self._linear(self.z1, weight, relu_in=True)
        out_f, in_f = weight.shape
        if in_f % 32 == 0 and out_f % 32 == 0 and out_f <= 128 and in_f * out_f * 4 <= 64 * 1024:
            return ops.rows_gemm_select(self.pre1, self.z1, self._sel1, weight, trans_w=True, const_w=True, relu_in=True)
        z = torch.where(self._sel1.bool()[:, None], self.z1, self.pre1)
        return self._linear(z, weight, relu_in=True)

    def _conv1_forward(self):
        """Frozen layer 1, recomputed every step exactly as upstream does, written into z1."""
        c = self.model.conv1
        g = self.graph
        if self._mode == 'rgcn':
            self._rgcn_conv(c, self.x, self.pre1, 0)
        elif self._mode == 'gcn' and self._rows_only and self._split1 and self._mfma_weight(c.lin.weight):
            ops.rows_gemm(self.x, self.idx2, c.lin.weight, trans_w=True, const_w=True, out=self._t1buf)
            self._spmm(False, g.val, self._t1buf, self.pre1, c.bias, 0.0, plan=self._plan2)
        elif self._mode == 'gcn':
            self._spmm(False, g.val, self._linear(self.x, c.lin.weight), self.pre1, c.bias, 0.0)
        elif self._mode == 'gin' and self._rows_only and self._split1 and self._mfma_weight(c.nn.weight):
            lin = c.nn
            ops.rows_gemm(self.x, self.idx2, lin.weight, trans_w=True, const_w=True, out=self._t1buf)
            self._spmm(False, None, self._t1buf, self.pre1, lin.bias, 1.0 + c.eps, plan=self._plan2)
        elif self._mode == 'gin':
            lin = c.nn
            if lin.out_features <= lin.in_features:
                self._spmm(False, None, self._linear(self.x, lin.weight), self.pre1, lin.bias, 1.0 + c.eps)
            else:
                agg = torch.empty_like(self.x)
                self._spmm(False, None, self.x, agg, None, 1.0 + c.eps)
                ops.rows_gemm(agg, None, lin.weight, trans_w=True, const_w=True, bias=lin.bias, out=self.pre1)
        elif self._mode == 'sage' and self._rows_only and self._split1 and self._mfma_weight(c.lin_l.weight):
            ops.rows_gemm(self.x, self.idx2, c.lin_l.weight, trans_w=True, const_w=True, out=self._t1buf)
            ops.rows_gemm(self.x, self.idx2, c.lin_r.weight, trans_w=True, const_w=True, out=self._t1rbuf)
            self._spmm(False, g.val, self._t1buf, self.pre1, c.lin_l.bias, 1.0, x_self=self._t1rbuf, plan=self._plan2)
        elif self._mode == 'sage':
            # out_i = mean_j (x_j W_l^T) + b_l + x_i W_r^T  (transform first, then aggregate at width H)
            t_l = self._linear(self.x, c.lin_l.weight)
            if (self._mfma_weight(c.lin_r.weight) and ops.rows_gemm_accumulate_ok(self.n, c.lin_r.weight.shape[1], c.lin_r.weight.shape[0])
                    and os.environ.get('GD_SAGE_ROOT_IN_SPMM') != '1'):
                # (round 6) the root term is ADDED by its own product (gd_rows_gemm_accumulate_f32: the accumulators of a unit start
                # from the aggregated rows) instead of travelling through the aggregation as a second row stream: the latency-bound
                # sweep reads one row per edge less per target, the matrix-bound product has the memory to spare
                self._spmm(False, g.val, t_l, self.pre1, c.lin_l.bias, 0.0)
                ops.rows_gemm_accumulate_(self.pre1, self.x, None, c.lin_r.weight, trans_w=True, const_w=True)
            else:
                t_r = self._linear(self.x, c.lin_r.weight)
                self._spmm(False, g.val, t_l, self.pre1, c.lin_l.bias, 1.0, x_self=t_r)
        else:
            wsrc = c.lin_src.weight
            if self._rows_only and self._split1 and self._mfma_weight(wsrc):
                h1 = ops.rows_gemm(self.x, self.idx2, wsrc, trans_w=True, const_w=True, out=self._t1buf)       # rows outside stay 0
                a_src, a_dst = ops.row_dots(h1, c.att_src, c.att_dst)
            elif self._gat_dots and ops.rows_gemm_dots_ok(wsrc.shape[1], wsrc.shape[0], self.n):
                h1, a_src, a_dst = ops.rows_gemm_dots(self.x, wsrc, c.att_src, c.att_dst, const_w=True)   # logits from the epilogue
            else:
                h1 = self._linear(self.x, wsrc)
                a_src, a_dst = ops.row_dots(h1, c.att_src, c.att_dst)
            ops.gat_forward_raw(g, h1, a_src, a_dst, c.bias, c.negative_slope, out=self.pre1,
                                plan=self._plan2 if (self._rows_only and self._split1) else None)

    def _rgcn_conv(self, conv, inp, out, trans, relu_in=False):
        """out = inp @ root (+ bias) + sum_r mean_{N_r} inp W_r (forward), or the input gradient with trans = 1:
        out = inp @ root^T + the typed kernel on the transposed graph with W_r^T.  Raw kernel calls, no tape.
        relu_in: both products read relu(inp), formed in their operand paths (no relu'd copy of inp)."""
        tg = self.typed
        nb = conv.num_blocks or 1
        ops.rows_gemm(inp, None, conv.root.detach(), trans_w=bool(trans), bias=None if trans else conv.bias.detach(), out=out,
                      relu_in=relu_in)
        ops.rgcn_typed_accumulate(tg, inp, conv.weight.detach(), nb, int(trans), out, relu_in=relu_in)

    def _conv2_forward(self):
        c = self.model.conv2
        if self._mode == 'rgcn':
            if self._split1:
                torch.where(self._sel1.bool()[:, None], self.z1, self.pre1, out=self._hbuf)
                self._hbuf.clamp_(min=0)
                self._rgcn_conv(c, self._hbuf, self.p2, 0)
            elif ops.rgcn_wave_relu_ok(self.typed, self.z1, self.p2, c.num_blocks or 1) and os.environ.get('GD_RGCN_RELU_PASS') != '1':
                # relu(z1) is formed where the two products read z1 (the typed conv's gathered rows, the root product's
                # operand path): no clamp pass over [n, h], no torch op in the step
                self._rgcn_conv(c, self.z1, self.p2, 0, relu_in=True)
            else:
                torch.clamp(self.z1, min=0, out=self._hbuf)
                self._rgcn_conv(c, self._hbuf, self.p2, 0)
        elif self._mode == 'gcn' and self._rows_only:
            if self._split1:
                t2 = ops.rows_gemm_select(self.pre1, self.z1, self._sel1, c.lin.weight, trans_w=True, const_w=True, relu_in=True,
                                          out=self._t2buf, idx=self.idx2)
            else:       # layer 1 cached: z1 holds conv1's output with the Del'd rows written over it
                t2 = ops.rows_gemm(self.z1, self.idx2, c.lin.weight, trans_w=True, const_w=True, relu_in=True, out=self._t2buf)
            self._spmm(False, self.graph.val, t2, self.p2, c.bias, 0.0, plan=self._plan2)
        elif self._mode == 'gcn':
            t2 = self._linear_relu_z1(c.lin.weight)
            self._spmm(False, self.graph.val, t2, self.p2, c.bias, 0.0)
        elif self._mode == 'gin' and self._rows_only:
            lin = c.nn
            if self._split1:
                t2 = ops.rows_gemm_select(self.pre1, self.z1, self._sel1, lin.weight, trans_w=True, const_w=True, relu_in=True,
                                          out=self._t2buf, idx=self.idx2)
            else:
                t2 = ops.rows_gemm(self.z1, self.idx2, lin.weight, trans_w=True, const_w=True, relu_in=True, out=self._t2buf)
            self._spmm(False, None, t2, self.p2, lin.bias, 1.0 + c.eps, plan=self._plan2)
        elif self._mode == 'gin':
            lin = c.nn
            if lin.out_features <= lin.in_features:
                t2 = self._linear_relu_z1(lin.weight)
                self._spmm(False, None, t2, self.p2, lin.bias, 1.0 + c.eps)
            else:
                raise NotImplementedError('GIN layer that widens its input is not on the fused path')
        elif self._mode == 'sage' and self._rows_only:
            if self._split1:
                t2 = ops.rows_gemm_select(self.pre1, self.z1, self._sel1, self._w2cat, trans_w=True, const_w=True, relu_in=True,
                                          out=self._t2buf, idx=self.idx2)
            else:
                t2 = ops.rows_gemm(self.z1, self.idx2, self._w2cat, trans_w=True, const_w=True, relu_in=True, out=self._t2buf)
            self._spmm(False, self.graph.val, t2[:, :self.o], self.p2, c.lin_l.bias, 1.0, x_self=t2[:, self.o:],
                       plan=self._plan2)
        elif self._mode == 'sage':
            t2 = self._linear_relu_z1(self._w2cat)              # [N, 2*O] = (t2_l | t2_r)
            self._spmm(False, self.graph.val, t2[:, :self.o], self.p2, c.lin_l.bias, 1.0, x_self=t2[:, self.o:])
        else:   # gat
            wsrc = c.lin_src.weight
            if self._gat_dots and self._split1 and ops.rows_gemm_dots_ok(wsrc.shape[1], wsrc.shape[0], self.s2 if self._rows_only else self.n,
                                                                        selected=self._rows_only):
                ro = self._rows_only
                h2, self._a_src, self._a_dst = ops.rows_gemm_dots(
                    self.pre1, wsrc, c.att_src, c.att_dst, inp_alt=self.z1, sel=self._sel1, relu_in=True, const_w=True,
                    out=self._t2buf if ro else None, idx=self.idx2 if ro else None,
                    dots_out=(self._adots[0], self._adots[1]) if ro else None)
            else:
                h2 = self._linear_relu_z1(wsrc)
                self._a_src, self._a_dst = ops.row_dots(h2, c.att_src, c.att_dst)
            self._h2 = h2
            _, self._rowmax, self._rowsum = ops.gat_forward_raw(self.graph, h2, self._a_src, self._a_dst, c.bias,
                                                                c.negative_slope, out=self.p2,
                                                                plan=self._plan2 if self._rows_only else None)

    def _conv2_backward_to_s1(self):
        """dh[S1] = d loss2 / d z1 (post-Del, pre-ReLU) restricted to the S1 rows (all that Del-1 needs)."""
        c = self.model.conv2
        g = self.graph
        if self._mode == 'rgcn':
            if self.typed_s1 is not None:
                # affected rows only: the input gradient of conv2 on the Del-1 rows (root product on the index list, the typed
                # kernel on the out-edges of those rows); identical dh, a sixth of the gathered rows on the biokg request
                ops.rows_gemm(self.dz2, self.idx1, c.root.detach(), trans_w=True, out=self._dxbuf)
                ops.rgcn_typed_accumulate(self.typed_s1, self.dz2, c.weight.detach(), c.num_blocks or 1, 1, self._dxbuf)
            else:
                self._rgcn_conv(c, self.dz2, self._dxbuf, 1)
            # dh[S1] = dx[S1] * [z1[S1] > 0]  (ReLU backward from the packed sign bits of the Del-1 output; folding the gate into
            # the typed launch's epilogue removed the launch and not a microsecond: tools/experiments/patches/r05_rgcn_gate_fold_*)
            ops.gate_rows(self._dxbuf, self.idx1, self.z1_pos, self.dh)
            return
        if self._mode == 'sage':
            self._spmm(True, g.val_t, self.dz2, self.dcat[:, :self.o], None, 0.0,
                       plan=self._plan_t1 if self._rows_only else None)
            dt2, w2 = self.dcat, self._w2cat
        elif self._mode in ('gcn', 'gin'):
            chain = self._chain1
            dt2 = self._dt2_keep if chain else torch.empty(self.n, self.o, dtype=torch.float32, device=self.x.device)
            if self._mode == 'gcn' and self._rows_only:
                dt2 = self._dt2buf
                self._spmm(True, g.val_t, self.dz2, dt2, None, 0.0, plan=self._plan_t1)
                w2 = c.lin.weight
            elif self._mode == 'gcn':
                self._spmm(True, g.val_t, self.dz2, dt2, None, 0.0)
                w2 = c.lin.weight
            elif self._rows_only:
                dt2 = self._dt2buf
                self._spmm(True, None, self.dz2, dt2, None, 1.0 + c.eps, plan=self._plan_t1)
                w2 = c.nn.weight
            else:
                self._spmm(True, None, self.dz2, dt2, None, 1.0 + c.eps)
                w2 = c.nn.weight
            if chain:
                return                    # (the NEXT iteration's Del-1 pass forms dh[S1] = (dt2[S1] W2) * [z1[S1] > 0] itself)
        else:
            chain = self._chain1
            dt2, da_s, da_d = ops.gat_backward_raw(g, self._h2, self._a_src, self._a_dst, self._rowmax, self._rowsum,
                                                   self.dz2, c.negative_slope,
                                                   plan=self._plan2 if self._rows_only else None,
                                                   plan_t=self._plan_t1 if self._rows_only else None,
                                                   bufs=self._gat_bufs if chain else None)
            if chain:
                return            # (the NEXT iteration's Del-1 pass forms dh[S1] from dt2, da_src, da_dst and the stored sign pattern)
            w2 = c.lin_src.weight
            if self._mfma_weight(w2) and os.environ.get('GD_NO_GAT_RANK1_EPILOGUE') != '1':
                # dh2 = A_alpha^T dy + da_src (x) att_src + da_dst (x) att_dst only feeds the product below: the two
                # rank-1 terms are added on ITS output side, (a (x) u) W2 = a (x) (u W2), with u W2 constants of the
                # frozen backbone - the pass over dh2 (gd_rank1_add2_f32) disappears
                ops.rows_gemm(dt2, self.idx1, w2, trans_w=False, out=self.dh, gate_bits=self.z1_pos,
                              rank1=(da_s, self._gat_r1[0], da_d, self._gat_r1[1]))
                return
            ops.rank1_add2_(dt2, da_s, c.att_src, da_d, c.att_dst)
        # dh[S1] = (dt2[S1] @ W2) * [z1[S1] > 0]   (W2 is [out, in] = [d_in, d_out] of this product; the
        # ReLU backward is applied in the GEMM epilogue so dh can outlive this iteration's z1)
        ops.rows_gemm(dt2, self.idx1, w2, trans_w=False, out=self.dh, gate_bits=self.z1_pos)

    def _spmm(self, transposed, val, x, y, bias, self_coef, x_self=None, plan=None):
        g = self.graph
        if transposed:
            ops._spmm_raw(g.rowptr_t, g.col_t, val, x, bias, self_coef, self.n, plan or g.plan_t, out=y, x_self=x_self)
        else:
            ops._spmm_raw(g.rowptr, g.col, val, x, bias, self_coef, self.n, plan or g.plan, out=y, x_self=x_self)

    def _wgrad(self, a_compact, g, g_idx, n_sel, out, accumulate, ws, adam=None, g_add=None, a_idx=None):
        """out (+)= a^T (g + g_add) over the selected rows; with `adam` the optimizer update of that Del
        weight is applied inside the split-K reduction (one launch less)."""
        d_a, d_b = a_compact.shape[1], g.shape[1]
        if adam is not None and self._tail:
            # partial products only: the reduction, the Adam update and the loss finalize are one launch at the end of the step
            adam.applied += 1
            self._tail_acc[0 if adam is self.adam1 else 1] = int(accumulate)
            check(_lib.lib().gd_rows_gemm_wgrad_f32(ptr(a_compact), a_compact.stride(0), ptr(a_idx), ptr(g), g.stride(0),
                                                    ptr(g_idx), None, ptr(g_add), n_sel, d_a, d_b, None,
                                                    int(accumulate), ptr(ws), stream_ptr(g.device)),
                  'gd_rows_gemm_wgrad_f32')
            return
        if adam is not None and self._overlap:
            adam.applied += 1
            check(_lib.lib().gd_rows_gemm_wgrad_f32(ptr(a_compact), a_compact.stride(0), ptr(a_idx), ptr(g), g.stride(0),
                                                    ptr(g_idx), None, ptr(g_add), n_sel, d_a, d_b, None,
                                                    int(accumulate), ptr(ws), stream_ptr(g.device)),
                  'gd_rows_gemm_wgrad_f32')
            self._reduce_on_side(ws, n_sel, d_a, d_b, out, accumulate, adam)
            return
        if adam is not None:
            adam.applied += 1
            check(_lib.lib().gd_rows_gemm_wgrad_adam_f32(
                ptr(a_compact), a_compact.stride(0), ptr(a_idx), ptr(g), g.stride(0), ptr(g_idx), None, ptr(g_add), n_sel,
                d_a, d_b, ptr(out), int(accumulate), ptr(ws), ptr(adam.param), ptr(adam.m), ptr(adam.v),
                ptr(adam.iter_ctr), adam.lr, adam.betas[0], adam.betas[1], adam.eps, stream_ptr(g.device)),
                'gd_rows_gemm_wgrad_adam_f32')
            return
        check(_lib.lib().gd_rows_gemm_wgrad_f32(ptr(a_compact), a_compact.stride(0), ptr(a_idx), ptr(g), g.stride(0),
                                                ptr(g_idx), None, ptr(g_add), n_sel, d_a, d_b, ptr(out),
                                                int(accumulate), ptr(ws), stream_ptr(g.device)),
              'gd_rows_gemm_wgrad_f32')

    def _fork(self):
        """Context: launches inside go to the side stream, ordered after everything enqueued on the current one."""
        self._side.wait_stream(torch.cuda.current_stream())
        return torch.cuda.stream(self._side)

    def _reduce_on_side(self, ws, n_sel, d_a, d_b, out, accumulate, adam):
        with self._fork():
            check(_lib.lib().gd_rows_gemm_wgrad_reduce_f32(
                ptr(ws), n_sel, d_a, d_b, ptr(out), int(accumulate), ptr(adam.param), ptr(adam.m), ptr(adam.v),
                ptr(adam.iter_ctr), adam.lr, adam.betas[0], adam.betas[1], adam.eps, stream_ptr(out.device)),
                'gd_rows_gemm_wgrad_reduce_f32')

    # ------------------------------------------------------------------ one iteration
    def _wgrad1(self, accumulate, g_add):
        """g1 (+)= xs1^T (dz1 + g_add) followed by Adam on W_D1.  With the fused form dz1 = coef (z1 - tbar)
        is formed inside the kernel's fetch from the folded layer-1 loss terms (no loss kernel, no dz1
        buffer traffic) and the layer-1 loss sums come out as per-block partials for the finalize kernel."""
        a1, a1_idx = (self.pre1, self.idx1) if self._split1 else (self.xs1, None)
        if not self._fuse_loss1:
            self._wgrad(a1, self.dz1, self.idx1, self.s1, self.g1, accumulate, self.ws1, adam=self.adam1,
                        g_add=g_add, a_idx=a1_idx)
            return
        a = self.adam1
        a.applied += 1
        ov = self._overlap or self._tail
        if self._tail:
            self._tail_acc[0] = int(accumulate)
        check(_lib.lib().gd_rows_gemm_wgrad_loss_f32(
            ptr(a1), a1.stride(0), ptr(a1_idx), ptr(self.z1), self.z1.stride(0), ptr(self.idx1),
            ptr(self._slot1), ptr(self.t1.tm), ptr(self.t1.coef), ptr(self._cnt_signed1), ptr(g_add), self.s1, self.h,
            self.h, None if ov else ptr(self.g1), int(accumulate), ptr(self.ws1), ptr(self._lp1),
            None if ov else ptr(a.param), ptr(a.m), ptr(a.v),
            ptr(a.iter_ctr), a.lr, a.betas[0], a.betas[1], a.eps, stream_ptr(self.x.device)),
            'gd_rows_gemm_wgrad_loss_f32')
        if self._overlap:
            self._reduce_on_side(self.ws1, self.s1, self.h, self.h, self.g1, accumulate, a)

    def _iteration(self):
        """One training iteration, a single dependent chain of launches (every kernel here fills the chip on
        its own: running the layer-1 loss branch on a second stream measured within 1 % of this).

        W_D1 receives gradient from the layer-1 loss (dz1 rows) and, through conv2, from the layer-2
        loss (dh rows, already ReLU-gated).  Both are products with the same loop-invariant operand
        xs1 = conv1(x)[S1], so they are taken in ONE pass: xs1^T (dz1 + dh).
          both_all        the two belong to the same iteration; g1 accumulates for ever (upstream
                          never calls zero_grad on this path, gnndelete_nodeemb.py:215-230);
          both_layerwise  optimizer[0].step() sees loss-1 of THIS iteration plus the loss-2 gradient
                          left in .grad by the PREVIOUS one (zero_grad only after loss-1's step,
                          :232-262), i.e. dh is consumed one iteration late (zeros at iteration 0)."""
        lt = self.loss_type
        # cached constant operands (transposed / padded / packed copies of frozen tensors) used in here are pinned by
        # this engine: a captured graph replays from their addresses (ops.keep_constants)
        with torch.no_grad(), ops.keep_constants(self._const_refs):
            # ---- forward, layer 1
            if not self.cache_layer1:
                self._conv1_forward()
            if self.cache_layer1 and not self._split1:
                ops.rows_gemm(self.p1, self.idx1, self.wd1, out=self.z1, sign_bits=self.z1_pos)   # other rows stay = p1
            elif self._fuse_del1:
                self._del1_fused(self.dh if lt == 'both_layerwise' else None)
            else:
                ops.rows_gemm(self.pre1, self.idx1, self.wd1, out=self.z1, save_in=self.xs1, sign_bits=self.z1_pos)
            fused_fin = self.t1.folded and self.t2.folded       # partials reduced by the finalize kernel
            if not fused_fin:
                self.sums.zero_()
            s1 = None if fused_fin else self.sums[0:2]
            s2 = None if fused_fin else self.sums[2:4]
            # ---- layer-1 loss (+ its W_D1 step for the layer-wise types)
            if not self._fuse_loss1 and lt != 'only2_all':      # (only2_all: neither its update nor its log line reads layer 1)
                self.t1.launch(self.z1, self.dz1, s1)
            elif self._out1 and lt != 'only2_all' and not self._out_pair:
                self.t1.launch_outside(self.z1, None, self._lp1[2 * self._lp1_blocks:])
            if self._fuse_del1:
                pass                                             # (its weight-gradient partials came out of the Del-1 pass)
            elif lt == 'both_layerwise' and self._side_w1:
                with self._fork():
                    self._wgrad1(False, self.dh)
            elif lt == 'both_layerwise':
                self._wgrad1(False, self.dh)
            elif lt == 'only1':
                self._wgrad1(False, None)
            # ---- forward layer 2 + its loss
            self._conv2_forward()
            if self._fuse_l2:
                self._del2_fused()
                if self._out2 and self._out_pair and lt != 'only2_all':
                    # both layers' DEC rows outside the Del rows in ONE launch (layer 1: loss sums only; layer 2: gradient rows)
                    check(_lib.lib().gd_rowtarget_mse_pair_f32(
                        *self.t1.outside_args(self.z1, None, self._lp1[2 * self._lp1_blocks:]),
                        *self.t2.outside_args(self.p2, self.dz2, self._lp2[2 * self._lp2_blocks:]), stream_ptr(self.x.device)),
                        'gd_rowtarget_mse_pair_f32')
                elif self._out2:
                    self.t2.launch_outside(self.p2, self.dz2, self._lp2[2 * self._lp2_blocks:])
            else:
                ops.rows_gemm(self.p2, self.idx2, self.wd2, out=self.z2, save_in=self.xs2)
                self.t2.launch(self.z2, self.dz2, s2)
            # ---- backward + update
            if self._side_w1:                                    # (join: the backward below overwrites dh, the tail reads the partials)
                torch.cuda.current_stream().wait_stream(self._side)
            if lt == 'both_layerwise':
                self._layer2_backward()                          # leaves dh for the next iteration
                if not self._fuse_l2 and not self._tail:
                    self.adam2.apply(self.g2)
            elif lt == 'both_all':
                self._layer2_backward(g2_accumulate=True)
                self._wgrad1(True, self.dh)
                if not self._fuse_l2 and not self._tail:
                    self.adam2.apply(self.g2)
            elif lt == 'only2_layerwise':
                self._layer2_backward(to_w1=False)
                if not self._fuse_l2 and not self._tail:
                    self.adam2.apply(self.g2)
            elif lt == 'only2_all':
                self._layer2_backward()
                a1, a1_idx = (self.pre1, self.idx1) if self._split1 else (self.xs1, None)
                self._wgrad(a1, self.dh, self.idx1, self.s1, self.g1, False, self.ws1, adam=self.adam1, a_idx=a1_idx)
                if not self._fuse_l2 and not self._tail:
                    self.adam2.apply(self.g2)
            # ---- loss sums -> history ring, advance the iteration counter (Adam's step number)
            p1, n1 = ((self._lp1, self._lp1_blocks + (self.t1.outside_blocks() if self._out1 else 0)) if self._fuse_loss1
                      else (self.t1.partials, self.t1.n_partial_blocks()))
            p2, n2 = ((self._lp2, self._lp2_blocks + (self.t2.outside_blocks() if self._out2 else 0)) if self._fuse_l2
                      else (self.t2.partials, self.t2.n_partial_blocks()))

            def finalize():
                check(_lib.lib().gd_loss_finalize_f32(
                    ptr(p1) if fused_fin else None, n1 if fused_fin else 0,
                    ptr(p2) if fused_fin else None, n2 if fused_fin else 0,
                    None if fused_fin else ptr(self.sums), ptr(self.hist), self.hist.shape[0], ptr(self.hist_pos),
                    ptr(self.iter_ctr), stream_ptr(self.x.device)), 'gd_loss_finalize_f32')
            if self._tail:
                a1, a2 = self.adam1, self.adam2
                lib_ = _lib.lib()
                nw1 = self._lp1_blocks if self._fuse_del1 else lib_.gd_rows_gemm_wgrad_blocks(self.s1)      # partial matrices of W_D1
                check(lib_.gd_step_tail_parts_f32(
                    ptr(self.ws1), nw1, self.h, self._tail_acc[0], ptr(self.g1), ptr(a1.param), ptr(a1.m), ptr(a1.v),
                    ptr(self.ws2), self._lp2_blocks if self._fuse_wg2 else lib_.gd_rows_gemm_wgrad_blocks(self.s2), self.o,
                    self._tail_acc[1], ptr(self.g2), ptr(a2.param),
                    ptr(a2.m), ptr(a2.v), a1.lr, a1.betas[0], a1.betas[1], a1.eps, ptr(p1), n1, ptr(p2), n2, ptr(self.hist),
                    self.hist.shape[0], ptr(self.hist_pos), ptr(self.iter_ctr), ptr(self._arrive), stream_ptr(self.x.device)),
                    'gd_step_tail_parts_f32')
            elif self._overlap:
                with self._fork():
                    finalize()
                torch.cuda.current_stream().wait_stream(self._side)       # join: the iteration ends when both branches have
            else:
                finalize()
            if self._user_wd2 is not None:          # padded class dimension: the caller's W_D2 is the top-left block
                self._user_wd2.data.copy_(self.wd2.data[:self._o_true, :self._o_true])

    def _del1_fused(self, g_add):
        """Del-1 forward (+ sign bits) + folded layer-1 loss + the W_D1 weight gradient's partial sums in one kernel
        (csrc/del_fused.hip, del1_loss_wgrad_ws_kernel); the tail launch reduces them and steps Adam."""
        self.adam1.applied += 1
        self._tail_acc[0] = 0
        if self._chain1:
            c2 = self.model.conv2
            if self._mode == 'gat':
                w_next, dt = c2.lin_src.weight.detach(), self._gat_bufs['dh']
                rank1 = (ptr(self._gat_bufs['da_src']), ptr(self._gat_r1[0]), ptr(self._gat_bufs['da_dst']), ptr(self._gat_r1[1]))
            else:
                w_next, dt = (c2.lin if self._mode == 'gcn' else c2.nn).weight.detach(), self._dt2_keep
                rank1 = (None, None, None, None)
            check(_lib.lib().gd_del1_chain_loss_wgrad_f32(
                ptr(self.pre1), self.pre1.stride(0), ptr(self.idx1), self.s1, ptr(self.wd1), self.h, ptr(self.z1), self.z1.stride(0),
                ptr(self.z1_pos), ptr(self._slot1), ptr(self.t1.tm), ptr(self.t1.coef), ptr(self._cnt_signed1), ptr(dt),
                dt.stride(0), self.o, ptr(w_next), *rank1, ptr(self._lp1), ptr(self.ws1), self._lp1_blocks,
                stream_ptr(self.x.device)), 'gd_del1_chain_loss_wgrad_f32')
            return
        check(_lib.lib().gd_del1_loss_wgrad_f32(
            ptr(self.pre1), self.pre1.stride(0), ptr(self.idx1), self.s1, ptr(self.wd1), self.h, ptr(self.z1), self.z1.stride(0),
            ptr(self.z1_pos), ptr(self._slot1), ptr(self.t1.tm), ptr(self.t1.coef), ptr(self._cnt_signed1), ptr(g_add),
            g_add.stride(0) if g_add is not None else 0, ptr(self._lp1), ptr(self.ws1), self._lp1_blocks, stream_ptr(self.x.device)),
            'gd_del1_loss_wgrad_f32')

    def _del2_fused(self):
        """Del-2 forward + folded layer-2 loss + Del-2 input gradient (+ the W_D2 weight gradient's partial sums) in one
        kernel (csrc/del_fused.hip)."""
        if self._fuse_wg2:
            check(_lib.lib().gd_del_loss_bwd_wgrad_parts_f32(
                ptr(self.p2), self.p2.stride(0), ptr(self.idx2), self.s2, ptr(self.wd2), self.o, ptr(self._slot2),
                ptr(self.t2.tm), ptr(self.t2.coef), ptr(self._cnt_signed2), None, self.o,
                ptr(self.dz2), self.dz2.stride(0), ptr(self._lp2), ptr(self.ws2), self._lp2_blocks, stream_ptr(self.x.device)),
                'gd_del_loss_bwd_wgrad_parts_f32')
            return
        check(_lib.lib().gd_del_loss_bwd_f32(
            ptr(self.p2), self.p2.stride(0), ptr(self.idx2), self.s2, ptr(self.wd2), self.o, ptr(self._slot2),
            ptr(self.t2.tm), ptr(self.t2.coef), ptr(self._cnt_signed2), ptr(self.dz2c), self.dz2c.stride(0),
            ptr(self.dz2), self.dz2.stride(0), ptr(self._lp2), stream_ptr(self.x.device)), 'gd_del_loss_bwd_f32')

    def _layer2_backward(self, to_w1=True, g2_accumulate=False):
        """g2 (+)= dW_D2; with to_w1 also dh[S1] = d loss2 / d z1[S1] (ReLU-gated)."""
        if self._fuse_wg2:         # the partial sums are in ws2 already (fused Del-2 kernel); the tail launch reduces them
            self.adam2.applied += 1
            self._tail_acc[1] = int(g2_accumulate)
        elif self._fuse_l2:        # dz2 (compact) and dp2 (in self.dz2) were produced by the fused Del-2 kernel,
            # so nothing reads W_D2 any more this iteration: its Adam step rides on the split-K reduction
            self._wgrad(self.p2, self.dz2c, None, self.s2, self.g2, g2_accumulate, self.ws2, a_idx=self.idx2,
                        adam=self.adam2)
        elif self._split2:
            self._wgrad(self.p2, self.dz2, self.idx2, self.s2, self.g2, g2_accumulate, self.ws2, a_idx=self.idx2,
                        adam=self.adam2 if self._tail else None)
        else:
            self._wgrad(self.xs2, self.dz2, self.idx2, self.s2, self.g2, g2_accumulate, self.ws2,
                        adam=self.adam2 if self._tail else None)
        if not to_w1:
            return
        if not self._fuse_l2:
            # dz2 -> dp2 in place (Del-2 input gradient on the masked rows, identity elsewhere)
            ops.rows_gemm(self.dz2, self.idx2, self.wd2, trans_w=True, out=self.dz2)
        self._conv2_backward_to_s1()

    # ------------------------------------------------------------------ public
    def step(self):
        if not self._use_graph:
            self._iteration()
        elif self._graph is None:
            # two eager warm-up iterations would change the trajectory: warm the libraries up on
            # a scratch copy of the mutable state instead, then capture
            self._capture()
            self._graph.replay()
        else:
            self._graph.replay()
        self.steps_done += 1

    def run(self, n_steps, unroll=4):
        """n_steps iterations, `unroll` of them per graph launch where possible (a replay boundary costs ~8 us of
        launch latency that a kernel boundary inside a graph does not); the remainder goes step by step.  Same
        iterations, same order."""
        if not self._use_graph or unroll <= 1:
            for _ in range(n_steps):
                self.step()
            return
        if n_steps >= unroll:
            self.prepare_unrolled(unroll)
            while n_steps >= unroll:
                self._graph_k[1].replay()
                self.steps_done += unroll
                n_steps -= unroll
        for _ in range(n_steps):
            self.step()

    def prepare_unrolled(self, unroll):
        """Capture the `unroll`-iteration graph now (nothing is executed, no state changes)."""
        if not self._use_graph or unroll <= 1:
            return
        if self._graph is None:
            self._capture()                               # warms the libraries up on scratch state first
        if getattr(self, '_graph_k', None) is None or self._graph_k[0] != unroll:
            saved = [t.clone() for t in self._mutable_state()]
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                for _ in range(unroll):
                    self._iteration()
            # the first launch of a graph exec uploads it to the device (tens of us): do it now, on state that is
            # restored right after - a short run (the driver times 20 steps = 5 launches) should not pay it
            graph.replay()
            torch.cuda.synchronize()
            for t, sv in zip(self._mutable_state(), saved):
                t.copy_(sv)
            self._graph_k = (unroll, graph)

    def _mutable_state(self):
        state = [self.wd1.data, self.wd2.data, self.g1, self.g2, self.adam1.m, self.adam1.v, self.iter_ctr,
                 self.adam2.m, self.adam2.v, self.hist, self.hist_pos, self.dz1, self.dz2, self.dh]
        if getattr(self, '_chain1', False):                 # carried from one iteration to the next in the chained form
            state += [self.z1_pos] + ([self._dt2_keep] if self._dt2_keep is not None else
                                      [self._gat_bufs[k] for k in ('dh', 'da_src', 'da_dst')])
        if getattr(self, '_arrive', None) is not None:      # step_tail's check-in counter: a launch that did not finish must not
            state.append(self._arrive)                       # leave it non-zero for the replays
        if getattr(self, '_user_wd2', None) is not None:    # the caller's W_D2 behind a padded class dimension
            state.append(self._user_wd2.data)
        return state

    def _capture(self):
        saved = [t.clone() for t in self._mutable_state()]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self._iteration()                      # warm-up (allocator, rocBLAS handles, code objects)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            self._iteration()
        graph.replay()                             # first launch = upload of the exec; on scratch state
        torch.cuda.synchronize()
        for t, s in zip(self._mutable_state(), saved):
            t.copy_(s)                             # undo the warm-up iterations
        self._graph = graph

    def loss_history(self):
        """[steps, 3] host tensor: train_loss, loss_r, loss_l as the reference logs them."""
        k = min(self.steps_done, self.hist.shape[0])
        s = self.hist[:k].double().cpu()
        nan = float('nan')
        r1 = (s[:, 0] + self.t1.k_const[0]) / self.t1.n_r if self.t1.count_r else s[:, 0] * nan
        l1 = (s[:, 1] + self.t1.k_const[1]) / self.t1.n_l if self.t1.count_l else s[:, 1] * nan
        r2 = (s[:, 2] + self.t2.k_const[0]) / self.t2.n_r if self.t2.count_r else s[:, 2] * nan
        l2 = (s[:, 3] + self.t2.k_const[1]) / self.t2.n_l if self.t2.count_l else s[:, 3] * nan
        a, lt = self.alpha, self.loss_type
        if lt in ('both_all', 'both_layerwise'):
            loss_r, loss_l = r1 + r2, l1 + l2
            loss = a * loss_r + (1 - a) * loss_l
        elif lt == 'only2_layerwise':
            loss_r, loss_l = r1 + r2, l1 + l2
            loss = a * r2 + (1 - a) * l2
        elif lt == 'only2_all':
            loss_r, loss_l = r2, l2
            loss = l2 + a * r2
        else:
            loss_r, loss_l = r1, l1
            loss = l1 + a * r1
        return torch.stack([loss, loss_r, loss_l], 1)
