"""Row-partitioned Del-training step for N GPUs of one node (one process per GPU, RCCL over xGMI).

north_star: "graphs that outgrow one GPU are 1-D partitioned across the 8 GPUs with RCCL exchange
of the aggregates and of the Del-operator gradients".  The partition used here is by TARGET rows
(contiguous equal blocks of the engine's locality order), which turns the exchange of partial
aggregates into an all-gather of the layer input instead of an all-reduce of [N, d] partial sums
(half the bytes, and the per-row summation order - hence the result - stays that of one GPU):

  every rank   t1 = x W1^T for ALL rows (the input is static and replicated; recomputing 7.7 GF
               is cheaper than gathering 121 MB over xGMI)
  own rows     z1 = A t1 + b1, Del-1, t2 = relu(z1) W2^T
  exchange 1   t2 rows                  [N, out] fp32 (see "exchange" below)
  own rows     z2 = A t2 + b2, Del-2, DEC + NI losses, dW_D2 partial, dz2 <- dz2 W_D2^T
  exchange 2   dz2 rows                 [N, out] fp32
  own rows     dt2 = A^T dz2, dh = dt2 W2, dW_D1 partials
  exchange 3   all-reduce of ONE packed buffer: dW_D1 (loss-1 part), dW_D1 (loss-2 part), dW_D2,
               4 loss sums  (~144 KiB)
  every rank   the --loss_type bookkeeping + Adam, identically (replicated Del weights)

Exchange 1/2 are sparse by default (`exchange='halo'`): a rank sends each peer only the rows
that peer's SpMM gathers (all-to-all with per-pair sorted row lists that both sides derive from
the replicated CSR); with the locality order most gathers are rank-local, so this is a fraction
of the dense all-gather (`exchange='allgather'`), which matters most at N=2 where a pair shares
a single xGMI link.

The four compute segments are captured as hipGraphs; the three collectives are issued between
the replays.  GCN and GIN backbones (GAT's backward would need the per-edge attention of remote
rows)."""
import torch

from . import _lib, ops
from ._lib import check, ptr, stream_ptr
from .collectives import all_gather_rows, all_reduce_sum, exchange_rows, halo_lists, row_blocks
from .engine import LOSS_TYPES, _Adam, _LayerTerms, _loss_coefficients
from .graph import SplitPlan, graph_for
from .nn import GCNConv, GINConv


class PartitionedNodeembEngine:
    def __init__(self, model, x, edge_index, z1_ori, z2_ori, pos_edge, neg_edge, ni_mask1, ni_mask2, rank, world,
                 loss_type='both_layerwise', alpha=0.5, lr=1e-3, reduction='mean', use_graph=True, history=4096,
                 reorder=True, group=None, exchange='halo'):
        assert loss_type in LOSS_TYPES, loss_type
        conv2 = model.conv2
        if not isinstance(conv2, (GCNConv, GINConv)):
            raise NotImplementedError('PartitionedNodeembEngine supports GCN and GIN backbones')
        dev = x.device
        if dev.type != 'cuda':
            raise _lib.GnnDeleteHipError('PartitionedNodeembEngine needs CUDA(HIP) tensors (no CPU fallback)')
        self.model, self.loss_type, self.alpha = model, loss_type, alpha
        self.rank, self.world, self.group = rank, world, group
        self.n = n = x.shape[0]
        m1, m2 = model.deletion1.mask.to(dev), model.deletion2.mask.to(dev)
        ni_mask1, ni_mask2 = ni_mask1.to(dev), ni_mask2.to(dev)
        pos_edge, neg_edge = pos_edge.to(dev), neg_edge.to(dev)
        self.perm = None
        if reorder and edge_index.shape[1] > 0 and n > 4096:
            from .reorder import locality_order
            perm, inv = locality_order(edge_index, n)      # deterministic: identical on every rank
            self.perm = perm
            x = x[perm].contiguous()
            edge_index = inv[edge_index]
            z1_ori, z2_ori = z1_ori[perm], z2_ori[perm]
            m1, m2, ni_mask1, ni_mask2 = m1[perm], m2[perm], ni_mask1[perm], ni_mask2[perm]
            pos_edge, neg_edge = inv[pos_edge], inv[neg_edge]
        self.x = x
        self.chunk, self.n_pad = row_blocks(n, world)
        self.lo = lo = min(n, rank * self.chunk)
        self.hi = hi = min(n, lo + self.chunk)
        self.wd1, self.wd2 = model.deletion1.deletion_weight, model.deletion2.deletion_weight
        self.h, self.o = self.wd1.shape[0], self.wd2.shape[0]

        def local(mask):
            idx = mask.nonzero().flatten()
            return idx[(idx >= lo) & (idx < hi)].to(torch.int32)
        self.idx1, self.idx2 = local(m1), local(m2)
        self.s1, self.s2 = int(self.idx1.numel()), int(self.idx2.numel())
        z1_ori, z2_ori = ops._f32_rows(z1_ori), ops._f32_rows(z2_ori)
        coef_r, coef_l = _loss_coefficients(loss_type, alpha)
        self.t1 = _LayerTerms(pos_edge, neg_edge, ni_mask1, z1_ori, coef_r, coef_l, reduction, (lo, hi))
        self.t2 = _LayerTerms(pos_edge, neg_edge, ni_mask2, z2_ori, coef_r, coef_l, reduction, (lo, hi))
        # constants of the folded losses are per-rank partial sums too: reduce them once
        k = torch.tensor(self.t1.k_const + self.t2.k_const, dtype=torch.float64, device=dev)
        all_reduce_sum(k, world, group)
        self.k_const = k.tolist()

        self._mode = 'gcn' if isinstance(conv2, GCNConv) else 'gin'
        self.graph = g = graph_for(edge_index, n, 'gcn' if self._mode == 'gcn' else 'sum')
        self.plan = SplitPlan(g.rowptr, row_range=(lo, hi))
        self.plan_t = SplitPlan(g.rowptr_t, row_range=(lo, hi))

        f32 = dict(dtype=torch.float32, device=dev)
        self.z1 = torch.zeros(self.n_pad, self.h, **f32)
        self.z2 = torch.zeros(self.n_pad, self.o, **f32)
        self.t2_full = torch.zeros(self.n_pad, self.o, **f32)
        self.dz1 = torch.zeros(self.n_pad, self.h, **f32)
        self.dz2 = torch.zeros(self.n_pad, self.o, **f32)        # all-gathered in place (own block written)
        self.dt2 = torch.zeros(self.n_pad, self.o, **f32)
        self.dh = torch.zeros(self.n_pad, self.h, **f32)
        self.xs1 = torch.empty(max(1, self.s1), self.h, **f32)
        self.xs2 = torch.empty(max(1, self.s2), self.o, **f32)
        hh, oo = self.h * self.h, self.o * self.o
        self.pack = torch.zeros(2 * hh + oo + 4, **f32)          # [dW1 loss-1 | dW1 loss-2 | dW2 | sums]
        self.p_a = self.pack[:hh].view(self.h, self.h)
        self.p_b = self.pack[hh:2 * hh].view(self.h, self.h)
        self.p_c = self.pack[2 * hh:2 * hh + oo].view(self.o, self.o)
        self.p_sums = self.pack[2 * hh + oo:]
        self.g1 = torch.zeros_like(self.wd1)
        self.g2 = torch.zeros_like(self.wd2)
        self.ws1 = torch.empty(max(1, _lib.lib().gd_rows_gemm_wgrad_workspace(self.s1, self.h, self.h)), **f32)
        self.ws2 = torch.empty(max(1, _lib.lib().gd_rows_gemm_wgrad_workspace(self.s2, self.o, self.o)), **f32)
        self.adam1, self.adam2 = _Adam(self.wd1, lr), _Adam(self.wd2, lr)
        self.hist = torch.zeros(history, 4, **f32)
        self.hist_pos = torch.zeros(1, dtype=torch.long, device=dev)
        self.steps_done = 0
        self._use_graph = use_graph
        self._graphs = None
        self.needs_b = loss_type in ('both_all', 'both_layerwise', 'only2_all')
        self.needs_a = loss_type in ('both_all', 'both_layerwise', 'only1')
        # sparse exchange: only the rows a peer's SpMM gathers travel (with the locality order that
        # is a fraction of a full all-gather); 'allgather' keeps the dense collective
        assert exchange in ('halo', 'allgather')
        self.exchange = exchange if world > 1 else 'allgather'
        if self.exchange == 'halo':
            self.halo_f = halo_lists(g.rowptr, g.col, n, rank, world, self.chunk)
            self.halo_b = halo_lists(g.rowptr_t, g.col_t, n, rank, world, self.chunk)
            self.recv_f = torch.empty(sum(self.halo_f[3]), self.o, **f32)
            self.recv_b = torch.empty(sum(self.halo_b[3]), self.o, **f32)

    # ------------------------------------------------------------------ helpers
    def _linear(self, x, weight, relu_in=False, out=None):
        out_f, in_f = weight.shape
        if x.shape[0] == 0:
            return out if out is not None else x.new_zeros(0, out_f)
        if in_f % 32 == 0 and out_f % 32 == 0 and out_f <= 128 and in_f * out_f * 4 <= 64 * 1024:
            return ops.rows_gemm(x, None, weight, trans_w=True, relu_in=relu_in, out=out)
        res = torch.nn.functional.linear(torch.relu(x) if relu_in else x, weight)
        if out is not None:
            out.copy_(res)
            return out
        return res

    def _spmm_own_rows(self, transposed, val, x, y, bias, self_coef):
        g = self.graph
        if transposed:
            ops._spmm_raw(g.rowptr_t, g.col_t, val, x, bias, self_coef, self.n, self.plan_t, out=y)
        else:
            ops._spmm_raw(g.rowptr, g.col, val, x, bias, self_coef, self.n, self.plan, out=y)

    def _wgrad(self, a_compact, g, g_idx, n_sel, relu_mask, out, ws):
        check(_lib.lib().gd_rows_gemm_wgrad_f32(ptr(a_compact), a_compact.stride(0), None, ptr(g), g.stride(0),
                                                ptr(g_idx), ptr(relu_mask), None, n_sel, a_compact.shape[1], g.shape[1],
                                                ptr(out), 0, ptr(ws), stream_ptr(g.device)), 'gd_rows_gemm_wgrad_f32')

    def _weights(self):
        c1, c2 = self.model.conv1, self.model.conv2
        if self._mode == 'gcn':
            return c1.lin.weight, c1.bias, c2.lin.weight, c2.bias, 0.0
        return c1.nn.weight, c1.nn.bias, c2.nn.weight, c2.nn.bias, 1.0 + c2.eps

    # ------------------------------------------------------------------ the four segments
    def _seg_forward1(self):
        w1, b1, w2, b2, sc = self._weights()
        val = self.graph.val
        lo, hi = self.lo, self.hi
        t1 = self._linear(self.x, w1)                                   # all rows (replicated)
        self._spmm_own_rows(False, val, t1, self.z1, b1, sc)
        ops.rows_gemm(self.z1, self.idx1, self.wd1, out=self.z1, save_in=self.xs1)
        self._linear(self.z1[lo:hi], w2, relu_in=True, out=self.t2_full[lo:hi])

    def _seg_forward2_backward1(self):
        w1, b1, w2, b2, sc = self._weights()
        self._spmm_own_rows(False, self.graph.val, self.t2_full, self.z2, b2, sc)
        ops.rows_gemm(self.z2, self.idx2, self.wd2, out=self.z2, save_in=self.xs2)
        self.p_sums.zero_()
        self.t1.launch(self.z1, self.dz1, self.p_sums[0:2])
        self.t2.launch(self.z2, self.dz2, self.p_sums[2:4])
        if self.loss_type != 'only1':
            self._wgrad(self.xs2, self.dz2, self.idx2, self.s2, None, self.p_c, self.ws2)
            if self.needs_b:
                ops.rows_gemm(self.dz2, self.idx2, self.wd2, trans_w=True, out=self.dz2)

    def _seg_backward2(self):
        w1, b1, w2, b2, sc = self._weights()
        if self.needs_a:
            self._wgrad(self.xs1, self.dz1, self.idx1, self.s1, None, self.p_a, self.ws1)
        if self.needs_b:
            self._spmm_own_rows(True, self.graph.val_t, self.dz2, self.dt2, None, sc)
            ops.rows_gemm(self.dt2, self.idx1, w2, trans_w=False, out=self.dh)
            self._wgrad(self.xs1, self.dh, self.idx1, self.s1, self.z1, self.p_b, self.ws1)

    def _seg_update(self):
        lt = self.loss_type
        self.hist.index_copy_(0, self.hist_pos, self.p_sums[None])
        self.hist_pos.add_(1).remainder_(self.hist.shape[0])
        if lt == 'both_layerwise':
            self.g1.add_(self.p_a)
            self.adam1.apply(self.g1)
            self.g1.copy_(self.p_b)
            self.g2.copy_(self.p_c)
            self.adam2.apply(self.g2)
        elif lt == 'both_all':
            self.g1.add_(self.p_a).add_(self.p_b)
            self.g2.add_(self.p_c)
            self.adam1.apply(self.g1)
            self.adam2.apply(self.g2)
        elif lt == 'only2_layerwise':
            self.g2.copy_(self.p_c)
            self.adam2.apply(self.g2)
        elif lt == 'only2_all':
            self.g1.copy_(self.p_b)
            self.g2.copy_(self.p_c)
            self.adam1.apply(self.g1)
            self.adam2.apply(self.g2)
        else:
            self.g1.copy_(self.p_a)
            self.adam1.apply(self.g1)

    def _segments(self):
        return [self._seg_forward1, self._seg_forward2_backward1, self._seg_backward2, self._seg_update]

    def _halo(self, full, lists, recv_buf):
        send_rows, in_splits, recv_rows, out_splits = lists
        exchange_rows(full.index_select(0, send_rows), recv_buf, in_splits, out_splits, self.world, self.group)
        full.index_copy_(0, recv_rows, recv_buf)

    def _exchange(self, after_segment):
        if after_segment == 0:
            if self.exchange == 'halo':
                self._halo(self.t2_full, self.halo_f, self.recv_f)
            else:
                all_gather_rows(self.t2_full, self.rank, self.world, self.chunk, self.group)
        elif after_segment == 1 and self.needs_b:
            if self.exchange == 'halo':
                self._halo(self.dz2, self.halo_b, self.recv_b)
            else:
                all_gather_rows(self.dz2, self.rank, self.world, self.chunk, self.group)
        elif after_segment == 2:
            all_reduce_sum(self.pack, self.world, self.group)

    # ------------------------------------------------------------------ public
    def _mutable_state(self):
        return [self.wd1.data, self.wd2.data, self.g1, self.g2, self.adam1.m, self.adam1.v, self.adam1.step,
                self.adam2.m, self.adam2.v, self.adam2.step, self.hist, self.hist_pos]

    def _run_eager(self):
        with torch.no_grad():
            for i, seg in enumerate(self._segments()):
                seg()
                self._exchange(i)

    def _capture(self):
        saved = [t.clone() for t in self._mutable_state()]
        self._run_eager()                                  # warm-up incl. the collectives (all ranks)
        torch.cuda.synchronize()
        graphs = []
        with torch.no_grad():
            for seg in self._segments():
                g = torch.cuda.CUDAGraph()
                # thread_local: the RCCL watchdog thread must not invalidate the capture
                with torch.cuda.graph(g, capture_error_mode='thread_local'):
                    seg()
                graphs.append(g)
        for t, s in zip(self._mutable_state(), saved):
            t.copy_(s)
        self._graphs = graphs

    def step(self):
        if not self._use_graph:
            self._run_eager()
        else:
            if self._graphs is None:
                self._capture()
            for i, g in enumerate(self._graphs):
                g.replay()
                self._exchange(i)
        self.steps_done += 1

    def loss_history(self):
        k = min(self.steps_done, self.hist.shape[0])
        s = self.hist[:k].double().cpu()
        kc = self.k_const
        nan = float('nan')
        r1 = (s[:, 0] + kc[0]) / self.t1.n_r if self.t1.count_r else s[:, 0] * nan
        l1 = (s[:, 1] + kc[1]) / self.t1.n_l if self.t1.count_l else s[:, 1] * nan
        r2 = (s[:, 2] + kc[2]) / self.t2.n_r if self.t2.count_r else s[:, 2] * nan
        l2 = (s[:, 3] + kc[3]) / self.t2.n_l if self.t2.count_l else s[:, 3] * nan
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
