"""CSR construction from the reference's ``edge_index`` layout ([2, E] int64, row 0 = source j,
row 1 = target i).  One-off device-side plumbing (sort / bincount / cumsum in PyTorch); the hot
kernels only ever see the resulting int32 CSR arrays.

Three aggregation flavours, matching what each torch_geometric conv does to the edge list:
  'gcn'  drop existing self loops, append exactly one per node, val = D^-1/2 (A+I) D^-1/2
  'gat'  same structure, no values (attention is computed by the kernel)
  'sum'  edges as given (GIN), no values
  'mean' edges as given, val = 1 / in-degree of the target (GraphSAGE mean aggregation)
Rows are sorted by (target, source), so the per-row summation order is deterministic.
"""
import os

import torch

from . import _lib


CHUNK = 64   # in-edges per SpMM work item (one coalesced (col, val) fetch of a 64-lane wave)


class SplitPlan:
    """Work items of the load-balanced SpMM (gd_spmm_csr_balanced_f32): every CSR row is cut
    into pieces of at most CHUNK in-edges so that hub rows are spread over many waves."""

    def __init__(self, rowptr, chunk=CHUNK, row_range=None, rows=None):
        """row_range=(lo, hi) keeps only the rows lo <= i < hi (a rank's share of a 1-D
        row partition); rows = a sorted list of row ids keeps only those (the rows an unlearning request
        can influence); item rows stay global ids."""
        dev = rowptr.device
        rpl = rowptr.long()
        if rows is not None:
            ids = rows.long()
        else:
            lo, hi = (0, rowptr.numel() - 1) if row_range is None else row_range
            ids = torch.arange(lo, hi, device=dev)
        r_start, r_end = rpl[ids], rpl[ids + 1]
        n = ids.numel()
        deg = r_end - r_start
        pieces = torch.clamp((deg + chunk - 1) // chunk, min=1)
        first = torch.cumsum(pieces, 0) - pieces
        row = torch.repeat_interleave(torch.arange(n, device=dev), pieces)
        k = torch.arange(row.numel(), device=dev) - first[row]
        start = r_start[row] + k * chunk
        end = torch.minimum(r_end[row], start + chunk)
        is_split = pieces[row] > 1
        slot = torch.where(is_split, torch.cumsum(is_split, 0) - 1, torch.full_like(row, -1))
        self.items = torch.stack([ids[row], start, end, slot], 1).to(torch.int32).contiguous()
        self.n_items = int(row.numel())
        srows = (pieces > 1).nonzero().flatten()
        split_pieces = torch.where(pieces > 1, pieces, torch.zeros_like(pieces))
        slot0 = (torch.cumsum(split_pieces, 0) - split_pieces)[srows]
        self.split = torch.stack([ids[srows], slot0, pieces[srows], torch.zeros_like(srows)], 1).to(torch.int32).contiguous()
        self.n_split = int(srows.numel())
        self.n_slots = int(split_pieces.sum())
        self._scratch = {}
        self._edges_per_item = (end - start)
        self._bounds = {}
        # one-launch form (gd_spmm_csr_onepass_f32), laid out per width on demand (onepass)
        self._row_ids, self._row_start, self._row_end = ids, r_start, r_end
        self._row_deg, self._row_pieces = deg, pieces
        self._onepass = {}
        for d in (64, 128):          # the path's widths, ahead of any hipGraph capture (the tables are built lazily)
            self.xcd_bounds(d)
            self.onepass(d)

    def xcd_bounds(self, d):
        """Item range of each of the 8 XCDs for a width-d SpMM (int32 [9] on the device, None for small plans),
        balanced by what an item costs its XCD: its in-edges (one feature row each) plus a fixed per-visit overhead
        worth about 6 KB of row traffic (measured with per-XCD time stamps on the bench graph,
        tools/experiments/spmm_lab.hip: a wave's visit is one dependent memory round trip however short the row).
        With equal item counts the XCD that gets the rows outside S_Df (self loop only) is done in half the time and
        its CUs and fabric link idle while the others still have a third of their traffic to go."""
        if self.n_items < 8 * 1024:
            return None
        a = int(min(48, max(8, 1536 // max(int(d), 1))))
        b = self._bounds.get(a)
        if b is None:
            cost = torch.cumsum((self._edges_per_item + a).double(), 0)
            dev = cost.device
            cuts = torch.searchsorted(cost, cost[-1] * torch.arange(1, 8, device=dev, dtype=torch.float64) / 8)
            b = torch.cat([cuts.new_zeros(1), cuts, cuts.new_full((1,), self.n_items)]).to(torch.int32).contiguous()
            self._bounds[a] = b
        return b

    def onepass(self, d, multirow=None):
        """(items int32 [n, 4], n, bounds int32 [9]) for gd_spmm_csr_onepass_f32 at width d.  The plan's rows are cut into
        eight contiguous ranges of equal cost (in-edges + the per-visit overhead of xcd_bounds, once per 64-edge piece),
        one per XCD.  A hub row (more than `chunk` in-edges) becomes a GROUP of four member items - the w-th contiguous
        share of the row's edges, a multiple of `chunk` - that the four waves of a block sum together; groups sit at
        4-aligned positions (padding items in front where needed) and every range limit is a multiple of 4.  Inside a
        range the rows keep their order (the sweep's window of consecutive rows is what keeps gathered rows in the XCD's
        L2: with the hub groups moved to the front of the range the d = 128 launch fetched 6 % more, d = 64 11 % more);
        only the few rows above 4 pieces, whose members walk several chunks in sequence, go first, heaviest first, so
        that they cannot end up as the tail of the sweep.

        multirow (rows per item; default by width unless GD_SPMM_MULTIROW=0: 1 above 64 floats, 2 for 33 .. 64, 4 below - what
        the kernel of that width carries accumulators for, csrc/spmm.hip MAXR): that many CONSECUTIVE light rows - adjacent in
        the plan and in the CSR, at most `chunk` in-edges together - share ONE item {first row, start, end, -(16 + v)}, v = the
        cumulative edge counts after rows 0 / 1 / 2 (7 bits each) | (rows - 1) << 21: a visit of the sweep is one dependent
        round trip however few rows it gathers, so packing rows multiplies the gathers in flight and divides the visits."""
        if multirow is None:
            multirow = 1 if (d > 64 or os.environ.get('GD_SPMM_MULTIROW') == '0') else (2 if d > 32 else 4)
        multirow = int(multirow)
        a = int(min(48, max(8, 1536 // max(int(d), 1))))
        key = (a, multirow)
        hit = self._onepass.get(key)
        if hit is not None:
            return hit
        ids, r_start, r_end, deg, pieces = self._row_ids, self._row_start, self._row_end, self._row_deg, self._row_pieces
        dev = ids.device
        n = int(ids.numel())
        chunk = CHUNK
        if n == 0:
            hit = (torch.zeros(0, 4, dtype=torch.int32, device=dev), 0, torch.zeros(9, dtype=torch.int32, device=dev))
            self._onepass[key] = hit
            return hit
        cost = torch.cumsum((deg + a * pieces).double(), 0)
        cuts = torch.searchsorted(cost, cost[-1] * torch.arange(1, 8, device=dev, dtype=torch.float64) / 8)
        lim = [0] + [int(c) for c in cuts.tolist()] + [n]
        for k in range(1, 9):
            lim[k] = max(lim[k], lim[k - 1])
        q4 = torch.arange(4, device=dev)

        def members(hub):                                   # [len(hub) * 4, 4] group member items
            share = chunk * ((pieces[hub] + 3) // 4)        # edges per member, a multiple of chunk
            ms = r_start[hub][:, None] + share[:, None] * q4[None, :]
            me = torch.minimum(ms + share[:, None], r_end[hub][:, None])
            ms = torch.minimum(ms, me)
            return torch.stack([ids[hub][:, None].expand(-1, 4), ms, me, torch.full_like(ms, -2)], 2).reshape(-1, 4)

        def light_items(rl):
            """rl: positions (ascending) of light rows -> (item rows [m, 4], index of every row's item)."""
            m = int(rl.numel())
            single = torch.stack([ids[rl], r_start[rl], r_end[rl], torch.full_like(rl, -1)], 1)
            if multirow < 2 or m < 2:
                return single, torch.arange(m, device=dev)
            # runs of rows that follow each other in the plan, in node ids and in the CSR
            adj = torch.zeros(m, dtype=torch.bool, device=dev)
            adj[1:] = (rl[1:] == rl[:-1] + 1) & (ids[rl][1:] == ids[rl][:-1] + 1) & (r_start[rl][1:] == r_end[rl][:-1])
            run = torch.cumsum((~adj).long(), 0) - 1
            first_of_run = torch.zeros(int(run[-1]) + 1, dtype=torch.long, device=dev)
            first_of_run[run[~adj]] = (~adj).nonzero().flatten()
            k_in = torch.arange(m, device=dev) - first_of_run[run]
            quad = run * (m + 1) + k_in // multirow               # fixed pairs / quadruples inside a run
            uq, inv, size = torch.unique_consecutive(quad, return_inverse=True, return_counts=True)
            dl = deg[rl]
            tot = torch.zeros(uq.numel(), dtype=torch.long, device=dev).index_add_(0, inv, dl)
            packed = ((tot <= chunk) & (size >= 2))[inv]                 # this row travels in a multi-row item
            first = torch.ones(m, dtype=torch.bool, device=dev)
            first[1:] = inv[1:] != inv[:-1]
            # cumulative edge counts inside the quadruple
            csum = torch.cumsum(dl, 0)
            base = (csum - dl)[first][inv]                                # edges before the quadruple
            cum_after = csum - base                                       # cumulative count after this row
            pos_in = torch.arange(m, device=dev) - first.nonzero().flatten()[inv]
            v = torch.zeros(uq.numel(), dtype=torch.long, device=dev)
            for q in range(3):
                sel = packed & (pos_in == q)
                v.index_add_(0, inv[sel], cum_after[sel] << (7 * q))
            v += (size - 1) << 21
            # one item per packed quadruple (at its first row), one per unpacked row
            is_item = (~packed) | first
            item_of_row = torch.cumsum(is_item.long(), 0) - 1
            fi = first & packed
            rows4 = single[is_item].clone()
            last_end = torch.zeros(uq.numel(), dtype=torch.long, device=dev).scatter_reduce(0, inv, r_end[rl], 'amax', include_self=True)
            pk = item_of_row[fi]
            rows4[pk, 2] = last_end[inv[fi]]
            rows4[pk, 3] = -(16 + v[inv[fi]])
            return rows4, item_of_row

        pad_item = torch.tensor([-1, 0, 0, -1], device=dev)
        parts, bounds, total = [], [0], 0
        n_multi = 0
        for k in range(8):
            lo, hi = lim[k], lim[k + 1]
            pk = pieces[lo:hi]
            big = (pk > 4).nonzero().flatten() + lo
            if big.numel():
                big = big[torch.argsort(deg[big], descending=True, stable=True)]
                parts.append(members(big))
                total += 4 * int(big.numel())
            rest = ((pk <= 4)).nonzero().flatten() + lo         # light rows and 2..4-piece hubs, in row order
            if rest.numel():
                is_hub = pieces[rest] > 1
                lt = (~is_hub).nonzero().flatten()
                l_items, item_of_light = light_items(rest[lt])
                n_multi += int((l_items[:, 3] <= -16).sum()) if l_items.numel() else 0
                # item slots per position of `rest`: a light row that opens an item 1, one that rides in a multi-row item 0,
                # a hub group 4 + its alignment padding
                opens = torch.zeros(rest.numel(), dtype=torch.long, device=dev)
                if lt.numel():
                    op = torch.ones(lt.numel(), dtype=torch.bool, device=dev)
                    op[1:] = item_of_light[1:] != item_of_light[:-1]
                    opens[lt] = op.long()
                light_run = torch.cumsum(opens, 0)
                hub_pos = is_hub.nonzero().flatten()
                prev = torch.cat([light_run.new_zeros(1), light_run[hub_pos][:-1]]) if hub_pos.numel() else light_run.new_zeros(0)
                pad = (-(light_run[hub_pos] - prev)) % 4
                slots = opens.clone()
                slots[hub_pos] = 4 + pad
                off = torch.cumsum(slots, 0) - slots
                m = int(slots.sum())
                blk = pad_item.expand(m, 4).clone()
                if lt.numel():
                    opening = lt[op]
                    blk[off[opening]] = l_items
                if hub_pos.numel():
                    first = off[hub_pos] + pad
                    blk[(first[:, None] + q4[None, :]).reshape(-1)] = members(rest[hub_pos])
                parts.append(blk)
                total += m
            tail = (-total) % 4
            if tail:
                parts.append(pad_item.expand(tail, 4))
                total += tail
            bounds.append(total)
        items = torch.cat(parts, 0).to(torch.int32).contiguous()
        # the kernels' barrier invariant (include/gnndelete_hip.h): limits and group starts at multiples of 4
        assert all(b % 4 == 0 for b in bounds) and total % 4 == 0, bounds
        hit = (items, total, torch.tensor(bounds, dtype=torch.int32, device=dev))
        self._onepass[key] = hit
        self.n_multirow_items = n_multi
        return hit

    def scratch_flat(self, tag, n_floats, device):
        """Named flat work buffers (e.g. the GAT kernels' merge scratch), kept per plan."""
        buf = self._scratch.get(tag)
        if buf is None or buf.numel() < n_floats:
            buf = torch.empty(max(int(n_floats), 4), dtype=torch.float32, device=device)
            self._scratch[tag] = buf
        return buf

    def scratch(self, d, device):
        if self.n_slots == 0:
            return None
        buf = self._scratch.get(d)
        if buf is None:
            buf = torch.empty(self.n_slots, d, dtype=torch.float32, device=device)
            self._scratch[d] = buf
        return buf


class CSRGraph:
    """Target-major CSR + its transpose (source-major) for the backward pass."""

    def __init__(self, n, rowptr, col, val, rowptr_t, col_t, val_t, perm_t, mode):
        self.n, self.mode = n, mode
        self.rowptr, self.col, self.val = rowptr, col, val
        self.rowptr_t, self.col_t, self.val_t, self.perm_t = rowptr_t, col_t, val_t, perm_t
        self.nnz = int(col.shape[0])
        self.plan = SplitPlan(rowptr)
        self.plan_t = SplitPlan(rowptr_t)

    @property
    def device(self):
        return self.col.device


def csr_from_coo(src, dst, n):
    """gd_csr_from_coo on device tensors: -> (rowptr int32 [n+1], col int32 [E], order int32 [E])."""
    src, dst = src.long().contiguous(), dst.long().contiguous()
    dev, e = src.device, int(src.numel())
    L = _lib.lib()
    nbytes = int(L.gd_csr_from_coo_workspace(n, e))
    if nbytes < 0:
        raise ValueError('graph too large for int32 CSR indices')
    rowptr = torch.empty(n + 1, dtype=torch.int32, device=dev)
    col = torch.empty(e, dtype=torch.int32, device=dev)
    order = torch.empty(e, dtype=torch.int32, device=dev)
    status = torch.empty(1, dtype=torch.int32, device=dev)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    _lib.check(L.gd_csr_from_coo(_lib.ptr(src), _lib.ptr(dst), e, n, _lib.ptr(rowptr), _lib.ptr(col), _lib.ptr(order),
                                 _lib.ptr(status), _lib.ptr(ws), nbytes, _lib.stream_ptr(dev)), 'gd_csr_from_coo')
    if int(status):
        raise IndexError(f'edge_index has endpoints outside [0, {n})')
    return rowptr, col, order


def _sorted_csr(src, dst, n):
    if src.is_cuda:
        rowptr, col, order = csr_from_coo(src, dst, n)
        return rowptr, col, order.long()
    # host-side layout only (CPU tests of the partition plan); no compute runs on these
    key = dst * n + src
    order = torch.argsort(key)
    counts = torch.bincount(dst, minlength=n)
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=src.device)
    rowptr[1:] = torch.cumsum(counts, 0)
    return rowptr.to(torch.int32), src[order].to(torch.int32), order


def build_csr(edge_index, num_nodes, mode='gcn'):
    assert mode in ('gcn', 'gat', 'sum', 'mean')
    assert edge_index.dim() == 2 and edge_index.shape[0] == 2
    dev = edge_index.device
    n = int(num_nodes)
    src, dst = edge_index[0].long(), edge_index[1].long()
    if mode in ('gcn', 'gat'):
        keep = src != dst
        loops = torch.arange(n, device=dev)
        src = torch.cat([src[keep], loops])
        dst = torch.cat([dst[keep], loops])
    if src.numel() >= 2 ** 31 or n >= 2 ** 31:
        raise ValueError('graph too large for int32 CSR indices')
    rowptr, col, order = _sorted_csr(src, dst, n)
    rowptr_t, col_t, order_t = _sorted_csr(dst, src, n)
    # position of every edge in the forward CSR, looked up in transposed order
    pos_fwd = torch.empty_like(order)
    pos_fwd[order] = torch.arange(order.numel(), device=dev)
    perm_t = pos_fwd[order_t].to(torch.int32)
    val = val_t = None
    if mode == 'gcn':
        val = torch.empty(col.shape[0], dtype=torch.float32, device=dev)
        if dev.type != 'cuda':
            raise _lib.GnnDeleteHipError('build_csr(mode="gcn") needs a GPU tensor: gcn_norm runs in the HIP library')
        _lib.check(_lib.lib().gd_gcn_norm_f32(rowptr.data_ptr(), col.data_ptr(), n, val.data_ptr(),
                                              _lib.stream_ptr(dev)), 'gd_gcn_norm_f32')
        val_t = val[perm_t.long()]
    elif mode == 'mean':
        # 1 / in-degree on every in-edge (SAGE mean aggregation); rows without in-edges stay empty
        deg = (rowptr[1:] - rowptr[:-1]).to(torch.float32).clamp(min=1.0)
        rows = torch.repeat_interleave(torch.arange(n, device=dev), (rowptr[1:] - rowptr[:-1]).long())
        val = (1.0 / deg)[rows]
        val_t = val[perm_t.long()]
    return CSRGraph(n, rowptr, col, val, rowptr_t, col_t, val_t, perm_t, mode)


class _GraphCache:
    """Small identity-keyed cache: the trainer passes the same edge_index tensor every epoch.
    Entries hold a reference to the key tensor, so its storage cannot be recycled for a
    different edge list while the entry is alive."""

    def __init__(self, capacity=8):
        self.capacity = capacity
        self.entries = []

    def get(self, edge_index, num_nodes, mode):
        for i, (t, ver, n, m, g) in enumerate(self.entries):
            if t is edge_index and ver == edge_index._version and n == num_nodes and m == mode:
                if i:
                    self.entries.insert(0, self.entries.pop(i))
                return g
        g = build_csr(edge_index, num_nodes, mode)
        self.entries.insert(0, (edge_index, edge_index._version, num_nodes, mode, g))
        del self.entries[self.capacity:]
        return g

    def clear(self):
        self.entries.clear()


CACHE = _GraphCache()


def graph_for(edge_index, num_nodes, mode):
    return CACHE.get(edge_index, num_nodes, mode)


def build_typed_csr(edge_index, edge_type, num_nodes, num_relations):
    """Relation-major CSR for R-GCN: virtual row r * n + i holds the in-edges of type r into i.
    Returns (rowptr[R*n+1], col, rowptr_t, col_t, inv_count_t) where the transposed arrays give,
    for every SOURCE node j, its out-edges as virtual rows (r * n + i) together with
    1/|N_r(i)| - what the backward (dx_j = sum_e dy[r, i] / cnt) gathers."""
    dev = edge_index.device
    n, r = int(num_nodes), int(num_relations)
    if r * n >= 2 ** 31:
        raise ValueError('num_relations * num_nodes overflows int32')
    src, dst, et = edge_index[0].long(), edge_index[1].long(), edge_type.long()
    vrow = et * n + dst
    order = torch.argsort(vrow * n + src)
    counts = torch.bincount(vrow, minlength=r * n)
    rowptr = torch.zeros(r * n + 1, dtype=torch.int64, device=dev)
    rowptr[1:] = torch.cumsum(counts, 0)
    col = src[order].to(torch.int32)
    order_t = torch.argsort(src * (r * n) + vrow)
    counts_t = torch.bincount(src, minlength=n)
    rowptr_t = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    rowptr_t[1:] = torch.cumsum(counts_t, 0)
    col_t = vrow[order_t].to(torch.int32)
    inv_t = (1.0 / counts.clamp(min=1).to(torch.float32))[vrow[order_t]]
    return rowptr.to(torch.int32), col, rowptr_t.to(torch.int32), col_t, inv_t


class TypedNodeCSR:
    """Node-major typed graph for the fused R-GCN conv (gd_rgcn_conv_f32): the in-edges of a node are
    sorted by relation and grouped into (node, relation) runs.  `fwd` aggregates sources into targets
    with w = 1 / |run| (PyG RGCNConv aggr='mean' per relation); `bwd` is the transposed graph whose
    edges carry the weight of the forward run they belong to (input gradient)."""

    def __init__(self, edge_index, edge_type, num_nodes, num_relations, row_range=None):
        """row_range = (lo, hi): a rank's share of a 1-D row partition - `fwd` keeps the in-edges of the TARGET rows
        lo <= i < hi (their (node, relation) runs are complete, so the mean weights are the global ones), `bwd` the
        out-edges of the SOURCE rows in the range, carrying the weights of the forward runs of the WHOLE graph."""
        n, r = int(num_nodes), int(num_relations)
        src, dst, et = edge_index[0].long(), edge_index[1].long(), edge_type.long()
        if src.numel() >= 2 ** 31:
            raise ValueError('graph too large for int32 CSR indices')
        self.n, self.num_relations = n, r
        run_f = dst * r + et
        order = torch.argsort(run_f * n + src)
        self.fwd_order = order                                     # fwd edge k = input edge fwd_order[k]
        self.fwd, w_sorted = self._runs(run_f[order], src[order], r, n, None)
        w_edge = torch.empty_like(w_sorted)
        w_edge[order] = w_sorted                                   # weight of every edge in input order
        if row_range is not None:
            lo, hi = row_range
            keep = (dst[order] >= lo) & (dst[order] < hi)
            self.fwd_order = order[keep]
            self.fwd, _ = self._runs(run_f[order][keep], src[order][keep], r, n, w_sorted[keep])
            mine = (src >= lo) & (src < hi)
            src, dst, et, w_edge = src[mine], dst[mine], et[mine], w_edge[mine]
        run_b = src * r + et
        order_b = torch.argsort(run_b * n + dst)
        self.bwd, _ = self._runs(run_b[order_b], dst[order_b], r, n, w_edge[order_b])
        self.bwd_order = order_b if row_range is None else None    # bwd edge k = input edge bwd_order[k] (whole graphs only)

    def restrict_bwd(self, node_mask):
        """A view of this graph whose TRANSPOSED side keeps only the out-edges of the source nodes in node_mask (bool [n]): the
        input gradient of a typed conv is then formed for those rows only (rows outside get nothing added).  The engine's
        affected-rows option uses it for conv2's input gradient, which only the Del-1 rows read (S1: 16 % of the nodes of the
        biokg request).  The forward side is shared; plans are cached per view."""
        node_ptr, seg_ptr, seg_rel, col, w = self.bwd
        dev = col.device
        runs_per_node = (node_ptr[1:] - node_ptr[:-1]).long()
        node_of_run = torch.repeat_interleave(torch.arange(self.n, device=dev), runs_per_node)
        keep_run = node_mask.to(dev)[node_of_run]
        seg_len = (seg_ptr[1:] - seg_ptr[:-1]).long()
        keep_edge = torch.repeat_interleave(keep_run, seg_len)
        new_len = seg_len[keep_run]
        new_seg_ptr = torch.zeros(int(new_len.numel()) + 1, dtype=torch.int64, device=dev)
        new_seg_ptr[1:] = torch.cumsum(new_len, 0)
        new_node_ptr = torch.zeros(self.n + 1, dtype=torch.int64, device=dev)
        new_node_ptr[1:] = torch.cumsum(torch.where(node_mask.to(dev), runs_per_node, torch.zeros_like(runs_per_node)), 0)
        i32 = lambda t: t.to(torch.int32).contiguous()
        view = object.__new__(TypedNodeCSR)
        view.n, view.num_relations = self.n, self.num_relations
        view.fwd, view.fwd_order, view.bwd_order = self.fwd, self.fwd_order, None
        view.bwd = (i32(new_node_ptr), i32(new_seg_ptr), seg_rel[keep_run].contiguous(), col[keep_edge].contiguous(), w[keep_edge].contiguous())
        return view

    def rel_major(self):
        """The forward edges in RELATION-major order for the gradients of trainable relation weights (gd_typed_wgrad_f32):
        -> dict(rel_ptr int32 [R + 1], src, dst int32 [E], w float32 [E] (the mean weights 1 / |N_r(i)|), from_fwd int64 [E]:
        position in the `fwd` arrays of every entry - per-edge coefficients given in fwd order are permuted with it), built
        once on the device (a stable sort by relation of the node-major arrays: inside a relation the edges keep their
        (target, source) order, so the sums are added in a fixed order)."""
        c = self.__dict__.get('_rel_major')
        if c is None:
            node_ptr, seg_ptr, seg_rel, col, w = self.fwd
            dev = col.device
            seg_len = (seg_ptr[1:] - seg_ptr[:-1]).long()
            runs_per_node = (node_ptr[1:] - node_ptr[:-1]).long()
            node_of_run = torch.repeat_interleave(torch.arange(self.n, device=dev), runs_per_node)
            rel_e = torch.repeat_interleave(seg_rel.long(), seg_len)
            dst_e = torch.repeat_interleave(node_of_run, seg_len)
            order = torch.argsort(rel_e, stable=True)
            rel_ptr = torch.zeros(self.num_relations + 1, dtype=torch.int64, device=dev)
            rel_ptr[1:] = torch.cumsum(torch.bincount(rel_e, minlength=self.num_relations), 0)
            i32 = lambda t: t.to(torch.int32).contiguous()
            c = dict(rel_ptr=i32(rel_ptr), src=i32(col.long()[order]), dst=i32(dst_e[order]), w=w[order].contiguous(), from_fwd=order)
            self.__dict__['_rel_major'] = c
        return c

    def fwd_edges(self):
        """(src, dst, rel) int32 [E] of the forward edges in `fwd` order (gd_typed_edge_dot_f32)."""
        c = self.__dict__.get('_fwd_edges')
        if c is None:
            node_ptr, seg_ptr, seg_rel, col, _ = self.fwd
            dev = col.device
            seg_len = (seg_ptr[1:] - seg_ptr[:-1]).long()
            runs_per_node = (node_ptr[1:] - node_ptr[:-1]).long()
            node_of_run = torch.repeat_interleave(torch.arange(self.n, device=dev), runs_per_node)
            i32 = lambda t: t.to(torch.int32).contiguous()
            c = (col, i32(torch.repeat_interleave(node_of_run, seg_len)), i32(torch.repeat_interleave(seg_rel.long(), seg_len)))
            self.__dict__['_fwd_edges'] = c
        return c

    def tile_plan(self, trans=False):
        """The (64-node tile, relation) regrouping gd_rgcn_tile_conv_f32 walks (include/gnndelete_hip.h): runs cut into
        pieces of <= 16 edges, piece k of a run in pass k; steps = distinct (tile, relation, pass); edges re-sorted into
        (step, row, source) order.  Built once per direction with sorts / uniques on the device."""
        key = 'bwd' if trans else 'fwd'
        cache = self.__dict__.setdefault('_tile_plans', {})
        if key not in cache:
            cache[key] = self._build_tile_plan(self.bwd if trans else self.fwd, self.n, self.num_relations)
        return cache[key]

    @staticmethod
    def _build_tile_plan(arrays, n, r, cap=16, tile=64, hub_steps=None, max_pieces=32):
        """hub_steps: a node whose runs make more than this many pieces-in-sequence (sum over its relations of
        ceil(|run| / cap)) is spread over V = 2^k <= 64 SLICE rows of extra tiles, piece k of a run to slice k mod V, so
        that a slice walks about hub_steps / 2 steps."""
        node_ptr, seg_ptr, seg_rel, col, w = arrays
        dev = col.device
        if hub_steps is None:
            hub_steps = int(os.environ.get('GD_RGCN_HUB_STEPS', 128))
        n_real = (n + tile - 1) // tile
        i32 = lambda t: t.to(torch.int32).contiguous()
        e = int(col.numel())
        z = torch.zeros(1, dtype=torch.int32, device=dev)
        if e == 0:
            return dict(n_tiles=n_real, tile_order=i32(torch.arange(n_real, device=dev)),
                        tile_step_ptr=torch.zeros(n_real + 1, dtype=torch.int32, device=dev), step_rel=z, step_piece_ptr=z,
                        piece=torch.zeros(1, 2, dtype=torch.int32, device=dev),
                        col=z, w=torch.zeros(1, dtype=torch.float32, device=dev), n_steps=0, n_pieces=0, n_hubs=0,
                        hub_node=z, hub_ptr=z, n_slice_rows=0)
        seg_len = (seg_ptr[1:] - seg_ptr[:-1]).long()
        n_runs = int(seg_len.numel())
        run_of_edge = torch.repeat_interleave(torch.arange(n_runs, device=dev), seg_len)
        runs_per_node = (node_ptr[1:] - node_ptr[:-1]).long()
        node_of_run = torch.repeat_interleave(torch.arange(n, device=dev), runs_per_node)
        pos = torch.arange(e, device=dev) - seg_ptr.long()[run_of_edge]
        pc = pos // cap                                                    # piece of the run this edge belongs to
        node = node_of_run[run_of_edge]
        rel = seg_rel.long()[run_of_edge]
        # hubs -> slice rows
        node_steps = torch.zeros(n, dtype=torch.int64, device=dev).scatter_add_(0, node_of_run, (seg_len + cap - 1) // cap)
        hub_node = (node_steps > hub_steps).nonzero().flatten()
        n_hubs = int(hub_node.numel())
        n_pad = n_real * tile
        row_id, passn = node, pc
        n_slice = 0
        hub_ptr = torch.zeros(n_hubs + 1, dtype=torch.int64, device=dev)
        if n_hubs:
            want = (2 * node_steps[hub_node] + hub_steps - 1) // hub_steps      # slices so that each walks ~hub_steps / 2
            v = torch.ones_like(want)
            while bool((v < want).any()):
                v = torch.where(v < want, v * 2, v)
            v = v.clamp(max=tile)
            hub_ptr[1:] = torch.cumsum(v, 0)
            n_slice = int(hub_ptr[-1])
            hub_of_node = torch.full((n,), -1, dtype=torch.int64, device=dev)
            hub_of_node[hub_node] = torch.arange(n_hubs, device=dev)
            h = hub_of_node[node]
            is_hub = h >= 0
            hv = v[h.clamp(min=0)]
            row_id = torch.where(is_hub, n_pad + hub_ptr[h.clamp(min=0)] + pc % hv, node)
            passn = torch.where(is_hub, pc // hv, pc)
        n_tiles = n_real + (n_slice + tile - 1) // tile
        passes = int(passn.max()) + 1
        pkey = (((row_id // tile) * r + rel) * passes + passn) * tile + row_id % tile     # (tile, rel, pass, row)
        order = torch.argsort(pkey, stable=True)
        piece_key, piece_len = torch.unique_consecutive(pkey[order], return_counts=True)
        piece_e0 = torch.cumsum(piece_len, 0) - piece_len
        piece_row = piece_key % tile
        # a step holds at most `max_pieces` pieces (the kernel gives each of its 8 waves 4): the rest of a crowded
        # (tile, relation, pass) goes to further steps of the same relation
        base_key, base_np = torch.unique_consecutive(piece_key // tile, return_counts=True)
        base_first = torch.cumsum(base_np, 0) - base_np
        base_of_piece = torch.repeat_interleave(torch.arange(base_key.numel(), device=dev), base_np)
        if os.environ.get('GD_RGCN_SORT_PIECES', '1') == '1':
            # inside a (tile, relation, pass) the pieces go longest first: the kernel gives a wave the pieces w, w + 8 (first
            # round, its two lane groups side by side) and w + 16, w + 24 (second round), a round lasts as long as its longer
            # piece, and the step's barrier waits for the slowest wave - sorted, the long pieces sit side by side in the first
            # round of different waves and the second round holds the short ones.  (The pieces of a step belong to distinct
            # rows: their order does not touch any sum.)
            by_len = torch.argsort(base_of_piece * (cap + 1) + (cap - piece_len), stable=True)
            piece_e0, piece_row, piece_len = piece_e0[by_len], piece_row[by_len], piece_len[by_len]
        sub = (torch.arange(piece_key.numel(), device=dev) - base_first[base_of_piece]) // max_pieces
        n_sub = int(sub.max()) + 1
        step_key, step_np = torch.unique_consecutive(base_key[base_of_piece] * n_sub + sub, return_counts=True)
        step_key = step_key // n_sub
        n_steps = int(step_key.numel())
        step_piece_ptr = torch.zeros(n_steps + 1, dtype=torch.int64, device=dev)
        step_piece_ptr[1:] = torch.cumsum(step_np, 0)
        steps_per_tile = torch.bincount(step_key // (passes * r), minlength=n_tiles)
        tile_step_ptr = torch.zeros(n_tiles + 1, dtype=torch.int64, device=dev)
        tile_step_ptr[1:] = torch.cumsum(steps_per_tile, 0)
        piece = torch.stack([piece_e0, piece_row | (piece_len << 8)], 1)
        return dict(n_tiles=n_tiles, tile_order=i32(torch.argsort(steps_per_tile, descending=True, stable=True)),
                    tile_step_ptr=i32(tile_step_ptr), step_rel=i32((step_key // passes) % r), step_piece_ptr=i32(step_piece_ptr),
                    piece=i32(piece), col=col[order].contiguous(), w=w[order].contiguous(),
                    n_steps=n_steps, n_pieces=int(piece_key.numel()), n_hubs=n_hubs, hub_node=i32(hub_node) if n_hubs else z,
                    hub_ptr=i32(hub_ptr), n_slice_rows=(n_tiles - n_real) * tile, max_steps=int(steps_per_tile.max()))

    def wave_plan(self, trans=False):
        """The (tile, relation) UNITS gd_rgcn_wave_conv_f32 walks (include/gnndelete_hip.h): every (node, relation) run cut
        into pieces of <= 4 edges, the pieces of a (64-node tile, relation) laid 16 to a unit in a fixed shape.  Built once
        per direction with sorts / uniques on the device."""
        key = 'bwd' if trans else 'fwd'
        cache = self.__dict__.setdefault('_wave_plans', {})
        if key not in cache:
            cache[key] = self._build_wave_plan(self.bwd if trans else self.fwd, self.n, self.num_relations)
        return cache[key]

    @staticmethod
    def _build_wave_plan(arrays, n, r, tile=64, cap=4, slots=16):
        """-> tile_unit_ptr [T + 1], n_units = U, unit_rel [U + 1], unit_row [U + 1, 16] (slot word: node % tile | same-row
        flags | last; 0 for an unused slot), unit_edges [U + 1, 16, 4, 2] int32 = (source, weight bits) pairs, unused pairs =
        (n, 0.0): one row past x, for which the kernel's buffer loads return zeros without a memory access; unit U is EMPTY
        (what the kernel's pipeline runs on past a tile's end); unit_rel = relation | scan steps any slot of the unit needs << 16;
        job_tile [T] = tiles by unit count, heaviest first."""
        node_ptr, seg_ptr, seg_rel, col, w = arrays
        dev = col.device
        assert r < 65536, 'the unit plan keeps relation | scan steps << 16 in one word'
        n_tiles = (n + tile - 1) // tile
        i32 = lambda t: t.to(torch.int32).contiguous()
        e = int(col.numel())
        if e == 0:
            empty = torch.zeros(1, slots, cap, 2, dtype=torch.int32, device=dev)
            empty[..., 0] = n
            return dict(n_tiles=n_tiles, tile=tile, tile_unit_ptr=torch.zeros(n_tiles + 1, dtype=torch.int32, device=dev),
                        unit_rel=torch.zeros(1, dtype=torch.int32, device=dev), unit_row=torch.zeros(1, slots, dtype=torch.int32, device=dev),
                        unit_edges=empty, job_tile=i32(torch.arange(n_tiles, device=dev)), n_units=0, n_pieces=0, max_units=0)
        seg_len = (seg_ptr[1:] - seg_ptr[:-1]).long()
        n_runs = int(seg_len.numel())
        run_of_edge = torch.repeat_interleave(torch.arange(n_runs, device=dev), seg_len)
        runs_per_node = (node_ptr[1:] - node_ptr[:-1]).long()
        node_of_run = torch.repeat_interleave(torch.arange(n, device=dev), runs_per_node)
        pos = torch.arange(e, device=dev) - seg_ptr.long()[run_of_edge]          # position inside the run
        # pieces, in (tile, relation, node, piece) order: the node-major arrays are (node, relation, source) already
        piece_start = pos % cap == 0
        piece_of_edge = torch.cumsum(piece_start, 0) - 1
        p_run = run_of_edge[piece_start]
        p_node, p_rel = node_of_run[p_run], seg_rel.long()[p_run]
        gkey = (p_node // tile) * r + p_rel                                     # group = (tile, relation)
        order = torch.argsort(gkey, stable=True)                                # keeps (node, piece) order inside a group
        rank = torch.empty_like(order)
        rank[order] = torch.arange(order.numel(), device=dev)                   # piece -> its place in group order
        gk_sorted = gkey[order]
        groups, g_np = torch.unique_consecutive(gk_sorted, return_counts=True)
        g_first = torch.cumsum(g_np, 0) - g_np
        g_units = (g_np + slots - 1) // slots
        g_tile = groups // r
        units_per_tile = torch.zeros(n_tiles, dtype=torch.int64, device=dev).scatter_add_(0, g_tile, g_units)
        padded = units_per_tile                                                 # (no per-tile padding: the kernel runs on the empty unit)
        tile_unit_ptr = torch.zeros(n_tiles + 1, dtype=torch.int64, device=dev)
        tile_unit_ptr[1:] = torch.cumsum(padded, 0)
        n_units = int(tile_unit_ptr[-1])
        g_excl = torch.cumsum(g_units, 0) - g_units                             # units before the group, unpadded
        tile_excl = torch.cumsum(units_per_tile, 0) - units_per_tile
        g_unit0 = tile_unit_ptr[:-1][g_tile] + (g_excl - tile_excl[g_tile])     # first unit of the group
        unit_rel = torch.zeros(n_units + 1, dtype=torch.int64, device=dev)
        unit_of_g = torch.repeat_interleave(torch.arange(groups.numel(), device=dev), g_units)
        u_idx = g_unit0[unit_of_g] + (torch.arange(unit_of_g.numel(), device=dev) - g_excl[unit_of_g])
        unit_rel[u_idx] = groups[unit_of_g] % r
        # piece -> (unit, slot)
        group_of_sorted = torch.repeat_interleave(torch.arange(groups.numel(), device=dev), g_np)
        in_group = torch.arange(order.numel(), device=dev) - g_first[group_of_sorted]
        unit_sorted = g_unit0[group_of_sorted] + in_group // slots
        slot_sorted = in_group % slots
        # slot word = node % tile | flags << 8 | last << 12.  The slots of one node row inside a unit are consecutive
        # ((node, piece) order); flag bit b says the slot 2^b to the left holds the same row (the kernel's segmented scan
        # over the 16 slots), `last` marks the slot that ends up with the row's total and adds it to the accumulator
        run_sorted = p_run[order]
        word = p_node[order] % tile
        for b in range(4):
            sh = 1 << b
            same = torch.zeros_like(word, dtype=torch.bool)
            same[sh:] = (run_sorted[sh:] == run_sorted[:-sh]) & (slot_sorted[sh:] >= sh)
            word = word | (same.long() << (8 + b))
        last = torch.ones_like(word, dtype=torch.bool)
        last[:-1] = (run_sorted[1:] != run_sorted[:-1]) | (unit_sorted[1:] != unit_sorted[:-1])
        word = word | (last.long() << 12)
        unit_row = torch.zeros((n_units + 1) * slots, dtype=torch.int64, device=dev)
        unit_row[unit_sorted * slots + slot_sorted] = word
        # which scan steps any slot of a unit needs: bits 16 .. 19 of the unit's relation word
        step_bits = torch.zeros(n_units + 1, dtype=torch.int64, device=dev)
        for b in range(4):
            has = torch.zeros(n_units + 1, dtype=torch.int64, device=dev).scatter_reduce_(0, unit_sorted, (word >> (8 + b)) & 1, 'amax', include_self=True)
            step_bits |= has << b
        unit_rel = unit_rel | (step_bits << 16)
        p_unit, p_slot = unit_sorted[rank], slot_sorted[rank]                   # back in piece order
        unit_edges = torch.zeros((n_units + 1) * slots * cap, 2, dtype=torch.int32, device=dev)
        unit_edges[:, 0] = n
        at = (p_unit[piece_of_edge] * slots + p_slot[piece_of_edge]) * cap + pos % cap
        unit_edges[at, 0] = col.to(torch.int32)
        unit_edges[at, 1] = w.to(torch.float32).view(torch.int32)
        return dict(n_tiles=n_tiles, tile=tile, tile_unit_ptr=i32(tile_unit_ptr), unit_rel=i32(unit_rel),
                    unit_row=i32(unit_row).view(-1, slots), unit_edges=unit_edges.view(-1, slots, cap, 2).contiguous(),
                    job_tile=i32(torch.argsort(padded, descending=True, stable=True)), n_units=n_units,
                    n_pieces=int(order.numel()), max_units=int(padded.max()))

    @staticmethod
    def _runs(run_sorted, col_sorted, r, n, w):
        dev = run_sorted.device
        runs, counts = torch.unique_consecutive(run_sorted, return_counts=True)
        seg_ptr = torch.zeros(runs.numel() + 1, dtype=torch.int64, device=dev)
        seg_ptr[1:] = torch.cumsum(counts, 0)
        node_ptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        node_ptr[1:] = torch.cumsum(torch.bincount(runs // r, minlength=n), 0)
        if w is None:
            w = torch.repeat_interleave(1.0 / counts.to(torch.float32), counts)
        i32 = lambda t: t.to(torch.int32).contiguous()
        return (i32(node_ptr), i32(seg_ptr), i32(runs % r), i32(col_sorted), w.to(torch.float32).contiguous()), w
