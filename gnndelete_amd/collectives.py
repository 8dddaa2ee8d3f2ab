"""The two exchange steps of the row-partitioned Del-training step, on torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU node, "gloo" in the CPU tests):

  all_gather_rows   every rank owns one equal-sized block of rows of an [n_pad, d] matrix and
                    needs all of it for the next SpMM (t2 forward, dz2 backward);
  exchange_rows     the sparse form of the same exchange: every rank sends each peer only the
                    rows that peer's SpMM actually gathers (all-to-all with per-pair row lists);
  all_reduce_sum    one packed fp32 buffer per step: the partial Del-weight gradients
                    (128^2 + 128^2 + 64^2 floats) and the four loss sums.
"""
import torch
import torch.distributed as dist


def row_blocks(n, world):
    """Equal row blocks: (chunk, n_pad); rank r owns rows [r*chunk, min(n, (r+1)*chunk))."""
    chunk = (n + world - 1) // world
    return chunk, chunk * world


def all_gather_rows(full, rank, world, chunk, group=None):
    """In-place all-gather of the [chunk, d] row blocks of `full` ([world*chunk, d]); the
    caller has written block `rank`."""
    if world == 1:
        return
    mine = full[rank * chunk:(rank + 1) * chunk]
    try:
        dist.all_gather_into_tensor(full, mine.clone(), group=group)
    except (RuntimeError, NotImplementedError):
        dist.all_gather([full[r * chunk:(r + 1) * chunk] for r in range(world)], mine.clone(), group=group)


def all_reduce_sum(buf, world, group=None):
    if world > 1:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)


def halo_lists(rowptr, col, n, rank, world, chunk):
    """Per-pair row lists of a 1-D row partition: rank q's SpMM over its rows [q*chunk, ..) gathers
    the source rows unique(col[rowptr[lo_q]:rowptr[hi_q]]); the part of that set owned by another
    rank p is what p must send to q.  Every rank holds the whole (small) graph structure, so both
    sides of every pair derive the same sorted list without communicating.
    -> (send_rows, in_splits, recv_rows, out_splits) for `rank` (global row ids, int64)."""
    dev = col.device
    send, recv = [[] for _ in range(world)], [[] for _ in range(world)]
    for q in range(world):
        lo_q, hi_q = min(n, q * chunk), min(n, (q + 1) * chunk)
        a, b = int(rowptr[lo_q]), int(rowptr[hi_q])
        needed = torch.unique(col[a:b].long())
        owner = needed // chunk
        if q == rank:
            for p in range(world):
                if p != rank:
                    recv[p] = needed[owner == p]
        else:
            send[q] = needed[owner == rank]
    empty = torch.empty(0, dtype=torch.long, device=dev)
    send = [s if torch.is_tensor(s) else empty for s in send]
    recv = [r if torch.is_tensor(r) else empty for r in recv]
    return (torch.cat(send), [int(s.numel()) for s in send], torch.cat(recv), [int(r.numel()) for r in recv])


def exchange_rows(send_buf, recv_buf, in_splits, out_splits, world, group=None):
    """recv_buf <- all-to-all of send_buf along dim 0 (rows per peer: in_splits sent, out_splits
    received).  Backends without device all-to-all (gloo + GPU tensors in the tests) stage
    through the host."""
    if world == 1:
        return
    try:
        dist.all_to_all_single(recv_buf, send_buf, out_splits, in_splits, group=group)
    except (RuntimeError, NotImplementedError):
        host = torch.empty(recv_buf.shape, dtype=recv_buf.dtype)
        dist.all_to_all_single(host, send_buf.cpu(), out_splits, in_splits, group=group)
        recv_buf.copy_(host)
