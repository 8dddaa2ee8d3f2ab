"""The two exchange steps of the row-partitioned Del-training step, on torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU node, "gloo" in the CPU tests):

  all_gather_rows   every rank owns one equal-sized block of rows of an [n_pad, d] matrix and
                    needs all of it for the next SpMM (t2 forward, dz2 backward);
  halo_plan /       the sparse form of the same exchange: every rank sends each peer only the
  exchange_rows     rows that peer's SpMM actually gathers (all-to-all with per-pair row lists);
  all_reduce_sum    one packed fp32 buffer per step: the partial Del-weight gradients
                    (128^2 + 128^2 + 64^2 floats) and the four loss sums.
"""
import os

import torch
import torch.distributed as dist

# GD_FORCE_COLLECTIVES=1: issue the collectives even in a world of one - lets a one-GPU box drive the real RCCL calls
# (tests/test_dist_cpu.py::test_partitioned_engine_over_rccl_in_a_world_of_one)
_FORCE = os.environ.get('GD_FORCE_COLLECTIVES') == '1'


class DirectComm:
    """An RCCL communicator owned by the HIP library itself (include/gnndelete_hip.h: gd_comm_init, gd_allreduce_f32,
    gd_exchange_rows_f32) instead of torch.distributed's process group: the 128-byte unique id is made on rank 0 and
    broadcast over an existing (host-side, e.g. gloo) group, after which the data path never goes through torch's
    collectives.  Pass it as `group` to all_reduce_sum / exchange_rows / PartitionedNodeembEngine.  The calls only enqueue
    work on the current HIP stream (no host synchronisation)."""

    def __init__(self, rank, world, device, bootstrap_group=None):
        import ctypes
        from . import _lib
        self.rank, self.world, self.device = rank, world, torch.device(device)
        self._bootstrap = bootstrap_group
        uid = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            _lib.check(_lib.lib().gd_comm_unique_id(uid.data_ptr()), 'gd_comm_unique_id')
        if world > 1:
            dist.broadcast(uid, 0, group=bootstrap_group)
        comm = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().gd_comm_init(uid.data_ptr(), world, rank, ctypes.byref(comm)), 'gd_comm_init')
        self._comm = comm
        self._counts = {}

    def all_reduce(self, buf):
        from . import _lib
        if buf.dtype != torch.float32:
            # set-up values only (the fp64 constants of the folded losses, once per request): over the host-side
            # bootstrap group - the data path of a step is fp32 and goes through gd_allreduce_f32
            if self.world > 1:
                host = buf.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self._bootstrap)
                buf.copy_(host)
            return
        assert buf.is_cuda and buf.is_contiguous()
        _lib.check(_lib.lib().gd_allreduce_f32(buf.data_ptr(), buf.numel(), self._comm, _lib.stream_ptr(buf.device)),
                   'gd_allreduce_f32')

    def exchange(self, send, recv, in_splits, out_splits):
        """rows of `send` grouped by destination (in_splits[p] rows to peer p) -> rows of `recv` grouped by source."""
        from . import _lib
        key = (tuple(in_splits), tuple(out_splits))
        host = self._counts.get(key)
        if host is None:
            host = self._counts[key] = (torch.tensor(in_splits, dtype=torch.int64), torch.tensor(out_splits, dtype=torch.int64))
        assert send.is_contiguous() and recv.is_contiguous() and send.shape[1:] == recv.shape[1:]
        row = int(send[0].numel()) if send.shape[0] else (int(recv[0].numel()) if recv.shape[0] else 1)
        _lib.check(_lib.lib().gd_exchange_rows_f32(send.data_ptr() if send.numel() else None, host[0].data_ptr(),
                                                   recv.data_ptr() if recv.numel() else None, host[1].data_ptr(), row,
                                                   self.world, self._comm, _lib.stream_ptr(self.device)),
                   'gd_exchange_rows_f32')

    def close(self):
        from . import _lib
        if self._comm is not None:
            _lib.lib().gd_comm_destroy(self._comm)
            self._comm = None


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
    if world > 1 or _FORCE:
        if isinstance(group, DirectComm):
            group.all_reduce(buf)
        else:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)


class HaloPlan:
    """One rank's side of a sparse row exchange: send_rows (global ids, grouped by destination rank, sorted inside a
    group), in_splits (rows sent to each rank), recv_rows / out_splits likewise for what it receives."""

    def __init__(self, send_rows, in_splits, recv_rows, out_splits, pair_counts):
        self.send_rows, self.in_splits = send_rows, in_splits
        self.recv_rows, self.out_splits = recv_rows, out_splits
        self.n_send, self.n_recv = int(sum(in_splits)), int(sum(out_splits))
        self.pair_counts = pair_counts              # [world][world]: rows rank p sends to rank q at [q][p]


def halo_plan(rowptr, col, n, rank, world, chunk, row_mask=None):
    """Per-pair row lists of a 1-D row partition, for all ranks at once ON THE DEVICE: rank q aggregates the rows
    [q*chunk, ..) (only those with row_mask set, if given) and gathers the source rows col[k] of their CSR entries;
    a gathered row owned by another rank p is a row p sends to q.  The distinct (q, p, row) triples are one
    torch.unique over the cross-rank CSR entries; their order (q, then p, then row id) is the order both ends use,
    so no list is ever communicated - every rank holds the replicated graph structure.  One device->host transfer:
    the [world, world] count matrix (the split sizes all_to_all_single wants as Python ints)."""
    dev = col.device
    deg = (rowptr[1:] - rowptr[:-1]).long()
    tgt = torch.repeat_interleave(torch.arange(n, device=dev), deg)
    src = col.long()
    q, p = tgt // chunk, src // chunk
    cross = q != p
    if row_mask is not None:
        cross &= row_mask.to(dev)[tgt]
    key = torch.unique((q[cross] * world + p[cross]) * n + src[cross])
    pair, rows = key // n, key % n
    counts = torch.bincount(pair, minlength=world * world).view(world, world).tolist()      # the one host sync
    recv_sel = (pair // world) == rank                 # what this rank receives: ordered by sender p, then row
    send_sel = (pair % world) == rank                  # what it sends: ordered by receiver q, then row
    return HaloPlan(rows[send_sel].contiguous(), [counts[qq][rank] for qq in range(world)],
                    rows[recv_sel].contiguous(), [counts[rank][pp] for pp in range(world)], counts)


def exchange_rows_reverse(send_buf, recv_buf, plan, world, group=None):
    """The same exchange run backwards (gradient contributions for the rows a rank RECEIVED in the forward exchange
    travel back to their owners): send_buf rows are ordered like plan.recv_rows, recv_buf like plan.send_rows."""
    rev = HaloPlan(plan.recv_rows, plan.out_splits, plan.send_rows, plan.in_splits, plan.pair_counts)
    exchange_rows(send_buf, recv_buf, rev, world, group)


def exchange_rows(send_buf, recv_buf, plan, world, group=None):
    """recv_buf[:n_recv] <- all-to-all of send_buf[:n_send] along dim 0 with the plan's per-peer row counts.
    RCCL ("nccl") moves device buffers directly and any failure propagates - a silent detour through the host would
    turn a broken transport into a 100x slowdown.  gloo (the CPU-side test backend) has no device all-to-all: there,
    and only there, the buffers are staged through the host."""
    if world == 1 and not _FORCE:
        return
    send, recv = send_buf[:plan.n_send], recv_buf[:plan.n_recv]
    if isinstance(group, DirectComm):
        group.exchange(send, recv, plan.in_splits, plan.out_splits)
        return
    if dist.get_backend(group) == 'gloo' and send.is_cuda:
        host = torch.empty(recv.shape, dtype=recv.dtype)
        dist.all_to_all_single(host, send.cpu(), plan.out_splits, plan.in_splits, group=group)
        recv.copy_(host)
        return
    dist.all_to_all_single(recv, send, plan.out_splits, plan.in_splits, group=group)
