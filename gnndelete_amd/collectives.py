"""The two exchange steps of the row-partitioned Del-training step, on torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU node, "gloo" in the CPU tests):

  all_gather_rows   every rank owns one equal-sized block of rows of an [n_pad, d] matrix and
                    needs all of it for the next SpMM (t2 forward, dz2 backward);
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
