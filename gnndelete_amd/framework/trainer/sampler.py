"""GraphSAINT random-walk mini-batching (reference: GraphSAINTRandomWalkSampler from
torch_geometric.loader, used at gnndelete_nodeemb.py:379-381 with batch_size roots, walk_length 2,
num_steps batches per epoch) and the mini-batch unlearning loop of gnndelete_nodeemb.py:352-495.

Kept for iteration-level parity studies: upstream needs it because ogbl-* graphs do not fit its
GPUs; here the full graph is a single batch (NodeembEngine).  The sampler's random stream
cannot match PyG's (torch_sparse random_walk), so parity with upstream is statistical.

The sampler works where its data lives: on CPU tensors with torch ops (tests, injected node sets), on device tensors
with the HIP random-walk kernel (gd_random_walk) and device-side unique / compaction - the mini-batch loops hand it
the device copy, so a batch never touches the host (upstream samples on the CPU and uploads every batch)."""
import os

import torch
import torch.nn as nn

from ..graph_utils import negative_sampling
from ._log import wandb_log
from .base import _require_gpu, device


class RandomWalkSubgraphSampler:
    """Yields induced subgraphs over the nodes visited by `batch_size` random walks of length
    `walk_length`; node- and edge-sized attributes are sliced like PyG's saint_subgraph."""

    def __init__(self, data, batch_size, walk_length=2, num_steps=32, generator=None):
        self.data, self.batch_size, self.walk_length, self.num_steps = data, batch_size, walk_length, num_steps
        self.gen = generator
        ei = data.edge_index
        self.dev = ei.device
        self.n = int(data.num_nodes)
        order = torch.argsort(ei[0] * self.n + ei[1], stable=True)         # ties (multi-edges) in input order on every device
        self.src_sorted, self.dst_sorted, self.order = ei[0][order], ei[1][order], order
        counts = torch.bincount(self.src_sorted, minlength=self.n)
        self.rowptr = torch.zeros(self.n + 1, dtype=torch.long, device=self.dev)
        self.rowptr[1:] = torch.cumsum(counts, 0)
        if self.dev.type == 'cuda':                              # int32 CSR over source rows for the walk kernel
            self._rowptr32, self._col32 = self.rowptr.to(torch.int32), self.dst_sorted.to(torch.int32).contiguous()

    def __len__(self):
        return self.num_steps

    def _walk(self):
        if self.dev.type == 'cuda':
            return self._walk_device()
        cur = torch.randint(0, self.n, (self.batch_size,), generator=self.gen)
        visited = [cur]
        for _ in range(self.walk_length):
            deg = self.rowptr[cur + 1] - self.rowptr[cur]
            step = (torch.rand(cur.shape[0], generator=self.gen) * deg.clamp(min=1)).long()
            nxt = self.dst_sorted[(self.rowptr[cur] + step).clamp(max=self.dst_sorted.numel() - 1)]
            cur = torch.where(deg > 0, nxt, cur)
            visited.append(cur)
        return torch.cat(visited).unique()

    def _walk_device(self):
        """Roots and the walk seed from torch's device generator (seed_everything controls them), the walks themselves
        by gd_random_walk; the visited set by torch.unique on the device."""
        from ... import _lib
        from ..._lib import check, ptr, stream_ptr
        start = torch.randint(0, self.n, (self.batch_size,), device=self.dev, generator=self.gen)
        seed = int(torch.randint(0, 2 ** 62, (1,), device=self.dev, generator=self.gen))
        out = torch.empty(self.walk_length + 1, self.batch_size, dtype=torch.long, device=self.dev)
        check(_lib.lib().gd_random_walk(ptr(self._rowptr32), ptr(self._col32), self.n, ptr(start), self.batch_size,
                                        self.walk_length, seed, ptr(out), stream_ptr(self.dev)), 'gd_random_walk')
        self.last_walks = out
        return out.flatten().unique()

    def subgraph(self, nodes):
        """The batch for one (sorted, unique) node set - what GraphSAINTSampler.__getitem__ + saint_subgraph hand
        the loop [PyG-mem]: induced edges in (source, target) order with relabelled endpoints, node-sized
        tensors sliced by the node ids, edge-sized ones by the kept edges, the rest passed through."""
        from ..data import Data
        d = self.data
        n_edges = d.edge_index.shape[1]
        nodes = nodes.to(self.dev)
        member = torch.zeros(self.n, dtype=torch.bool, device=self.dev)
        member[nodes] = True
        keep = self.order[(member[d.edge_index[0]] & member[d.edge_index[1]])[self.order]]
        relabel = torch.full((self.n,), -1, dtype=torch.long, device=self.dev)
        relabel[nodes] = torch.arange(nodes.numel(), device=self.dev)
        batch = Data(num_nodes=int(nodes.numel()), edge_index=relabel[d.edge_index[:, keep]])
        for key, val in d.items():
            if key in ('edge_index', 'num_nodes'):
                continue
            if torch.is_tensor(val) and val.dim() >= 1 and val.shape[0] == self.n:
                batch[key] = val[nodes]
            elif torch.is_tensor(val) and val.dim() >= 1 and val.shape[0] == n_edges:
                batch[key] = val[keep]
            else:
                batch[key] = val
        return batch

    def node_sets(self):
        for _ in range(self.num_steps):
            yield self._walk()

    def __iter__(self):
        for nodes in self.node_sets():
            yield self.subgraph(nodes)


class FixedNodeSets(RandomWalkSubgraphSampler):
    """The same batches every epoch from a given list of node sets (parity tests inject the node sets the
    reference's loop was run on; the random-walk stream itself cannot be matched)."""

    def __init__(self, data, node_sets):
        super().__init__(data, batch_size=0, num_steps=len(node_sets))
        self._sets = [torch.as_tensor(s, dtype=torch.long).unique() for s in node_sets]

    def node_sets(self):
        return iter(self._sets)


def data_parallel_world():
    """(rank, world) of the batch-data-parallel mini-batch mode (SURVEY 8e option 1): the mini-batch loops are
    data-parallel whenever the process runs under torch.distributed with more than one rank - every rank draws its OWN
    GraphSAINT batches and the Del-weight gradients are averaged over the ranks before each optimizer step."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def sync_gradients(optimizer):
    """all-reduce (mean) of the gradients this optimizer is about to apply - one call per optimizer step, in step order
    (both_layerwise: W_D1's gradient, then W_D2's: two small all-reduces of 64 KiB and 16 KiB at H = 128, O = 64).
    RCCL on the GPU node (backend "nccl"), gloo in the tests; a no-op outside torch.distributed."""
    rank, world = data_parallel_world()
    if world == 1:
        return
    import torch.distributed as dist
    # Fixed layout over EVERY parameter of the optimizer (zeros where this rank's batch gave a parameter no gradient):
    # each rank draws different batches, and a collective whose size - or whose existence - depended on which
    # parameters happened to receive a gradient would hang or silently misalign the averages.
    params = [p for g in optimizer.param_groups for p in g['params']]
    if not params:
        return
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
    dist.all_reduce(flat)
    flat /= world
    off = 0
    for p in params:
        g = flat[off:off + p.numel()].view_as(p)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        off += p.numel()


def make_sampler(data, batch_size, num_steps, walk_length=2):
    """Sampler factory of the mini-batch loops (a seam for tests: monkeypatch to inject FixedNodeSets).  The loops keep
    their Data on the host as upstream does; the sampler gets a device copy and cuts its batches there."""
    if torch.cuda.is_available() and data.edge_index.device.type == 'cpu':
        data = data.clone().to(device)
    rank, world = data_parallel_world()
    gen = None
    if world > 1:                                   # every rank its own stream of roots and walks
        gen = torch.Generator(device=data.edge_index.device).manual_seed(int(torch.initial_seed()) % (2 ** 31) * 64 + rank + 1)
    return RandomWalkSubgraphSampler(data, batch_size=batch_size, walk_length=walk_length, num_steps=num_steps, generator=gen)


def train_minibatch(trainer, model, data, optimizer, args):
    """gnndelete_nodeemb.py:352-495: per batch, original embeddings on all batch edges, Del forward
    on the batch's S_Df edges with per-batch masks, fresh negatives, layer-wise update."""
    from .gnndelete_nodeemb import _four_terms, _non_df_masks
    _require_gpu()
    loss_fct = nn.MSELoss()
    data = data.to('cpu')
    _non_df_masks(data)
    data.edge_index = data.train_pos_edge_index
    data.node_id = torch.arange(data.x.shape[0])
    loader = make_sampler(data, args.batch_size, args.num_steps)
    model = model.to(device)
    best_metric = 0
    trainer.trainer_log['steps'] = []
    for epoch in range(args.epochs):
        model.train()
        sums = {'loss': 0.0, 'loss_l': 0.0, 'loss_r': 0.0}
        steps = 0
        for batch in loader:
            batch = batch.to(device)
            with torch.no_grad():
                z1_ori, z2_ori = model.get_original_embeddings(batch.x, batch.edge_index, return_all_emb=True)
            z1, z2 = model(batch.x, batch.edge_index[:, batch.sdf_mask].contiguous(), batch.sdf_node_1hop_mask,
                           batch.sdf_node_2hop_mask, return_all_emb=True)
            pos_edge = batch.edge_index[:, batch.df_mask]
            neg_edge = negative_sampling(batch.edge_index, batch.x.shape[0], pos_edge.shape[1])
            r1, r2, l1, l2 = _four_terms(loss_fct, z1, z2, z1_ori, z2_ori, pos_edge, neg_edge,
                                         batch.sdf_node_1hop_mask_non_df_mask, batch.sdf_node_2hop_mask_non_df_mask)
            loss1 = trainer.args.alpha * r1 + (1 - trainer.args.alpha) * l1
            loss1.backward(retain_graph=True)
            sync_gradients(optimizer[0])
            optimizer[0].step()
            optimizer[0].zero_grad()
            loss2 = trainer.args.alpha * r2 + (1 - trainer.args.alpha) * l2
            loss2.backward(retain_graph=True)
            sync_gradients(optimizer[1])
            optimizer[1].step()
            optimizer[1].zero_grad()
            step_log = {'Epoch': epoch, 'train_loss': (loss1 + loss2).item(), 'train_loss_l': (l1 + l2).item(),
                        'train_loss_r': (r1 + r2).item()}
            wandb_log(step_log)
            trainer.trainer_log['steps'].append(step_log)
            sums['loss'] += step_log['train_loss']
            sums['loss_l'] += step_log['train_loss_l']
            sums['loss_r'] += step_log['train_loss_r']
            steps += 1
        if (epoch + 1) % args.valid_freq == 0:
            valid_loss, dt_auc, dt_aup, df_auc, df_aup, df_logit, _, valid_log = trainer.eval(model, data, 'val')
            denom = max(steps - 1, 1)          # upstream divides by the last enumerate index
            train_log = {'epoch': epoch, 'train_loss': sums['loss'] / denom, 'train_loss_l': sums['loss_l'] / denom,
                         'train_loss_r': sums['loss_r'] / denom}
            trainer._record(train_log, valid_log)
            if dt_auc + df_auc > best_metric:
                best_metric = dt_auc + df_auc
                torch.save({'model_state': model.state_dict()}, os.path.join(args.checkpoint_dir, 'model_best.pt'))
            data = data.to('cpu')
    torch.save({'model_state': {k: v.to('cpu') for k, v in model.state_dict().items()}},
               os.path.join(args.checkpoint_dir, 'model_final.pt'))
