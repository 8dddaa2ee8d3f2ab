"""GNNDelete node-embedding trainers - the hot loop of the reference
(framework/trainer/gnndelete_nodeemb.py: loss zoo :19-97, GNNDeleteNodeembTrainer :99-495,
GNNDeleteNodeClassificationTrainer :498-657).

Two execution paths behind the same API:
  * fused path (default for --loss_fct mse_mean / mse_sum on GCN / GAT / GIN / GraphSAGE): the whole
    iteration - frozen-backbone forward, Del, Deleted-Edge-Consistency + Neighborhood-Influence
    losses, hand-derived backward, Adam - is gnndelete_amd.engine.NodeembEngine, one hipGraph
    replay per epoch, no per-step host sync;
  * generic path (any other loss function): autograd through the HIP-backed model with
    torch.optim.Adam, reproducing every --loss_type branch including its zero_grad placement.
Both keep the update rules of gnndelete_nodeemb.py:215-299 and the model-selection /
checkpoint behaviour of :315-349."""
import os
import time

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..graph_utils import negative_sampling
from ._log import wandb_log
from ..training_args import is_large
from .base import NodeClassificationTrainer, Trainer, _require_gpu, device

# ----------------------------------------------------------------------------- loss zoo


# The row losses run on the HIP row-pair kernel (value and gradient in one pass, gnndelete_amd.ops.rowpair_loss) where it
# applies - device tensors, a constant second argument as the trainers pass it - and as the reference's torch expressions
# otherwise; both forms are checked against the reference's own functions (tests/golden/losses.npz).
def _kld(logits, truth, per_row_mean):
    from ... import ops
    if ops.rowpair_loss_ok(logits, truth):
        kl = ops.rowpair_loss(logits, truth, 'kld').sum()
        return 1 - torch.exp(-(kl / logits.shape[0] if per_row_mean else kl))
    return 1 - torch.exp(-F.kl_div(F.log_softmax(logits, -1), truth.softmax(-1), reduction='batchmean' if per_row_mean else 'sum'))


def _cosine(logits, truth):
    from ... import ops
    if ops.rowpair_loss_ok(logits, truth):
        return ops.rowpair_loss(logits, truth, 'cosine')
    return 1 - F.cosine_similarity(logits, truth)


def BoundedKLDMean(logits, truth):
    return _kld(logits, truth, True)


def BoundedKLDSum(logits, truth):
    return _kld(logits, truth, False)


def CosineDistanceMean(logits, truth):
    return _cosine(logits, truth).mean()


def CosineDistanceSum(logits, truth):
    return _cosine(logits, truth).sum()


def centering(K):
    """H K H with H = I - 11^T / n (gnndelete_nodeemb.py:30-35), written out: K minus its row means and column means plus
    its total mean - the same matrix without two n x n x n products."""
    return K - K.mean(0, keepdim=True) - K.mean(1, keepdim=True) + K.mean()


def _gram(X):
    from ... import ops
    return ops.gram(X)


def rbf(X, sigma=None):
    gram = _gram(X)
    sq = torch.diag(gram) - gram
    dist = sq + sq.T
    if sigma is None:
        sigma = torch.median(dist[dist != 0]).sqrt()
    return torch.exp(dist * (-0.5 / (sigma * sigma)))


def linear_HSIC(X, Y):
    return torch.sum(centering(_gram(X)) * centering(_gram(Y)))


def kernel_HSIC(X, Y, sigma=None):
    return torch.sum(centering(rbf(X, sigma)) * centering(rbf(Y, sigma)))


def LinearCKA(X, Y):
    return linear_HSIC(X, Y) / (torch.sqrt(linear_HSIC(X, X)) * torch.sqrt(linear_HSIC(Y, Y)))


def RBFCKA(X, Y, sigma=None):
    return kernel_HSIC(X, Y, sigma) / (torch.sqrt(kernel_HSIC(X, X, sigma)) * torch.sqrt(kernel_HSIC(Y, Y, sigma)))


_LOSSES = {
    'kld_mean': lambda: BoundedKLDMean, 'kld_sum': lambda: BoundedKLDSum,
    'mse_mean': lambda: nn.MSELoss(reduction='mean'), 'mse_sum': lambda: nn.MSELoss(reduction='sum'),
    'cosine_mean': lambda: CosineDistanceMean, 'cosine_sum': lambda: CosineDistanceSum,
    'linear_cka': lambda: LinearCKA, 'rbf_cka': lambda: RBFCKA,
}


def get_loss_fct(name):
    if name not in _LOSSES:
        raise NotImplementedError(name)
    return _LOSSES[name]()


# ----------------------------------------------------------------------------- shared pieces
def _non_df_masks(data):
    """S_Df node masks without the endpoints of the deleted edges (gnndelete_nodeemb.py:169-173)."""
    keep = torch.ones(data.x.shape[0], dtype=torch.bool, device=data.x.device)
    keep[data.directed_df_edge_index.flatten().unique()] = False
    data.sdf_node_1hop_mask_non_df_mask = data.sdf_node_1hop_mask & keep
    data.sdf_node_2hop_mask_non_df_mask = data.sdf_node_2hop_mask & keep


def _four_terms(loss_fct, z1, z2, z1_ori, z2_ori, pos_edge, neg_edge, ni1, ni2):
    def dec(z, zo):
        return loss_fct(torch.cat([z[pos_edge[0]], z[pos_edge[1]]], dim=0),
                        torch.cat([zo[neg_edge[0]], zo[neg_edge[1]]], dim=0))
    return dec(z1, z1_ori), dec(z2, z2_ori), loss_fct(z1[ni1], z1_ori[ni1]), loss_fct(z2[ni2], z2_ori[ni2])


def _autograd_update(loss_type, alpha, r1, r2, l1, l2, optimizer):
    """One optimizer update for the given --loss_type; returns (loss, loss_r, loss_l).
    zero_grad placement is upstream's: both_all never zeroes, both_layerwise zeroes W_D1 before
    the second backward (so d loss2/d W_D1 carries into the next epoch)."""
    if loss_type == 'both_all':
        loss_r, loss_l = r1 + r2, l1 + l2
        loss = alpha * loss_r + (1 - alpha) * loss_l
        loss.backward()
        optimizer.step()
    elif loss_type == 'both_layerwise':
        loss_r, loss_l = r1 + r2, l1 + l2
        first = alpha * r1 + (1 - alpha) * l1
        first.backward(retain_graph=True)
        optimizer[0].step()
        optimizer[0].zero_grad()
        second = alpha * r2 + (1 - alpha) * l2
        second.backward(retain_graph=True)
        optimizer[1].step()
        optimizer[1].zero_grad()
        loss = first + second
    elif loss_type == 'only2_layerwise':
        loss_r, loss_l = r1 + r2, l1 + l2
        optimizer[0].zero_grad()
        loss = alpha * r2 + (1 - alpha) * l2
        loss.backward()
        optimizer[1].step()
        optimizer[1].zero_grad()
    elif loss_type == 'only2_all':
        loss_r, loss_l = r2, l2
        loss = loss_l + alpha * loss_r
        loss.backward()
        optimizer.step()
        optimizer.zero_grad()
    elif loss_type == 'only1':
        loss_r, loss_l = r1, l1
        loss = loss_l + alpha * loss_r
        loss.backward()
        optimizer.step()
        optimizer.zero_grad()
    else:
        raise NotImplementedError(loss_type)
    return loss, loss_r, loss_l


def _adam_hyper(optimizer):
    opt = optimizer[0] if isinstance(optimizer, (list, tuple)) else optimizer
    g = opt.param_groups[0]
    return g['lr'], tuple(g['betas']), g['eps']


def _export_adam_state(engine, model, optimizer):
    """Write the engine's Adam moments back into the torch optimizers the caller handed in, so
    optimizer.state_dict() is what it would be after upstream's loop."""
    opts = list(optimizer) if isinstance(optimizer, (list, tuple)) else [optimizer, optimizer]
    for opt, adam, p in [(opts[0], engine.adam1, model.deletion1.deletion_weight),
                         (opts[1], engine.adam2, model.deletion2.deletion_weight)]:
        steps = adam.applied if adam.iter_ctr is not None else int(adam.step)
        if steps:
            # (the engine's W_D2 may carry zero padding behind the parameter's own block - NodeembEngine's padded class dimension)
            r, c = p.shape
            opt.state[p] = {'step': torch.tensor(float(steps)), 'exp_avg': adam.m[:r, :c].clone(),
                            'exp_avg_sq': adam.v[:r, :c].clone()}


class _EmbeddingUnlearner:
    """Loop shared by the link-prediction and node-classification variants."""

    def _can_fuse(self, model, loss_name):
        from ...engine import NodeembEngine  # noqa: F401
        from ...nn import GATConv, GCNConv, GINConv, SAGEConv
        conv2 = getattr(model, 'conv2', None)
        return loss_name in ('mse_mean', 'mse_sum') and isinstance(conv2, (GCNConv, GATConv, GINConv, SAGEConv)) and \
            not (isinstance(conv2, GINConv) and conv2.nn.out_features > conv2.nn.in_features)

    def _unlearn(self, model, data, optimizer, args, edge_key, loss_name, loss_type, select_best):
        _require_gpu()
        model = model.to(device)
        data = data.to(device)
        edges = data[edge_key]
        _non_df_masks(data)
        e_dr = edges[:, data.dr_mask].contiguous()
        e_sdf = edges[:, data.sdf_mask].contiguous()        # hoisted: upstream re-slices it every epoch
        pos_edge = edges[:, data.df_mask]
        with torch.no_grad():
            z1_ori, z2_ori = model.get_original_embeddings(data.x, e_dr, return_all_emb=True)
        neg_edge = negative_sampling(edge_index=edges, num_nodes=data.num_nodes, num_neg_samples=int(data.df_mask.sum()))
        ni1, ni2 = data.sdf_node_1hop_mask_non_df_mask, data.sdf_node_2hop_mask_non_df_mask

        engine = None
        if self._can_fuse(model, loss_name) and not getattr(args, 'no_fused_step', False):
            from ...engine import NodeembEngine
            lr, betas, eps = _adam_hyper(optimizer)
            engine = NodeembEngine(model, data.x, e_sdf, z1_ori, z2_ori, pos_edge, neg_edge, ni1, ni2,
                                   loss_type=loss_type, alpha=self.args.alpha, lr=lr,
                                   reduction='mean' if loss_name == 'mse_mean' else 'sum',
                                   history=max(16, args.epochs),
                                   cache_layer1=not getattr(args, 'no_layer1_cache', False),
                                   affected_rows_only=not getattr(args, 'all_rows', False))
            engine.adam1.betas = engine.adam2.betas = betas
            engine.adam1.eps = engine.adam2.eps = eps
        loss_fct = get_loss_fct(loss_name)

        best_metric = 0
        t_block = time.time()
        for epoch in range(args.epochs):
            model.train()
            if engine is not None:
                engine.step()
            else:
                z1, z2 = model(data.x, e_sdf, return_all_emb=True)
                r1, r2, l1, l2 = _four_terms(loss_fct, z1, z2, z1_ori, z2_ori, pos_edge, neg_edge, ni1, ni2)
                loss, loss_r, loss_l = _autograd_update(loss_type, self.args.alpha, r1, r2, l1, l2, optimizer)

            if (epoch + 1) % self.args.valid_freq == 0:
                if engine is not None:
                    last = engine.loss_history()[-1]          # the only host sync of the block
                    cur = {'train_loss': float(last[0]), 'loss_r': float(last[1]), 'loss_l': float(last[2])}
                else:
                    cur = {'train_loss': loss.item(), 'loss_r': loss_r.item(), 'loss_l': loss_l.item()}
                torch.cuda.synchronize()
                epoch_time = (time.time() - t_block) / self.args.valid_freq
                train_log = {'epoch': epoch, **cur, 'train_time': epoch_time}
                metric, valid_loss, valid_log = select_best(model, data)
                valid_log['epoch'] = epoch
                self._record(train_log, valid_log)
                if metric > best_metric:
                    best_metric = metric
                    print(f'Save best checkpoint at epoch {epoch:04d}. Valid loss = {valid_loss:.4f}')
                    torch.save({'model_state': model.state_dict()}, os.path.join(args.checkpoint_dir, 'model_best.pt'))
                t_block = time.time()
            elif engine is None:
                wandb_log({'Epoch': epoch, 'train_loss': loss.item(), 'loss_r': loss_r.item(), 'loss_l': loss_l.item()})

        if engine is not None:
            _export_adam_state(engine, model, optimizer)
            self.trainer_log['loss_history'] = engine.loss_history().tolist()
        torch.save({'model_state': {k: v.to('cpu') for k, v in model.state_dict().items()}},
                   os.path.join(args.checkpoint_dir, 'model_final.pt'))


class GNNDeleteNodeembTrainer(_EmbeddingUnlearner, Trainer):

    def train(self, model, data, optimizer, args, logits_ori=None, attack_model_all=None, attack_model_sub=None):
        if is_large(self.args.dataset) and getattr(args, 'minibatch', False):
            return self.train_minibatch(model, data, optimizer, args, logits_ori, attack_model_all, attack_model_sub)
        return self.train_fullbatch(model, data, optimizer, args, logits_ori, attack_model_all, attack_model_sub)

    def train_fullbatch(self, model, data, optimizer, args, logits_ori=None, attack_model_all=None,
                        attack_model_sub=None):
        def select(model, data):
            valid_loss, dt_auc, dt_aup, df_auc, df_aup, df_logit, _, valid_log = self.eval(model, data, 'val')
            return dt_auc + df_auc, valid_loss, valid_log
        self._unlearn(model, data, optimizer, args, 'train_pos_edge_index', self.args.loss_fct,
                      self.args.loss_type, select)

    def train_minibatch(self, model, data, optimizer, args, logits_ori=None, attack_model_all=None,
                        attack_model_sub=None):
        """Upstream trains ogbl-* graphs on GraphSAINT random-walk subgraphs because they do not fit
        its GPUs together with the N x N bookkeeping (gnndelete_nodeemb.py:352-495).  On MI355X the
        whole graph is one batch; see .sampler for the mini-batch loop kept for parity studies."""
        from .sampler import train_minibatch
        return train_minibatch(self, model, data, optimizer, args)


class GNNDeleteNodeClassificationTrainer(_EmbeddingUnlearner, NodeClassificationTrainer):
    """Node / node-feature unlearning (delete_node.py): same losses on data.edge_index, always the
    layer-wise update; model selection on accuracy + micro-F1 (gnndelete_nodeemb.py:498-657)."""

    def train(self, model, data, optimizer, args, logits_ori=None, attack_model_all=None, attack_model_sub=None):
        def select(model, data):
            valid_loss, dt_acc, dt_f1, valid_log = self.eval(model, data, 'val')
            return dt_acc + dt_f1, valid_loss, valid_log
        self._unlearn(model, data, optimizer, args, 'edge_index', self.args.loss_fct, 'both_layerwise', select)
