"""CPU restatement of the code the reference OWNS on the GNNDelete hot path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Pinned against the reference
itself by tests/golden/make_golden.py -> tests/golden/*.npz (run in the build
container, where /root/reference can be imported under third-party stubs) and
checked in tests/test_oracle_golden.py.

What each piece follows (paths relative to /root/reference):

  DeletionLayer            framework/models/deletion.py:8-29
  TwoLayer / TwoLayerDelete framework/models/{gcn,gat,gin}.py, rgcn.py:9-47,
                           deletion.py:52-163 (conv1 -> Del1 -> relu -> conv2 -> Del2)
  decode / distmult        framework/models/gcn.py:26-36, rgcn.py:40-47
  LOSSES                   framework/trainer/gnndelete_nodeemb.py:19-97
  nodeemb_terms / epoch    framework/trainer/gnndelete_nodeemb.py:169-299
  edgeprob_terms/fullbatch framework/trainer/gnndelete.py:174-193, 211-258
  original_fullbatch       framework/trainer/base.py:75-142
  eval_linkpred            framework/trainer/base.py:229-305
  negative_sampling_kg     framework/utils.py:46-58
  nodeemb_minibatch        framework/trainer/gnndelete_nodeemb.py:352-443 (GraphSAINT batches injected)
  kg_nodeemb_minibatch     framework/trainer/gnndelete_nodeemb.py:734-800
  eval_kg                  framework/trainer/base.py:495-567
  retrain_fullbatch        framework/trainer/retrain.py:57-131
  kg_retrain_minibatch     framework/trainer/retrain.py:235-339 (GraphSAINT batches injected)
  verification_error       framework/evaluation.py:63-81
  split_edges              prepare_dataset.py:31-136 (+ IN / OUT masks :205-214)
  original_minibatch       framework/trainer/base.py:144-227 (GraphSAINT batches injected)
  kg_original_minibatch    framework/trainer/base.py:394-493

Build semantics: the backbone is truly frozen - conv1
runs under no_grad for every architecture (upstream does so for GAT/GIN/RGCN; for GCN
it leaves conv1 differentiable, which only adds a wasted weight-gradient and makes
GCN + both_layerwise crash).  ``grad_through_conv1=True`` restores the upstream GCN
behaviour for the golden-vector comparison.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import pyg_semantics as pyg


# ----------------------------------------------------------------------------
# Del operator + model wiring
# ----------------------------------------------------------------------------
class DeletionLayer(nn.Module):
    """x[mask] <- x[mask] @ W_D on a copy; identity when no mask (deletion.py:17-29)."""

    def __init__(self, dim, mask):
        super().__init__()
        self.dim = dim
        self.mask = mask
        self.deletion_weight = nn.Parameter(torch.full((dim, dim), 1.0 / 1000))

    def forward(self, x, mask=None):
        m = self.mask if mask is None else mask
        if m is None:
            return x
        out = x.clone()
        out[m] = out[m] @ self.deletion_weight
        return out


def _make_convs(gnn, in_dim, hidden, out_dim, num_edge_type=None):
    if gnn == 'gcn':
        return pyg.GCNConv(in_dim, hidden), pyg.GCNConv(hidden, out_dim)
    if gnn == 'gat':
        return pyg.GATConv(in_dim, hidden), pyg.GATConv(hidden, out_dim)
    if gnn == 'gin':
        return pyg.GINConv(nn.Linear(in_dim, hidden)), pyg.GINConv(nn.Linear(hidden, out_dim))
    if gnn == 'sage':
        return pyg.SAGEConv(in_dim, hidden), pyg.SAGEConv(hidden, out_dim)
    if gnn == 'rgcn':
        nb = 4 if num_edge_type > 20 else None          # rgcn.py:17-22
        return (pyg.RGCNConv(in_dim, hidden, 2 * num_edge_type, nb),
                pyg.RGCNConv(hidden, out_dim, 2 * num_edge_type, nb))
    if gnn == 'rgat':
        nb = 4 if num_edge_type > 20 else None          # rgat.py:361-366
        return (pyg.RGATConv(in_dim, hidden, 2 * num_edge_type, nb),
                pyg.RGATConv(hidden, out_dim, 2 * num_edge_type, nb))
    raise NotImplementedError(gnn)


class TwoLayer(nn.Module):
    """The reference's 2-layer backbones; ``gnn`` in {gcn, gat, gin, rgcn, rgat} (+ sage)."""

    def __init__(self, gnn, in_dim, hidden, out_dim, num_nodes=None, num_edge_type=None):
        super().__init__()
        self.gnn = gnn
        self.relational = gnn in ('rgcn', 'rgat')
        if self.relational:
            self.num_edge_type = num_edge_type
            self.node_emb = nn.Embedding(num_nodes, in_dim)
        self.conv1, self.conv2 = _make_convs(gnn, in_dim, hidden, out_dim, num_edge_type)
        if self.relational:
            self.W = nn.Parameter(torch.empty(num_edge_type, out_dim))
            nn.init.xavier_uniform_(self.W, gain=nn.init.calculate_gain('relu'))

    def _args(self, edge_index, edge_type):
        return (edge_index, edge_type) if self.relational else (edge_index,)

    def backbone(self, x, edge_index, edge_type=None, return_all_emb=False):
        if self.relational:
            x = self.node_emb(x)
        g = self._args(edge_index, edge_type)
        x1 = self.conv1(x, *g)
        x2 = self.conv2(F.relu(x1), *g)
        return (x1, x2) if return_all_emb else x2

    def forward(self, x, edge_index, edge_type=None, return_all_emb=False):
        return self.backbone(x, edge_index, edge_type, return_all_emb)

    def decode(self, z, pos_edge_index, neg_or_type=None):
        if self.relational:                                           # DistMult
            return (z[pos_edge_index[0]] * self.W[neg_or_type] * z[pos_edge_index[1]]).sum(1)
        ei = pos_edge_index if neg_or_type is None else torch.cat([pos_edge_index, neg_or_type], -1)
        return (z[ei[0]] * z[ei[1]]).sum(-1)


class TwoLayerDelete(TwoLayer):
    def __init__(self, gnn, in_dim, hidden, out_dim, mask_1hop=None, mask_2hop=None,
                 num_nodes=None, num_edge_type=None, grad_through_conv1=False):
        super().__init__(gnn, in_dim, hidden, out_dim, num_nodes, num_edge_type)
        self.deletion1 = DeletionLayer(hidden, mask_1hop)
        self.deletion2 = DeletionLayer(out_dim, mask_2hop)
        self.grad_through_conv1 = grad_through_conv1

    def forward(self, x, edge_index, edge_type=None, mask_1hop=None, mask_2hop=None,
                return_all_emb=False):
        g = self._args(edge_index, edge_type)
        with torch.set_grad_enabled(self.grad_through_conv1 and torch.is_grad_enabled()):
            if self.relational:
                x = self.node_emb(x)
            p1 = self.conv1(x, *g)
        x1 = self.deletion1(p1, mask_1hop)
        x2 = self.deletion2(self.conv2(F.relu(x1), *g), mask_2hop)
        return (x1, x2) if return_all_emb else x2

    def get_original_embeddings(self, x, edge_index, edge_type=None, return_all_emb=False):
        return self.backbone(x, edge_index, edge_type, return_all_emb)


# ----------------------------------------------------------------------------
# loss zoo (gnndelete_nodeemb.py:19-97)
# ----------------------------------------------------------------------------
def _bounded_kld(reduction):
    def f(logits, truth):
        kl = F.kl_div(F.log_softmax(logits, -1), truth.softmax(-1), reduction=reduction)
        return 1 - torch.exp(-kl)
    return f


def _cosine(reduce):
    def f(logits, truth):
        d = 1 - F.cosine_similarity(logits, truth)
        return d.mean() if reduce == 'mean' else d.sum()
    return f


def _center(k):
    n = k.shape[0]
    h = torch.eye(n, dtype=k.dtype) - torch.ones(n, n, dtype=k.dtype) / n
    return h @ k @ h


def _linear_hsic(x, y):
    return (_center(x @ x.T) * _center(y @ y.T)).sum()


def linear_cka(x, y):
    return _linear_hsic(x, y) / (torch.sqrt(_linear_hsic(x, x)) * torch.sqrt(_linear_hsic(y, y)))


def _rbf(x, sigma):
    """gnndelete_nodeemb.py:38-47 with the sigma given (the default-sigma branch calls `math.sqrt` in a module that never
    imports math: NameError upstream)."""
    g = x @ x.T
    k = torch.diag(g) - g
    k = k + k.T
    return torch.exp(k * (-0.5 / (sigma * sigma)))


def _kernel_hsic(x, y, sigma):
    return (_center(_rbf(x, sigma)) * _center(_rbf(y, sigma))).sum()


def rbf_cka(x, y, sigma):
    """gnndelete_nodeemb.py:49-50,62-66 (RBFCKA)."""
    return _kernel_hsic(x, y, sigma) / (torch.sqrt(_kernel_hsic(x, x, sigma)) * torch.sqrt(_kernel_hsic(y, y, sigma)))


LOSSES = {
    'mse_mean': nn.MSELoss(reduction='mean'),
    'mse_sum': nn.MSELoss(reduction='sum'),
    'kld_mean': _bounded_kld('batchmean'),
    'kld_sum': _bounded_kld('sum'),
    'cosine_mean': _cosine('mean'),
    'cosine_sum': _cosine('sum'),
    'linear_cka': linear_cka,
    'rbf_cka': rbf_cka,                      # (x, y, sigma)
}


# ----------------------------------------------------------------------------
# node-embedding trainer, full batch (gnndelete_nodeemb.py:169-299)
# ----------------------------------------------------------------------------
def non_df_masks(num_nodes, directed_df_edge_index, sdf1, sdf2):
    keep = torch.ones(num_nodes, dtype=torch.bool)
    keep[directed_df_edge_index.flatten().unique()] = False
    return sdf1 & keep, sdf2 & keep


def nodeemb_terms(z1, z2, z1_ori, z2_ori, pos_edge, neg_edge, ni_mask1, ni_mask2, loss_fct):
    """Deleted-Edge-Consistency (r) and Neighborhood-Influence (l) terms per layer."""
    def dec(z, zo):
        return loss_fct(torch.cat([z[pos_edge[0]], z[pos_edge[1]]], 0),
                        torch.cat([zo[neg_edge[0]], zo[neg_edge[1]]], 0))
    r1, r2 = dec(z1, z1_ori), dec(z2, z2_ori)
    l1 = loss_fct(z1[ni_mask1], z1_ori[ni_mask1])
    l2 = loss_fct(z2[ni_mask2], z2_ori[ni_mask2])
    return r1, r2, l1, l2


def nodeemb_epoch(model, fwd, targets, optimizer, loss_type, alpha, loss_fct):
    """One pass of the loop body.  ``fwd()`` -> (z1, z2); ``targets`` = dict with
    z1_ori, z2_ori, pos_edge, neg_edge, ni_mask1, ni_mask2.  ``optimizer`` is a list of
    two Adams for *layerwise* types, one Adam otherwise (delete_gnn.py:221-226).
    Reproduces the zero_grad placement of every branch, including the gradient
    carry-over of both_layerwise and the never-zeroed grads of both_all."""
    z1, z2 = fwd()
    r1, r2, l1, l2 = nodeemb_terms(z1, z2, targets['z1_ori'], targets['z2_ori'],
                                   targets['pos_edge'], targets['neg_edge'],
                                   targets['ni_mask1'], targets['ni_mask2'], loss_fct)
    if loss_type == 'both_all':
        loss_l, loss_r = l1 + l2, r1 + r2
        loss = alpha * loss_r + (1 - alpha) * loss_l
        loss.backward()
        optimizer.step()
    elif loss_type == 'both_layerwise':
        loss_l, loss_r = l1 + l2, r1 + r2
        loss1 = alpha * r1 + (1 - alpha) * l1
        loss1.backward(retain_graph=True)
        optimizer[0].step()
        optimizer[0].zero_grad()
        loss2 = alpha * r2 + (1 - alpha) * l2
        loss2.backward(retain_graph=True)
        optimizer[1].step()
        optimizer[1].zero_grad()
        loss = loss1 + loss2
    elif loss_type == 'only2_layerwise':
        loss_l, loss_r = l1 + l2, r1 + r2
        optimizer[0].zero_grad()
        loss = alpha * r2 + (1 - alpha) * l2
        loss.backward()
        optimizer[1].step()
        optimizer[1].zero_grad()
    elif loss_type == 'only2_all':
        loss_l, loss_r = l2, r2
        loss = loss_l + alpha * loss_r
        loss.backward()
        optimizer.step()
        optimizer.zero_grad()
    elif loss_type == 'only1':
        loss_l, loss_r = l1, r1
        loss = loss_l + alpha * loss_r
        loss.backward()
        optimizer.step()
        optimizer.zero_grad()
    else:
        raise NotImplementedError(loss_type)
    return {'train_loss': loss.item(), 'loss_r': loss_r.item(), 'loss_l': loss_l.item(),
            'z1': z1.detach(), 'z2': z2.detach()}


def make_optimizer(model, loss_type, lr):
    if 'layerwise' in loss_type:
        return [torch.optim.Adam(model.deletion1.parameters(), lr=lr),
                torch.optim.Adam(model.deletion2.parameters(), lr=lr)]
    dels = [p for n, p in model.named_parameters() if 'del' in n]
    return torch.optim.Adam([{'params': dels, 'weight_decay': 0.0}], lr=lr)


def nodeemb_fullbatch(model, data, epochs, loss_type='both_layerwise', alpha=0.5,
                      loss_fct='mse_mean', lr=1e-3, neg_edge=None, optimizer=None):
    """train_fullbatch without logging/validation.  ``data`` is a dict with the
    reference's Data attributes; ``neg_edge`` is injected (PyG negative_sampling uses
    Python's ``random`` and cannot be reproduced)."""
    relational = model.relational
    et = data.get('edge_type')
    ni1, ni2 = non_df_masks(data['x'].shape[0], data['directed_df_edge_index'],
                            data['sdf_node_1hop_mask'], data['sdf_node_2hop_mask'])
    E = data['train_pos_edge_index']
    dr, sdf, df = data['dr_mask'], data['sdf_mask'], data['df_mask']
    with torch.no_grad():
        z1o, z2o = model.get_original_embeddings(
            data['x'], E[:, dr], et[dr] if relational else None, return_all_emb=True)
    targets = dict(z1_ori=z1o, z2_ori=z2o, pos_edge=E[:, df], neg_edge=neg_edge,
                   ni_mask1=ni1, ni_mask2=ni2)
    opt = optimizer if optimizer is not None else make_optimizer(model, loss_type, lr)
    fct = LOSSES[loss_fct]

    def fwd():
        return model(data['x'], E[:, sdf], et[sdf] if relational else None, return_all_emb=True)

    logs = []
    for _ in range(epochs):
        model.train()
        logs.append(nodeemb_epoch(model, fwd, targets, opt, loss_type, alpha, fct))
    return logs, targets


def original_fullbatch(model, data, epochs, lr, neg_edge):
    """Original-model training, Trainer.train_fullbatch (framework/trainer/base.py:75-142) without
    evaluation: BCE-with-logits over all training edges (label 1) and the injected negatives (label 0),
    one Adam over every parameter, zero_grad after the step.  -> per-epoch train_loss."""
    E = data['train_pos_edge_index']
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    label = torch.cat([torch.ones(E.shape[1]), torch.zeros(neg_edge.shape[1])])
    losses = []
    for _ in range(epochs):
        model.train()
        z = model(data['x'], E)
        loss = F.binary_cross_entropy_with_logits(model.decode(z, E, neg_edge), label)
        loss.backward()
        opt.step()
        opt.zero_grad()
        losses.append(float(loss))
    return losses


# ----------------------------------------------------------------------------
# edge-probability trainer (gnndelete.py:138-309)
# ----------------------------------------------------------------------------
def sdf_pair_index(num_nodes, sdf_node_mask, df_edges):
    """Strictly-lower-triangular pairs (i>j) of S_Df nodes, minus the Df pairs
    (gnndelete.py:174-193), as a [2,P] index instead of an N x N mask, in the
    row-major order boolean indexing of the N x N matrix would give."""
    nodes = sdf_node_mask.nonzero().flatten()
    i, j = torch.meshgrid(nodes, nodes, indexing='ij')
    sel = i > j
    i, j = i[sel], j[sel]
    dfk = set((torch.maximum(df_edges[0], df_edges[1]) * num_nodes
               + torch.minimum(df_edges[0], df_edges[1])).tolist())
    key = (i * num_nodes + j).tolist()
    keep = torch.tensor([k not in dfk for k in key], dtype=torch.bool)
    return torch.stack([i[keep], j[keep]])


def edgeprob_terms(model, z, df_edges, neg_edge, pair_index, logits_ori_pairs):
    """loss_r = MSE(logit(Df), logit(neg)); loss_l = MSE(sigma(z z^T)[pairs], sigma(ori)[pairs])."""
    n = df_edges.shape[1]
    lg = model.decode(z, df_edges, neg_edge)
    loss_r = F.mse_loss(lg[:n], lg[n:])
    if pair_index.shape[1] == 0:
        return loss_r, torch.zeros(())
    cur = (z[pair_index[0]] * z[pair_index[1]]).sum(-1).sigmoid()
    return loss_r, F.mse_loss(cur, logits_ori_pairs.sigmoid())


def edgeprob_fullbatch(model, data, epochs, logits_ori, lr, neg_edge):
    """GNNDeleteTrainer.train_fullbatch (gnndelete.py:211-258) without evaluation: fixed injected
    negatives, loss = 0.5 loss_r + 0.5 loss_l, ONE Adam over the Del weights, zero_grad after the
    step.  -> per-epoch dicts(train_loss, loss_r, loss_l)."""
    E = data['train_pos_edge_index']
    df_edges, e_sdf = E[:, data['df_mask']], E[:, data['sdf_mask']]
    pairs = sdf_pair_index(data['x'].shape[0], data['sdf_node_2hop_mask'], df_edges)
    ori_pairs = logits_ori[pairs[0], pairs[1]]
    opt = torch.optim.Adam([p for n, p in model.named_parameters() if 'del' in n], lr=lr)
    logs = []
    for _ in range(epochs):
        model.train()
        z = model(data['x'], e_sdf)
        loss_r, loss_l = edgeprob_terms(model, z, df_edges, neg_edge, pairs, ori_pairs)
        loss = 0.5 * loss_r + 0.5 * loss_l
        loss.backward()
        opt.step()
        opt.zero_grad()
        logs.append(dict(train_loss=float(loss), loss_r=float(loss_r), loss_l=float(loss_l)))
    return logs


# ----------------------------------------------------------------------------
# evaluation (base.py:229-305) and KG negatives (utils.py:46-58)
# ----------------------------------------------------------------------------
def eval_linkpred(model, data, stage, df_pos_masks, unlearning_model='gnndelete_nodeemb'):
    """Trainer.eval.  ``df_pos_masks`` = the cached list of boolean Dr-subset masks
    (base.py:263-268; drawn with torch.randperm by the caller).  Quirks kept: BCE
    *with logits* applied to already-sigmoided scores; Df labelled 0, Dr labelled 1."""
    from sklearn.metrics import roc_auc_score, average_precision_score
    model.eval()
    with torch.no_grad():
        pos, neg = data[f'{stage}_pos_edge_index'], data[f'{stage}_neg_edge_index']
        E = data['train_pos_edge_index']
        mask = data['dtrain_mask'] if 'dtrain_mask' in data else data['dr_mask']
        z = model(data['x'], E[:, mask])
        prob = model.decode(z, pos, neg).sigmoid()
        label = torch.zeros(pos.shape[1] + neg.shape[1])
        label[:pos.shape[1]] = 1.0
        loss = F.binary_cross_entropy_with_logits(prob, label).item()
        dt_auc = roc_auc_score(label.numpy(), prob.numpy())
        dt_aup = average_precision_score(label.numpy(), prob.numpy())
        if unlearning_model == 'original':
            df_logit = []
        else:
            df_logit = model.decode(z, data['directed_df_edge_index']).sigmoid().tolist()
        if df_logit:
            dr_edges = E[:, data['dr_mask']]
            aucs, aups = [], []
            lab = [0] * len(df_logit) + [1] * len(df_logit)
            for m in df_pos_masks:
                pl = model.decode(z, dr_edges[:, m]).sigmoid().tolist()
                aucs.append(roc_auc_score(lab, df_logit + pl))
                aups.append(average_precision_score(lab, df_logit + pl))
            df_auc, df_aup = float(np.mean(aucs)), float(np.mean(aups))
        else:
            df_auc = df_aup = float('nan')
    return {'loss': loss, 'dt_auc': dt_auc, 'dt_aup': dt_aup, 'df_auc': df_auc,
            'df_aup': df_aup, 'df_logit': df_logit, 'z': z}


def negative_sampling_kg(edge_index, edge_type, generator=None):
    """Per relation type, permute the head column with torch.randperm (global RNG)."""
    out = edge_index.clone()
    for et in edge_type.unique():
        sel = edge_type == et
        heads = out[0, sel]
        perm = torch.randperm(heads.shape[0], generator=generator)
        out[0, sel] = heads[perm]
    return out


# ----------------------------------------------------------------------------
# mini-batch loops on injected GraphSAINT batches
# ----------------------------------------------------------------------------
def _layerwise_step(optimizer, alpha, r1, r2, l1, l2):
    """The update both mini-batch loops share (gnndelete_nodeemb.py:433-443, 788-798): step + zero_grad per layer;
    loss2's gradient w.r.t. W_D1 (through conv2) stays in W_D1.grad and is consumed by the NEXT batch's step."""
    loss1 = alpha * r1 + (1 - alpha) * l1
    loss1.backward(retain_graph=True)
    optimizer[0].step()
    optimizer[0].zero_grad()
    loss2 = alpha * r2 + (1 - alpha) * l2
    loss2.backward(retain_graph=True)
    optimizer[1].step()
    optimizer[1].zero_grad()
    return loss1 + loss2


def nodeemb_minibatch(model, data, node_sets, negs, epochs, alpha, lr):
    """GNNDeleteNodeembTrainer.train_minibatch (gnndelete_nodeemb.py:352-443) without validation: per batch the
    original embeddings on ALL batch edges (Df included, :399-400), Del forward on the batch's S_Df edges with the
    batch's node masks, plain MSE (:357), negatives per batch (``negs``, consumed in order), layer-wise update.
    -> per-step dicts(train_loss, train_loss_l, train_loss_r)."""
    d = dict(data)
    d['sdf_node_1hop_mask_non_df_mask'], d['sdf_node_2hop_mask_non_df_mask'] = non_df_masks(
        d['x'].shape[0], d['directed_df_edge_index'], d['sdf_node_1hop_mask'], d['sdf_node_2hop_mask'])
    d['edge_index'] = d['train_pos_edge_index']
    opt = make_optimizer(model, 'both_layerwise', lr)
    fct = nn.MSELoss()
    negs = iter(negs)
    logs = []
    for _ in range(epochs):
        model.train()
        for nodes in node_sets:
            b = pyg.saint_subgraph(d, nodes)
            with torch.no_grad():
                z1o, z2o = model.get_original_embeddings(b['x'], b['edge_index'], return_all_emb=True)
            z1, z2 = model(b['x'], b['edge_index'][:, b['sdf_mask']], None, b['sdf_node_1hop_mask'], b['sdf_node_2hop_mask'],
                           return_all_emb=True)
            pos = b['edge_index'][:, b['df_mask']]
            neg = next(negs)
            assert neg.shape[1] == pos.shape[1]
            r1, r2, l1, l2 = nodeemb_terms(z1, z2, z1o, z2o, pos, neg, b['sdf_node_1hop_mask_non_df_mask'],
                                           b['sdf_node_2hop_mask_non_df_mask'], fct)
            loss = _layerwise_step(opt, alpha, r1, r2, l1, l2)
            logs.append(dict(train_loss=float(loss), train_loss_l=float(l1 + l2), train_loss_r=float(r1 + r2)))
    return logs


def edgeprob_minibatch(model, data, node_sets, negs, epochs, lr):
    """GNNDeleteTrainer.train_minibatch (framework/trainer/gnndelete.py:312-450) without validation, with the
    data.dtrain_mask upstream never sets (base.py:59; delete_gnn.py:124,190 commented out) taken as dr_mask: z_ori = the
    model's embedding of the whole graph on those edges, once (:330); per batch the Del forward on the batch's S_Df edges
    (:355), negatives per batch (``negs``, consumed in order, :360-364), loss_e = MSE(Df logits, negative logits)
    (:366-367), loss_l = MSE of the dot products over the batch's S_Df edges with row < col against z_ori indexed with
    the BATCH-LOCAL node ids (:378-386 - upstream's quirk, kept), 0.5 / 0.5 (:391-399), single Adam, zero_grad after the
    step.  -> per-step dicts(loss, loss_e, loss_l); the epoch logs upstream prints divide the epoch sums by the last
    enumerate index TWICE and swap the names of the two terms (:411-423): `edgeprob_minibatch_epoch_log`."""
    d = dict(data)
    d['edge_index'] = d['train_pos_edge_index']
    with torch.no_grad():
        z_ori = model(d['x'], d['train_pos_edge_index'][:, d['dr_mask']])
    opt = make_optimizer(model, 'both_all', lr)
    fct = nn.MSELoss()
    negs = iter(negs)
    logs = []
    for _ in range(epochs):
        model.train()
        for nodes in node_sets:
            b = pyg.saint_subgraph(d, nodes)
            ei = b['edge_index']
            z = model(b['x'], ei[:, b['sdf_mask']], None, b['sdf_node_1hop_mask'], b['sdf_node_2hop_mask'])
            pos = ei[:, b['df_mask']]
            neg = next(negs)
            k = pos.shape[1]
            assert neg.shape[1] == k
            df_logits = model.decode(z, pos, neg)
            loss_e = fct(df_logits[:k], df_logits[k:])
            edge = ei[:, b['sdf_mask']]
            lower = edge[0] < edge[1]
            row, col = edge[0][lower], edge[1][lower]
            logits_ori = (z_ori[row] * z_ori[col]).sum(-1)
            logits = (z[row] * z[col]).sum(-1)
            loss_l = fct(logits, logits_ori)
            loss = 0.5 * loss_e + 0.5 * loss_l
            loss.backward()
            opt.step()
            opt.zero_grad()
            logs.append(dict(loss=loss.item(), loss_e=loss_e.item(), loss_l=loss_l.item()))
    return logs


def edgeprob_minibatch_epoch_log(step_logs):
    """What gnndelete.py:401-423 prints for one epoch from its per-step values: sums divided by step = the LAST enumerate
    index (n - 1), then by step again in the log dict, with the names of the two terms swapped."""
    step = max(len(step_logs) - 1, 1)
    tot = {k: sum(s_[k] for s_ in step_logs) / step for k in ('loss', 'loss_e', 'loss_l')}
    return {'train_loss': tot['loss'] / step, 'train_loss_l': tot['loss_e'] / step, 'train_loss_e': tot['loss_l'] / step}


def kg_nodeemb_minibatch(model, data, node_sets, num_edge_type, epochs, alpha, lr, loss_fct='mse_mean'):
    """KGGNNDeleteNodeembTrainer.train (gnndelete_nodeemb.py:734-800) without validation: message passing on the
    batch's Dr edges with the S_Df-minus-Df node masks as the Del masks (:749-751), DEC on the forward-direction Df
    triples only (:761) against per-relation head-shuffled negatives drawn from the GLOBAL torch RNG (:766-768),
    NI on the same node masks, layer-wise update.  -> per-step dicts(train_loss, loss_r, loss_l)."""
    d = dict(data)
    d['sdf_node_1hop_mask_non_df_mask'], d['sdf_node_2hop_mask_non_df_mask'] = non_df_masks(
        d['x'].shape[0], d['directed_df_edge_index'], d['sdf_node_1hop_mask'], d['sdf_node_2hop_mask'])
    opt = make_optimizer(model, 'both_layerwise', lr)
    fct = LOSSES[loss_fct]
    logs = []
    for _ in range(epochs):
        model.train()
        for nodes in node_sets:
            b = pyg.saint_subgraph(d, nodes)
            ei, et = b['edge_index'][:, b['dr_mask']], b['edge_type'][b['dr_mask']]
            m1, m2 = b['sdf_node_1hop_mask_non_df_mask'], b['sdf_node_2hop_mask_non_df_mask']
            z1, z2 = model(b['x'], ei, et, m1, m2, return_all_emb=True)
            with torch.no_grad():
                z1o, z2o = model.get_original_embeddings(b['x'], ei, et, return_all_emb=True)
            pos, pt = b['edge_index'][:, b['df_mask']], b['edge_type'][b['df_mask']]
            fw = pt < num_edge_type
            dec, dec_t = pos[:, fw], pt[fw]
            neg = negative_sampling_kg(dec, dec_t)
            r1, r2, l1, l2 = nodeemb_terms(z1, z2, z1o, z2o, dec, neg, m1, m2, fct)
            loss = _layerwise_step(opt, alpha, r1, r2, l1, l2)
            logs.append(dict(train_loss=float(loss), loss_r=float(r1 + r2), loss_l=float(l1 + l2)))
    return logs


def original_minibatch(model, data, node_sets, negs, epochs, lr):
    """Trainer.train_minibatch (base.py:144-227) without validation: per batch BCE-with-logits link prediction on the
    batch's edges against one negative per edge (``negs``, consumed in order), Adam on every parameter, zero_grad
    after the step.  -> per-step train_loss."""
    d = dict(data)
    d['edge_index'] = d['train_pos_edge_index']
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    negs = iter(negs)
    losses = []
    for _ in range(epochs):
        model.train()
        for nodes in node_sets:
            b = pyg.saint_subgraph(d, nodes)
            ei = b['edge_index']
            z = model(b['x'], ei)
            neg = next(negs)
            assert neg.shape[1] == ei.shape[1]
            label = torch.cat([torch.ones(ei.shape[1]), torch.zeros(neg.shape[1])])
            loss = F.binary_cross_entropy_with_logits(model.decode(z, ei, neg), label)
            loss.backward()
            opt.step()
            opt.zero_grad()
            losses.append(float(loss))
    return losses


def kg_original_minibatch(model, data, node_sets, num_edge_type, epochs, lr):
    """KGTrainer.train (base.py:394-493) without validation: message passing on ALL batch edges (both directions),
    DistMult scores of the forward-direction batch edges against per-relation head-shuffled negatives (global torch
    RNG), BCE-with-logits, Adam on every parameter.  -> per-step train_loss."""
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    losses = []
    for _ in range(epochs):
        model.train()
        for nodes in node_sets:
            b = pyg.saint_subgraph(dict(data), nodes)
            ei, et = b['edge_index'], b['edge_type']
            z = model(b['x'], ei, et)
            fw = et < num_edge_type
            dec, dec_t = ei[:, fw], et[fw]
            neg = negative_sampling_kg(dec, dec_t)
            logits = torch.cat([model.decode(z, dec, dec_t), model.decode(z, neg, dec_t)], -1)
            label = torch.cat([torch.ones(dec.shape[1]), torch.zeros(neg.shape[1])])
            loss = F.binary_cross_entropy_with_logits(logits, label)
            loss.backward()
            opt.step()
            opt.zero_grad()
            losses.append(float(loss))
    return losses


def kg_retrain_minibatch(model, data, node_sets, num_edge_type, epochs, lr):
    """KGRetrainTrainer.train (framework/trainer/retrain.py:235-339) without validation: per GraphSAINT batch, message
    passing and positives on the batch's Dr edges ONLY (the deleted triples never enter), DistMult scores of the
    forward-direction types against per-relation head-shuffled negatives (global torch RNG), BCE-with-logits, gradient
    norm clipped to 1 (:285), Adam on every parameter.  -> per-step train_loss."""
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    losses = []
    for _ in range(epochs):
        model.train()
        for nodes in node_sets:
            b = pyg.saint_subgraph(dict(data), nodes)
            keep = b['dr_mask']
            ei, et = b['edge_index'][:, keep], b['edge_type'][keep]
            z = model(b['x'], ei, et)
            fw = et < num_edge_type
            dec, dec_t = ei[:, fw], et[fw]
            neg = negative_sampling_kg(dec, dec_t)
            logits = torch.cat([model.decode(z, dec, dec_t), model.decode(z, neg, dec_t)], -1)
            label = torch.cat([torch.ones(dec.shape[1]), torch.zeros(neg.shape[1])])
            loss = F.binary_cross_entropy_with_logits(logits, label)
            loss.backward()
            torch.nn.utils.clip_grad_norm_(model.parameters(), 1)
            opt.step()
            opt.zero_grad()
            losses.append(float(loss))
    return losses


def eval_kg(model, data, stage, unlearning_model='gnndelete_nodeemb'):
    """KGTrainer.eval (base.py:495-567): full-graph message passing on Dr, DistMult scores WITHOUT sigmoid for the
    Dt loss / AUC / AUP, sigmoid scores of the directed Df triples, and 500 FRESH Dr subsets per call
    (torch.randperm on the global RNG, :530-539) for the Df-vs-Dr AUC / AUP (Df labelled 0, Dr labelled 1)."""
    from sklearn.metrics import roc_auc_score, average_precision_score
    model.eval()
    with torch.no_grad():
        pos, neg, et = data[f'{stage}_pos_edge_index'], data[f'{stage}_neg_edge_index'], data[f'{stage}_edge_type']
        z = model(data['x'], data['edge_index'][:, data['dr_mask']], data['edge_type'][data['dr_mask']])
        logits = model.decode(z, torch.cat([pos, neg], -1), torch.cat([et, et], -1))
        label = torch.zeros(pos.shape[1] + neg.shape[1])
        label[:pos.shape[1]] = 1.0
        loss = F.binary_cross_entropy_with_logits(logits, label).item()
        dt_auc = roc_auc_score(label.numpy(), logits.numpy())
        dt_aup = average_precision_score(label.numpy(), logits.numpy())
        if unlearning_model == 'original':
            df_logit = []
        else:
            df_logit = model.decode(z, data['directed_df_edge_index'], data['directed_df_edge_type']).sigmoid().tolist()
        half = data['dr_mask'][:data['dr_mask'].shape[0] // 2]
        if df_logit:
            dr_e, dr_t = data['train_pos_edge_index'][:, half], data['train_edge_type'][half]
            aucs, aups = [], []
            lab = [0] * len(df_logit) + [1] * len(df_logit)
            for _ in range(500):
                m = torch.zeros(dr_e.shape[1], dtype=torch.bool)
                m[torch.randperm(dr_e.shape[1])[:len(df_logit)]] = True
                pl = model.decode(z, dr_e[:, m], dr_t[m]).sigmoid().tolist()
                aucs.append(roc_auc_score(lab, df_logit + pl))
                aups.append(average_precision_score(lab, df_logit + pl))
            df_auc, df_aup = float(np.mean(aucs)), float(np.mean(aups))
        else:
            df_auc = df_aup = float('nan')
    return {'loss': loss, 'dt_auc': dt_auc, 'dt_aup': dt_aup, 'df_auc': df_auc, 'df_aup': df_aup, 'df_logit': df_logit}


# ----------------------------------------------------------------------------
# retrain baseline (retrain.py:57-131), verification error (evaluation.py:63-81)
# ----------------------------------------------------------------------------
def retrain_fullbatch(model, data, epochs, lr, negs):
    """RetrainTrainer.train_fullbatch without validation: BCE-with-logits link prediction on the Dr edges only,
    fresh negatives per epoch (``negs``, consumed in order), one Adam over every parameter, zero_grad after the
    step.  -> per-epoch train_loss."""
    E = data['train_pos_edge_index'][:, data['dr_mask']]
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    losses = []
    for ep in range(epochs):
        model.train()
        neg = negs[ep]
        z = model(data['x'], E)
        label = torch.cat([torch.ones(E.shape[1]), torch.zeros(neg.shape[1])])
        loss = F.binary_cross_entropy_with_logits(model.decode(z, E, neg), label)
        loss.backward()
        opt.step()
        opt.zero_grad()
        losses.append(float(loss))
    return losses


def verification_error(model1, model2):
    """Sum over the parameters both models have (by name) of the L2 norm of their difference."""
    p1, p2 = dict(model1.named_parameters()), dict(model2.named_parameters())
    diff = torch.tensor(0.0)
    with torch.no_grad():
        for n in set(p1) & set(p2):
            diff += torch.norm(p1[n] - p2[n])
    return float(diff)


# ----------------------------------------------------------------------------
# dataset split (prepare_dataset.py:31-136) and Df candidate masks (:205-214)
# ----------------------------------------------------------------------------
def split_edges(edge_index, num_nodes, val_ratio=0.05, test_ratio=0.1, two_hop_degree=None, kg=False, edge_type=None):
    """train_test_split_edges_no_neg_adj_mask: keep ``row < col`` (not for KGs), permute with torch.randperm on the
    GLOBAL RNG (low-two-hop-degree edges first when ``two_hop_degree`` is given, :54-64), TEST edges first, then
    validation, the rest is training.  Upstream slices the KG edge TYPES without applying the permutation
    (:79, :104, :119) - reproduced.  Negatives are sampled by the caller.  -> dict."""
    row, col = edge_index
    if not kg:
        m = row < col
        row, col = row[m], col[m]
    n_v, n_t = int(math.floor(val_ratio * row.shape[0])), int(math.floor(test_ratio * row.shape[0]))
    if two_hop_degree is not None:
        low_mask = two_hop_degree < 50
        low, high = low_mask.nonzero().squeeze(), (~low_mask).nonzero().squeeze()
        low = low[torch.randperm(low.shape[0])]
        high = high[torch.randperm(high.shape[0])]
        perm = torch.cat([low, high])
    else:
        perm = torch.randperm(row.shape[0])
    row, col = row[perm], col[perm]
    out = {'train': torch.stack([row[n_v + n_t:], col[n_v + n_t:]]), 'test': torch.stack([row[:n_t], col[:n_t]]),
           'val': torch.stack([row[n_t:n_t + n_v], col[n_t:n_t + n_v]])}
    if kg:
        out.update(train_type=edge_type[n_v + n_t:], test_type=edge_type[:n_t], val_type=edge_type[n_t:n_t + n_v])
    _, _, local = pyg.k_hop_subgraph(out['test'].flatten().unique(), 2, out['train'], num_nodes)
    out['in_mask'] = local
    return out


def process_kg_ogbl(split_edge, num_nodes_dict, rev_offset=51):
    """The ogbl branch of process_kg (prepare_dataset.py:300-353): OGB hands out per-type local entity ids; the global id
    is the local one plus the running offset of its type in num_nodes_dict order (:305-310, :326-333); validation / test
    negatives = (head, FIRST corrupted tail as OGB stores it) (:337, :341); of the training triples, those between two
    entity types are kept as they are, those inside one type only once (head < tail) and appended after the others
    (:343-346); every training triple gets an inverse with relation + 51 (:351-355, the constant is upstream's).
    -> dict with the fields of the Data object upstream pickles, plus the IN candidate mask (:383-393)."""
    offset, cur = {}, 0
    for key in num_nodes_dict:
        offset[key] = cur
        cur += num_nodes_dict[key]
    n_entity = cur

    def to_global(d):
        head = [int(i) + offset[t] for i, t in zip(d['head'], d['head_type'])]
        tail = [int(i) + offset[t] for i, t in zip(d['tail'], d['tail_type'])]
        return head, tail

    out = {}
    for name, key in (('val', 'valid'), ('test', 'test')):
        d = split_edge[key]
        head, tail = to_global(d)
        pos = torch.tensor([head, tail])
        out[f'{name}_pos_edge_index'] = pos
        out[f'{name}_edge_type'] = torch.as_tensor(np.asarray(d['relation']))
        out[f'{name}_neg_edge_index'] = torch.stack([pos[0], torch.as_tensor(np.asarray(d['tail_neg']))[:, 0]])
    d = split_edge['train']
    head, tail = to_global(d)
    directed, undirected = [], []
    for h, t, r, ht, tt in zip(head, tail, d['relation'], d['head_type'], d['tail_type']):
        if ht != tt:
            directed.append((h, t, int(r)))
        elif h < t:
            undirected.append((h, t, int(r)))
    uni = directed + undirected
    train = torch.tensor([[u[0] for u in uni], [u[1] for u in uni]])
    train_type = torch.tensor([u[2] for u in uni])
    out.update(x=torch.arange(n_entity), train_pos_edge_index=train, train_edge_type=train_type,
               edge_index=torch.cat([train, train.flip(0)], 1), edge_type=torch.cat([train_type, train_type + rev_offset]))
    _, _, local = pyg.k_hop_subgraph(out['test_pos_edge_index'].flatten().unique(), 2, train, n_entity)
    out['in_mask'] = local
    return out


def df_size_from_arg(df_size, num_train_edges):
    """delete_gnn.py:88-91."""
    return int(df_size) if df_size >= 100 else int(df_size / 100 * num_train_edges)
