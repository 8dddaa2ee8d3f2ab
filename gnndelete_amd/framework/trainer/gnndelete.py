"""GNNDeleteTrainer - the edge-probability variant (`--unlearning_model gnndelete`, reference:
framework/trainer/gnndelete.py:138-309):

  loss_r = MSE( logit(Df edges), logit(fresh negatives) )            (per-epoch negatives :221-225)
  loss_l = MSE( sigmoid(z z^T)[P], sigmoid(logits_ori)[P] )          (:239-241)
  P = strictly-lower-triangular pairs of 2-hop S_Df nodes minus the Df pairs (:174-193)
  loss = 0.5 loss_r + 0.5 loss_l, single Adam, zero_grad after the step (:249-255)

Upstream materialises z z^T over ALL N x N node pairs plus an N x N boolean mask on the CPU and
indexes it every epoch.  Only the |S| x |S| block of S_Df nodes is ever read, and even that block is
never formed here: gd_pairs_sigmoid_mse_f32 (csrc/pairs.hip) computes the loss and its gradient tile
by tile on the matrix cores; the original probabilities of the included pairs are gathered once into
a dense [|S|, |S|] target (negative = pair excluded).  Autograd runs through the HIP-backed model."""
import os
import time

import torch
import torch.nn.functional as F

from ... import ops
from ..graph_utils import negative_sampling
from .base import Trainer, _require_gpu, device, is_large


def sdf_pair_mask(num_nodes, sdf_node_mask, df_edges):
    """(nodes S sorted, [|S|,|S|] bool mask of the pairs i > j of S that are not Df edges)."""
    nodes = sdf_node_mask.nonzero().flatten()
    s = nodes.numel()
    mask = torch.ones(s, s, dtype=torch.bool, device=nodes.device).tril_(-1)
    pos = torch.full((num_nodes,), -1, dtype=torch.long, device=nodes.device)
    pos[nodes] = torch.arange(s, device=nodes.device)
    a, b = pos[df_edges[0]], pos[df_edges[1]]
    ok = (a >= 0) & (b >= 0)
    mask[a[ok], b[ok]] = False
    mask[b[ok], a[ok]] = False
    return nodes, mask


class GNNDeleteTrainer(Trainer):

    def train(self, model, data, optimizer, args, logits_ori=None, attack_model_all=None, attack_model_sub=None):
        # upstream sends ogbl-* / Physics to the mini-batch loop (gnndelete.py:130-136) to fit its GPUs; one MI355X holds
        # these graphs whole, so - as for the other trainers - it is taken on request (--minibatch)
        if is_large(self.args.dataset) and getattr(args, 'minibatch', False):
            return self.train_minibatch(model, data, optimizer, args, logits_ori, attack_model_all, attack_model_sub)
        return self.train_fullbatch(model, data, optimizer, args, logits_ori, attack_model_all, attack_model_sub)

    def train_minibatch(self, model, data, optimizer, args, logits_ori=None, attack_model_all=None, attack_model_sub=None):
        """The edge-probability mini-batch loop (framework/trainer/gnndelete.py:312-450) on GraphSAINT random-walk batches
        cut on the device.  Upstream this loop cannot run: Trainer.get_embedding reads data.dtrain_mask (base.py:59), which
        delete_gnn.py never sets (:124, :190 commented out).  Here dtrain_mask defaults to dr_mask - what those lines
        assigned - and the loop is the reference's, quirks included: z_ori (the model's embedding of the whole graph, once,
        under no_grad) is indexed with the BATCH-LOCAL node ids in the locality term (:378-386), and the epoch log divides
        by the last enumerate index twice with the two terms' names swapped (:401-423).  Pinned by
        tests/golden/traj_edgeprob_minibatch_*.npz (the reference's loop run with dtrain_mask injected)."""
        from . import sampler as S
        _require_gpu()
        # the loop's variants (gnndelete.py:314-317, :391-398): BoundedKLD for '...kld...' models, the two ablations by name
        um = getattr(args, 'unlearning_model', 'gnndelete')
        if 'kld' in um:
            loss_fct = lambda logits, truth: 1 - torch.exp(-F.kl_div(F.log_softmax(logits, -1), truth.softmax(-1), None, None, 'batchmean'))
        else:
            loss_fct = F.mse_loss
        if attack_model_all is not None or attack_model_sub is not None:
            raise NotImplementedError('membership-inference attacks before / after unlearning (gnndelete.py:320-328) are outside the '
                                      'Del path (SURVEY.md section 2): pass attack_model_all = attack_model_sub = None')
        model = model.to(device)
        data = data.to('cpu')
        if not hasattr(data, 'dtrain_mask'):
            data.dtrain_mask = data.dr_mask
        with torch.no_grad():
            dd = data.clone().to(device) if hasattr(data, 'clone') else data.to(device)
            z_ori = self.get_embedding(model, dd)
        data.edge_index = data.train_pos_edge_index
        data.node_id = torch.arange(data.x.shape[0])
        loader = S.make_sampler(data, args.batch_size, args.num_steps)
        best_metric = 0
        self.trainer_log['steps'] = []
        for epoch in range(args.epochs):
            model.train()
            step_logs = []
            start = time.time()
            for batch in loader:
                batch = batch.to(device)
                ei = batch.edge_index
                e_sdf = ei[:, batch.sdf_mask].contiguous()
                z = model(batch.x, e_sdf, batch.sdf_node_1hop_mask, batch.sdf_node_2hop_mask)
                pos = ei[:, batch.df_mask]
                k = pos.shape[1]
                neg = S.negative_sampling(ei, batch.x.shape[0], k)
                df_logits = model.decode(z, pos, neg)
                loss_e = loss_fct(df_logits[:k], df_logits[k:])
                lower = e_sdf[0] < e_sdf[1]
                row, col = e_sdf[0][lower], e_sdf[1][lower]
                logits_ori = (z_ori[row] * z_ori[col]).sum(-1)         # (batch-local ids into the global embedding: upstream)
                logits = (z[row] * z[col]).sum(-1)
                loss_l = loss_fct(logits, logits_ori)
                if 'ablation_random' in um:                            # (gnndelete.py:391-398)
                    loss_l = torch.tensor(0)
                    loss = loss_e
                elif 'ablation_locality' in um:
                    loss_e = torch.tensor(0)
                    loss = loss_l
                else:
                    loss = 0.5 * loss_e + 0.5 * loss_l
                loss.backward()
                S.sync_gradients(optimizer)
                optimizer.step()
                optimizer.zero_grad()
                step_logs.append({'loss': loss.item(), 'loss_e': loss_e.item(), 'loss_l': loss_l.item()})
            self.trainer_log['steps'].extend(step_logs)
            if (epoch + 1) % args.valid_freq == 0:
                valid_loss, dt_auc, dt_aup, df_auc, df_aup, df_logit, _, valid_log = self.eval(model, data, 'val')
                step = max(len(step_logs) - 1, 1)                       # upstream divides by the last enumerate index, twice
                tot = {k_: sum(s_[k_] for s_ in step_logs) / step for k_ in ('loss', 'loss_e', 'loss_l')}
                self._record({'epoch': epoch, 'train_loss': tot['loss'] / step, 'train_loss_l': tot['loss_e'] / step,
                              'train_loss_e': tot['loss_l'] / step, 'train_time': (time.time() - start) / step / step}, valid_log)
                if dt_auc + df_auc > best_metric:
                    best_metric = dt_auc + df_auc
                    print(f'Save best checkpoint at epoch {epoch:04d}. Valid loss = {valid_loss:.4f}')
                    torch.save({'model_state': model.state_dict(), 'optimizer_state': optimizer.state_dict()},
                               os.path.join(args.checkpoint_dir, 'model_best.pt'))
                    torch.save(z.detach(), os.path.join(args.checkpoint_dir, 'node_embeddings.pt'))
                data = data.to('cpu')
        torch.save({'model_state': {k_: v.to('cpu') for k_, v in model.state_dict().items()},
                    'optimizer_state': optimizer.state_dict()}, os.path.join(args.checkpoint_dir, 'model_final.pt'))

    def train_fullbatch(self, model, data, optimizer, args, logits_ori=None, attack_model_all=None,
                        attack_model_sub=None):
        _require_gpu()
        model = model.to(device)
        data = data.to(device)
        edges = data.train_pos_edge_index
        e_sdf = edges[:, data.sdf_mask].contiguous()
        df_edges = edges[:, data.df_mask]
        nodes, pair_mask = sdf_pair_mask(data.num_nodes, data.sdf_node_2hop_mask, df_edges)
        target, n_pairs = None, int(pair_mask.sum())
        if n_pairs:
            if logits_ori is None:
                raise ValueError('GNNDeleteTrainer needs logits_ori (pred_proba.pt of the original model)')
            ori = logits_ori.to(device) if logits_ori.device != nodes.device else logits_ori
            target = ori[nodes][:, nodes].float().sigmoid()
            target.masked_fill_(~pair_mask, -1.0)           # excluded: upper triangle, diagonal, Df pairs
            s_pad = (target.shape[0] + 3) // 4 * 4          # 16-byte aligned rows for the kernel's float4 loads
            if s_pad != target.shape[0]:
                target = F.pad(target, (0, s_pad - target.shape[0]), value=-1.0)
            target = target.contiguous()
            nodes32 = nodes.to(torch.int32).contiguous()
        del pair_mask
        neg_size = int(data.df_mask.sum())
        best_metric = 0
        for epoch in range(args.epochs):
            model.train()
            start = time.time()
            z = model(data.x, e_sdf)
            neg = negative_sampling(edge_index=edges, num_nodes=data.num_nodes, num_neg_samples=neg_size)
            df_logits = model.decode(z, df_edges, neg)
            loss_r = F.mse_loss(df_logits[:neg_size], df_logits[neg_size:])
            if target is not None:
                loss_l = ops.pairs_sigmoid_mse(z, nodes32, target, n_pairs)
            else:
                loss_l = torch.tensor(0.0, device=device)
            loss = 0.5 * loss_r + 0.5 * loss_l
            loss.backward()
            optimizer.step()
            optimizer.zero_grad()
            if (epoch + 1) % self.args.valid_freq == 0:
                valid_loss, dt_auc, dt_aup, df_auc, df_aup, df_logit, _, valid_log = self.eval(model, data, 'val')
                valid_log['epoch'] = epoch
                self._record({'epoch': epoch, 'train_loss': loss.item(), 'train_loss_l': loss_l.item(),
                              'train_loss_r': loss_r.item(), 'train_time': time.time() - start}, valid_log)
                if dt_auc + df_auc > best_metric:
                    best_metric = dt_auc + df_auc
                    print(f'Save best checkpoint at epoch {epoch:04d}. Valid loss = {valid_loss:.4f}')
                    torch.save({'model_state': model.state_dict(), 'optimizer_state': optimizer.state_dict()},
                               os.path.join(args.checkpoint_dir, 'model_best.pt'))
        torch.save({'model_state': {k: v.to('cpu') for k, v in model.state_dict().items()},
                    'optimizer_state': optimizer.state_dict()}, os.path.join(args.checkpoint_dir, 'model_final.pt'))
