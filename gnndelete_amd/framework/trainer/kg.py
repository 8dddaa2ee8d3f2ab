"""Knowledge-graph trainers for the R-GCN path (reference: KGTrainer base.py:394-692,
KGGNNDeleteNodeembTrainer gnndelete_nodeemb.py:659-846).  Mini-batches of random-walk subgraphs
(.sampler), message passing on the relation-typed HIP kernels (per-relation mean + block-diagonal /
dense relation weights), DistMult decoder, per-relation head-shuffled negatives."""
import os
import time

import numpy as np
import torch
import torch.nn.functional as F

from ..evaluation import verification_error
from ..metrics import batched_average_precision, batched_roc_auc
from ..utils import get_link_labels, negative_sampling_kg
from .base import Trainer, _all_pairs, _require_gpu, device
from .gnndelete_nodeemb import _four_terms, _non_df_masks, get_loss_fct
from . import sampler as _sampler
from ._log import wandb_log


class KGTrainer(Trainer):
    def train(self, model, data, optimizer, args):
        """BCE link prediction with DistMult on random-walk subgraphs of 128 roots (base.py:394-493);
        only forward-direction relation types are decoded; model selection on validation AUP."""
        _require_gpu()
        model = model.to(device)
        data = data.to('cpu')
        loader = _sampler.make_sampler(data, 128, args.num_steps)
        best_metric, best_epoch = 0, 0
        start = time.time()
        self.trainer_log['steps'] = []
        for epoch in range(args.epochs):
            model.train()
            epoch_loss, steps = 0.0, 0
            for batch in loader:
                batch = batch.to(device)
                edge_index, edge_type = batch.edge_index.contiguous(), batch.edge_type.contiguous()
                z = model(batch.x, edge_index, edge_type)
                decoding = edge_type < args.num_edge_type
                dec_index, dec_type = edge_index[:, decoding], edge_type[decoding]
                neg_index = negative_sampling_kg(edge_index=dec_index, edge_type=dec_type)
                logits = torch.cat([model.decode(z, dec_index, dec_type), model.decode(z, neg_index, dec_type)], dim=-1)
                loss = F.binary_cross_entropy_with_logits(logits, get_link_labels(dec_index, neg_index))
                loss.backward()
                optimizer.step()
                optimizer.zero_grad()
                rec = {'epoch': epoch, 'step': steps, 'train_loss': loss.item()}
                wandb_log(rec)
                self.trainer_log['steps'].append(rec)
                epoch_loss += rec['train_loss']
                steps += 1
            if (epoch + 1) % args.valid_freq == 0:
                valid_loss, dt_auc, dt_aup, df_auc, df_aup, df_logit, _, valid_log = self.eval(model, data, 'val')
                self._record({'epoch': epoch, 'train_loss': epoch_loss / max(steps - 1, 1)}, valid_log)
                data = data.to('cpu')
                if dt_aup > best_metric:
                    best_metric, best_epoch = dt_aup, epoch
                    print(f'Save best checkpoint at epoch {epoch:04d}. Valid loss = {valid_loss:.4f}')
                    torch.save({'model_state': model.state_dict(), 'optimizer_state': optimizer.state_dict()},
                               os.path.join(args.checkpoint_dir, 'model_best.pt'))
        self.trainer_log['training_time'] = time.time() - start
        torch.save({'model_state': model.state_dict(), 'optimizer_state': optimizer.state_dict()},
                   os.path.join(args.checkpoint_dir, 'model_final.pt'))
        self.trainer_log['best_epoch'], self.trainer_log['best_metric'] = best_epoch, best_metric

    @torch.no_grad()
    def eval(self, model, data, stage='val', pred_all=False):
        """KGTrainer.eval (base.py:495-567): full-graph message passing on Dr, DistMult scores
        WITHOUT sigmoid for the Dt metrics, 500 fresh Dr subsets per call for the Df metrics."""
        _require_gpu()
        model.eval()
        model = model.to(device)
        data = data.to(device)
        pos, neg = data[f'{stage}_pos_edge_index'], data[f'{stage}_neg_edge_index']
        etype = data[f'{stage}_edge_type']
        z = model(data.x, data.edge_index[:, data.dr_mask].contiguous(), data.edge_type[data.dr_mask].contiguous())
        logits = model.decode(z, torch.cat([pos, neg], dim=-1), torch.cat([etype, etype], dim=-1))
        label = get_link_labels(pos, neg)
        loss = F.binary_cross_entropy_with_logits(logits, label).cpu().item()
        dt_auc = float(batched_roc_auc(logits, label)[0])
        dt_aup = float(batched_average_precision(logits, label)[0])

        if self.args.unlearning_model in ['original']:
            df_logit = []
        else:
            df_logit = model.decode(z, data.directed_df_edge_index, data.directed_df_edge_type).sigmoid().tolist()
        if len(df_logit) > 0:
            half = data.dr_mask[:data.dr_mask.shape[0] // 2]
            dr_edges, dr_types = data.train_pos_edge_index[:, half], data.train_edge_type[half]
            k = len(df_logit)
            dr_score = model.decode(z, dr_edges, dr_types).sigmoid()
            n_dr = dr_edges.shape[1]
            if n_dr <= self.FAST_SUBSETS_ABOVE:
                # upstream's 500 fresh host permutations per evaluation (base.py:530-539)
                picks = torch.stack([torch.randperm(n_dr)[:k].sort().values for _ in range(500)])
            else:
                # ogbl-biokg size: the same statistic from device-side permutations (see Trainer._ensure_df_subsets)
                gen = torch.Generator(device=dr_score.device)
                gen.manual_seed(int(torch.randint(0, 2 ** 31 - 1, (1,))))
                picks = torch.stack([torch.randperm(n_dr, device=dr_score.device, generator=gen)[:k] for _ in range(500)])
            df_score = torch.tensor(df_logit, dtype=dr_score.dtype, device=dr_score.device)
            scores = torch.cat([df_score[None].expand(500, k), dr_score[picks.to(dr_score.device)]], dim=1)
            labels = torch.cat([torch.zeros(k), torch.ones(k)]).to(dr_score.device)
            df_auc = float(batched_roc_auc(scores, labels).mean())
            df_aup = float(batched_average_precision(scores, labels).mean())
        else:
            df_auc = df_aup = np.nan
        logit_all_pair = _all_pairs(z) if pred_all else None
        log = {f'{stage}_loss': loss, f'{stage}_dt_auc': dt_auc, f'{stage}_dt_aup': dt_aup, f'{stage}_df_auc': df_auc,
               f'{stage}_df_aup': df_aup,
               f'{stage}_df_logit_mean': np.mean(df_logit) if len(df_logit) > 0 else np.nan,
               f'{stage}_df_logit_std': np.std(df_logit) if len(df_logit) > 0 else np.nan}
        return loss, dt_auc, dt_aup, df_auc, df_aup, df_logit, logit_all_pair, log

    @torch.no_grad()
    def test(self, model, data, model_retrain=None, attack_model_all=None, attack_model_sub=None, ckpt='best'):
        if ckpt == 'best':
            state = torch.load(os.path.join(self.args.checkpoint_dir, 'model_best.pt'), map_location='cpu')
            model.load_state_dict(state['model_state'])
        loss, dt_auc, dt_aup, df_auc, df_aup, df_logit, logit_all_pair, test_log = self.eval(model, data, 'test', False)
        self.logit_all_pair = logit_all_pair
        self.trainer_log.update({'dt_loss': loss, 'dt_auc': dt_auc, 'dt_aup': dt_aup, 'df_logit': df_logit,
                                 'df_auc': df_auc, 'df_aup': df_aup, 'auc_sum': dt_auc + df_auc,
                                 'aup_sum': dt_aup + df_aup, 'auc_gap': abs(dt_auc - df_auc),
                                 'aup_gap': abs(dt_aup - df_aup)})
        if model_retrain is not None:
            self.trainer_log['ve'] = verification_error(model, model_retrain).cpu().item()
        return loss, dt_auc, dt_aup, df_auc, df_aup, df_logit, logit_all_pair, test_log


class KGGNNDeleteNodeembTrainer(KGTrainer):
    def train(self, model, data, optimizer, args, logits_ori=None, attack_model_all=None, attack_model_sub=None):
        """gnndelete_nodeemb.py:659-846: per batch, message passing on the batch's Dr edges with the
        S_Df-minus-Df node masks as Del masks, DEC on the forward-direction Df triples vs
        per-relation head-shuffled negatives, NI on the masked nodes, layer-wise update."""
        _require_gpu()
        if getattr(args, 'fullgraph', False):
            return self.train_fullgraph(model, data, optimizer, args)
        loss_fct = get_loss_fct(self.args.loss_fct)
        model = model.to(device)
        data = data.to('cpu')
        _non_df_masks(data)
        loader = _sampler.make_sampler(data, args.batch_size, args.num_steps)
        best_metric = 0
        alpha = self.args.alpha
        self.trainer_log['steps'] = []
        for epoch in range(args.epochs):
            model.train()
            last = None
            for batch in loader:
                batch = batch.to(device)
                edge_index = batch.edge_index[:, batch.dr_mask].contiguous()
                edge_type = batch.edge_type[batch.dr_mask].contiguous()
                m1, m2 = batch.sdf_node_1hop_mask_non_df_mask, batch.sdf_node_2hop_mask_non_df_mask
                z1, z2 = model(batch.x, edge_index, edge_type, m1, m2, return_all_emb=True)
                with torch.no_grad():
                    z1_ori, z2_ori = model.get_original_embeddings(batch.x, edge_index, edge_type, return_all_emb=True)
                pos_index, pos_type = batch.edge_index[:, batch.df_mask], batch.edge_type[batch.df_mask]
                forward = pos_type < self.args.num_edge_type
                dec_index, dec_type = pos_index[:, forward], pos_type[forward]
                neg_index = negative_sampling_kg(edge_index=dec_index, edge_type=dec_type)
                r1, r2, l1, l2 = _four_terms(loss_fct, z1, z2, z1_ori, z2_ori, dec_index, neg_index, m1, m2)
                loss1 = alpha * r1 + (1 - alpha) * l1
                loss1.backward(retain_graph=True)
                _sampler.sync_gradients(optimizer[0])
                optimizer[0].step()
                optimizer[0].zero_grad()
                loss2 = alpha * r2 + (1 - alpha) * l2
                loss2.backward(retain_graph=True)
                _sampler.sync_gradients(optimizer[1])
                optimizer[1].step()
                optimizer[1].zero_grad()
                last = (loss1 + loss2, r1 + r2, l1 + l2)
                step_log = {'Epoch': epoch, 'train_loss': last[0].item(), 'loss_r': last[1].item(), 'loss_l': last[2].item()}
                wandb_log(step_log)
                self.trainer_log['steps'].append(step_log)
            if (epoch + 1) % self.args.valid_freq == 0 and last is not None:
                valid_loss, dt_auc, dt_aup, df_auc, df_aup, df_logit, _, valid_log = self.eval(model, data, 'val')
                valid_log['epoch'] = epoch
                self._record({'epoch': epoch, 'train_loss': last[0].item(), 'loss_r': last[1].item(),
                              'loss_l': last[2].item()}, valid_log)
                if dt_auc + df_auc > best_metric:
                    best_metric = dt_auc + df_auc
                    print(f'Save best checkpoint at epoch {epoch:04d}. Valid loss = {valid_loss:.4f}')
                    torch.save({'model_state': model.state_dict()}, os.path.join(args.checkpoint_dir, 'model_best.pt'))
                data = data.to('cpu')
        torch.save({'model_state': {k: v.to('cpu') for k, v in model.state_dict().items()}},
                   os.path.join(args.checkpoint_dir, 'model_final.pt'))

    def train_fullgraph(self, model, data, optimizer, args):
        """--fullgraph: the same objective as the batch loop above, on the WHOLE graph with the fused R-GCN step
        (gnndelete_amd.engine.NodeembEngine, mode rgcn: typed tile conv, Del, folded losses, Del-weight gradients and
        Adam in one hipGraph per epoch) - message passing on the Dr triples with the S_Df-minus-Df node masks as Del
        masks, DEC on the forward-direction Df triples against head-shuffled negatives drawn ONCE (upstream redraws
        them per batch), NI on the masked nodes.  Upstream has no full-graph KG loop: ogbl-biokg does not fit its GPUs."""
        from ...engine import NodeembEngine
        from .gnndelete_nodeemb import _adam_hyper, _export_adam_state
        if self.args.loss_fct not in ('mse_mean', 'mse_sum'):
            raise NotImplementedError('--fullgraph needs --loss_fct mse_mean | mse_sum (the fused step folds the MSE terms)')
        model = model.to(device)
        data = data.to(device)
        _non_df_masks(data)
        ei = data.edge_index[:, data.dr_mask].contiguous()
        et = data.edge_type[data.dr_mask].contiguous()
        m1, m2 = data.sdf_node_1hop_mask_non_df_mask, data.sdf_node_2hop_mask_non_df_mask
        with torch.no_grad():
            z1_ori, z2_ori = model.get_original_embeddings(data.x, ei, et, return_all_emb=True)
        pos, pos_type = data.edge_index[:, data.df_mask], data.edge_type[data.df_mask]
        forward = pos_type < self.args.num_edge_type
        dec = pos[:, forward].contiguous()
        neg = negative_sampling_kg(edge_index=dec, edge_type=pos_type[forward])
        lr, betas, eps = _adam_hyper(optimizer)
        engine = NodeembEngine(model, data.x, ei, z1_ori, z2_ori, dec, neg, m1, m2, loss_type=self.args.loss_type,
                               alpha=self.args.alpha, lr=lr, reduction='mean' if self.args.loss_fct == 'mse_mean' else 'sum',
                               mask_1hop=m1, mask_2hop=m2, history=max(16, args.epochs), edge_type=et,
                               # as the link-prediction trainer: the frozen conv1 output is computed once and conv2's input
                               # gradient only on the Del-1 rows that read it (identical Del weights; --no_layer1_cache /
                               # --all_rows restore the epoch as upstream would run it)
                               cache_layer1=not getattr(args, 'no_layer1_cache', False),
                               affected_rows_only=not getattr(args, 'all_rows', False))
        engine.adam1.betas = engine.adam2.betas = betas
        engine.adam1.eps = engine.adam2.eps = eps
        best_metric = 0
        for epoch in range(args.epochs):
            model.train()
            engine.step()
            if (epoch + 1) % self.args.valid_freq == 0:
                last = engine.loss_history()[-1]
                valid_loss, dt_auc, dt_aup, df_auc, df_aup, df_logit, _, valid_log = self.eval(model, data, 'val')
                valid_log['epoch'] = epoch
                self._record({'epoch': epoch, 'train_loss': float(last[0]), 'loss_r': float(last[1]), 'loss_l': float(last[2])},
                             valid_log)
                if dt_auc + df_auc > best_metric:
                    best_metric = dt_auc + df_auc
                    print(f'Save best checkpoint at epoch {epoch:04d}. Valid loss = {valid_loss:.4f}')
                    torch.save({'model_state': model.state_dict()}, os.path.join(args.checkpoint_dir, 'model_best.pt'))
                data = data.to(device)
        _export_adam_state(engine, model, optimizer)
        self.trainer_log['loss_history'] = engine.loss_history().tolist()
        torch.save({'model_state': {k: v.to('cpu') for k, v in model.state_dict().items()}},
                   os.path.join(args.checkpoint_dir, 'model_final.pt'))
