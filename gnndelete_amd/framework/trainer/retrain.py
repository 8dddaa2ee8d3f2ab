"""Retrain-from-scratch baseline on the retained edges Dr (reference: framework/trainer/retrain.py,
RetrainTrainer :19-232, KGRetrainTrainer :235-339): the comparison column of the paper's tables and the model
``Trainer.test`` measures the verification error against (delete_gnn.py:262-279, framework/evaluation.py:63-81).

Differences to original-model training (base.py:75-142) that matter for parity: message passing, positives and the
number of negatives all come from ``train_pos_edge_index[:, dr_mask]`` (the deleted edges never enter), and the best
checkpoint is chosen by ``dt_auc + df_auc`` of the validation pass, not by the validation loss."""
import os
import time

import torch
import torch.nn.functional as F

from ..graph_utils import negative_sampling
from ..utils import get_link_labels, negative_sampling_kg
from . import sampler as _sampler
from ._log import wandb_log
from .base import Trainer, _require_gpu, device
from .kg import KGTrainer


class RetrainTrainer(Trainer):
    def train(self, model, data, optimizer, args, logits_ori=None, attack_model_all=None, attack_model_sub=None):
        # upstream mini-batches ogbl-* graphs to fit its GPUs (retrain.py:133-232); one MI355X holds them whole
        return self.train_fullbatch(model, data, optimizer, args, logits_ori, attack_model_all, attack_model_sub)

    def train_fullbatch(self, model, data, optimizer, args, logits_ori=None, attack_model_all=None,
                        attack_model_sub=None):
        """retrain.py:39-131: BCE link prediction on Dr, fresh negatives every epoch, Adam on every parameter."""
        _require_gpu()
        model = model.to(device)
        data = data.to(device)
        edges = data.train_pos_edge_index[:, data.dr_mask].contiguous()
        n_neg = int(data.dr_mask.sum())
        best_metric, best_epoch = 0, 0
        start = time.time()
        self.trainer_log['steps'] = []
        for epoch in range(args.epochs):
            model.train()
            neg = negative_sampling(edge_index=edges, num_nodes=data.num_nodes, num_neg_samples=n_neg)
            z = model(data.x, edges)
            logits = model.decode(z, edges, neg)
            loss = F.binary_cross_entropy_with_logits(logits, get_link_labels(edges, neg))
            loss.backward()
            optimizer.step()
            optimizer.zero_grad()
            step_log = {'Epoch': epoch, 'train_loss': loss.item()}
            wandb_log(step_log)
            self.trainer_log['steps'].append(step_log)
            if (epoch + 1) % self.args.valid_freq == 0:
                valid_loss, dt_auc, dt_aup, df_auc, df_aup, df_logit, _, valid_log = self.eval(model, data, 'val')
                valid_log['epoch'] = epoch
                self._record({'epoch': epoch, 'train_loss': step_log['train_loss']}, valid_log)
                if dt_auc + df_auc > best_metric:
                    best_metric, best_epoch = dt_auc + df_auc, epoch
                    print(f'Save best checkpoint at epoch {epoch:04d}. Valid loss = {valid_loss:.4f}')
                    torch.save({'model_state': model.state_dict(), 'optimizer_state': optimizer.state_dict()},
                               os.path.join(args.checkpoint_dir, 'model_best.pt'))
        self.trainer_log['training_time'] = time.time() - start
        torch.save({'model_state': model.state_dict(), 'optimizer_state': optimizer.state_dict()},
                   os.path.join(args.checkpoint_dir, 'model_final.pt'))
        self.trainer_log['best_epoch'], self.trainer_log['best_metric'] = best_epoch, best_metric


class KGRetrainTrainer(KGTrainer):
    def train(self, model, data, optimizer, args, logits_ori=None, attack_model_all=None, attack_model_sub=None):
        """retrain.py:235-339: DistMult link prediction on random-walk subgraphs of 128 roots, message passing and
        positives restricted to the batch's Dr edges (forward-direction types only are decoded), gradient norm
        clipped to 1, model selection on dt_auc + df_auc."""
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
                edge_index = batch.edge_index[:, batch.dr_mask].contiguous()
                edge_type = batch.edge_type[batch.dr_mask].contiguous()
                z = model(batch.x, edge_index, edge_type)
                decoding = edge_type < args.num_edge_type
                dec_index, dec_type = edge_index[:, decoding], edge_type[decoding]
                neg_index = negative_sampling_kg(edge_index=dec_index, edge_type=dec_type)
                logits = torch.cat([model.decode(z, dec_index, dec_type), model.decode(z, neg_index, dec_type)], dim=-1)
                loss = F.binary_cross_entropy_with_logits(logits, get_link_labels(dec_index, neg_index))
                loss.backward()
                torch.nn.utils.clip_grad_norm_(model.parameters(), 1)
                optimizer.step()
                optimizer.zero_grad()
                step_log = {'epoch': epoch, 'step': steps, 'train_loss': loss.item()}
                wandb_log(step_log)
                self.trainer_log['steps'].append(step_log)
                epoch_loss += step_log['train_loss']
                steps += 1
            if (epoch + 1) % args.valid_freq == 0:
                valid_loss, dt_auc, dt_aup, df_auc, df_aup, df_logit, _, valid_log = self.eval(model, data, 'val')
                self._record({'epoch': epoch, 'train_loss': epoch_loss / max(steps - 1, 1)}, valid_log)
                if dt_auc + df_auc > best_metric:
                    best_metric, best_epoch = dt_auc + df_auc, epoch
                    print(f'Save best checkpoint at epoch {epoch:04d}. Valid loss = {valid_loss:.4f}')
                    torch.save({'model_state': model.state_dict(), 'optimizer_state': optimizer.state_dict()},
                               os.path.join(args.checkpoint_dir, 'model_best.pt'))
                data = data.to('cpu')
        self.trainer_log['training_time'] = time.time() - start
        torch.save({'model_state': model.state_dict(), 'optimizer_state': optimizer.state_dict()},
                   os.path.join(args.checkpoint_dir, 'model_final.pt'))
        self.trainer_log['best_epoch'], self.trainer_log['best_metric'] = best_epoch, best_metric
