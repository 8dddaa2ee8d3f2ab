"""Base trainers: original link-prediction / node-classification training, evaluation, logs,
checkpoints.  Mirrors the public surface of the reference's framework/trainer/base.py
(Trainer :24-392, NodeClassificationTrainer :694-864): same method names and signatures, same
trainer_log / checkpoint / file formats, same evaluation protocol including its quirks
(BCE-with-logits on already-sigmoided scores :243-247; Df labelled 0 and Dr labelled 1
:274-277; 500 cached random Dr subsets :263-268; `test` stage of the node classifier reading
val_mask :768-771).  All model arithmetic runs on the HIP kernels; AUC / AUP / accuracy stay
on the host with scikit-learn exactly as upstream."""
import json
import os
import time

import numpy as np
import torch
import torch.nn.functional as F
from sklearn.metrics import accuracy_score, f1_score

from ..evaluation import verification_error
from ..graph_utils import negative_sampling
from ..metrics import batched_average_precision, batched_roc_auc
from ..training_args import is_large
from ..utils import get_link_labels
from ._log import fmt, wandb_log

device = torch.device('cuda') if torch.cuda.is_available() else torch.device('cpu')


def _all_pairs(z):
    """z z^T on the host, as Trainer.eval(pred_all=True) hands it to test() / save_log (framework/trainer/base.py:288 upstream);
    the product runs on the HIP dense kernels (ops.gram), not on torch's matmul."""
    from ... import ops
    return ops.gram(z.detach()).cpu()


def _require_gpu():
    if not torch.cuda.is_available():
        from ... import _lib
        raise _lib.GnnDeleteHipError('training/evaluation runs on the HIP kernels: a GPU is required (no CPU fallback)')


class Trainer:
    def __init__(self, args):
        self.args = args
        self.trainer_log = {'unlearning_model': args.unlearning_model, 'dataset': args.dataset, 'log': []}
        self.logit_all_pair = None
        self.df_pos_edge = []
        os.makedirs(self.args.checkpoint_dir, exist_ok=True)
        with open(os.path.join(self.args.checkpoint_dir, 'training_args.json'), 'w') as f:
            json.dump(vars(args), f)

    # ------------------------------------------------------------------ helpers
    @torch.no_grad()
    def get_link_labels(self, pos_edge_index, neg_edge_index):
        return get_link_labels(pos_edge_index, neg_edge_index)

    @torch.no_grad()
    def get_embedding(self, model, data, on_cpu=False):
        return model(data.x, data.train_pos_edge_index[:, data.dtrain_mask])

    def _record(self, *records, keep=True):
        for rec in records:
            wandb_log(rec)
            print(fmt(rec))
            if keep:
                self.trainer_log['log'].append(rec)

    # ------------------------------------------------------------------ original training
    def train(self, model, data, optimizer, args):
        # Cora / PubMed / DBLP / CS full batch; the reference's mini-batch branch (ogbl-*, Physics) exists to fit a
        # 16-32 GB GPU - one MI355X holds these graphs whole, so it is taken only on request (--minibatch)
        if is_large(self.args.dataset) and getattr(args, 'minibatch', False):
            return self.train_minibatch(model, data, optimizer, args)
        return self.train_fullbatch(model, data, optimizer, args)

    def train_minibatch(self, model, data, optimizer, args):
        """base.py:144-227: per GraphSAINT batch, BCE link prediction on the batch's edges against one negative per
        edge; best-validation-loss checkpoint."""
        from . import sampler as _sampler
        _require_gpu()
        start = time.time()
        best_valid_loss, best_epoch = 1000000, 0
        model = model.to(device)
        data = data.to('cpu')
        data.edge_index = data.train_pos_edge_index
        loader = _sampler.make_sampler(data, args.batch_size, args.num_steps)
        self.trainer_log['steps'] = []
        for epoch in range(args.epochs):
            model.train()
            epoch_loss, steps, z = 0.0, 0, None
            for batch in loader:
                edges = batch.edge_index.to(device).contiguous()
                z = model(batch.x.to(device), edges)
                neg = _sampler.negative_sampling(edges, z.size(0), edges.shape[1])
                loss = F.binary_cross_entropy_with_logits(model.decode(z, edges, neg), get_link_labels(edges, neg))
                loss.backward()
                optimizer.step()
                optimizer.zero_grad()
                rec = {'epoch': epoch, 'step': steps, 'train_loss': loss.item()}
                wandb_log(rec)
                self.trainer_log['steps'].append(rec)
                epoch_loss += rec['train_loss']
                steps += 1
            if (epoch + 1) % args.valid_freq == 0:
                valid_loss, *_, valid_log = self.eval(model, data, 'val')
                self._record({'epoch': epoch, 'train_loss': epoch_loss / max(steps - 1, 1)}, valid_log)
                if valid_loss < best_valid_loss:
                    best_valid_loss, best_epoch = valid_loss, epoch
                    print(f'Save best checkpoint at epoch {epoch:04d}. Valid loss = {valid_loss:.4f}')
                    torch.save({'model_state': model.state_dict(), 'optimizer_state': optimizer.state_dict()},
                               os.path.join(args.checkpoint_dir, 'model_best.pt'))
                    torch.save(z.detach(), os.path.join(args.checkpoint_dir, 'node_embeddings.pt'))
                data = data.to('cpu')
        self.trainer_log['training_time'] = time.time() - start
        torch.save({'model_state': model.state_dict(), 'optimizer_state': optimizer.state_dict()},
                   os.path.join(args.checkpoint_dir, 'model_final.pt'))
        self.trainer_log['best_epoch'], self.trainer_log['best_valid_loss'] = best_epoch, best_valid_loss

    def train_fullbatch(self, model, data, optimizer, args):
        """BCE link prediction on all train edges vs fresh negatives each epoch (base.py:75-142)."""
        _require_gpu()
        start = time.time()
        best_valid_loss, best_epoch = 1000000, 0
        model = model.to(device)
        data = data.to(device)
        edges = data.train_pos_edge_index
        n_neg = int(data.dtrain_mask.sum()) if hasattr(data, 'dtrain_mask') else edges.shape[1]
        for epoch in range(args.epochs):
            model.train()
            neg = negative_sampling(edges, data.num_nodes, n_neg)
            z = model(data.x, edges)
            logits = model.decode(z, edges, neg)
            loss = F.binary_cross_entropy_with_logits(logits, get_link_labels(edges, neg))
            loss.backward()
            optimizer.step()
            optimizer.zero_grad()
            if (epoch + 1) % args.valid_freq == 0:
                valid_loss, *_, valid_log = self.eval(model, data, 'val')
                self._record({'epoch': epoch, 'train_loss': loss.item()}, valid_log)
                if valid_loss < best_valid_loss:
                    best_valid_loss, best_epoch = valid_loss, epoch
                    print(f'Save best checkpoint at epoch {epoch:04d}. Valid loss = {valid_loss:.4f}')
                    torch.save({'model_state': model.state_dict(), 'optimizer_state': optimizer.state_dict()},
                               os.path.join(args.checkpoint_dir, 'model_best.pt'))
                    torch.save(z.detach(), os.path.join(args.checkpoint_dir, 'node_embeddings.pt'))
        self.trainer_log['training_time'] = time.time() - start
        self.trainer_log['best_epoch'] = best_epoch
        self.trainer_log['best_valid_loss'] = best_valid_loss

    # ------------------------------------------------------------------ evaluation
    def _message_passing_edges(self, data):
        mask = data.dtrain_mask if hasattr(data, 'dtrain_mask') else data.dr_mask
        return data.train_pos_edge_index[:, mask]

    # above this many Dr edges the 500 subsets are drawn on the device (see _ensure_df_subsets)
    FAST_SUBSETS_ABOVE = 1 << 18

    def _ensure_df_subsets(self, n_dr, k, dev=None):
        """500 random Dr subsets of size |Df|, drawn once and reused (base.py:263-268 upstream).

        Up to FAST_SUBSETS_ABOVE Dr edges they are drawn exactly as upstream does - 500 x torch.randperm(n_dr) on
        the host generator, kept as boolean masks in `self.df_pos_edge` - so the same seed gives the same subsets
        (tests/test_cli_gpu.py pins them against the reference).  Beyond that (ogbl-size graphs: 2 M Dr edges make
        those 500 host permutations 19 s of a 27 s unlearning request, and the masks 1 GB) the subsets come from
        torch.randperm on the device, seeded from the host generator: the same statistic - the mean over 500
        uniformly random |Df|-subsets of Dr - from a different random stream, kept as an index matrix only."""
        if len(self.df_pos_edge) > 0:
            return
        if n_dr <= self.FAST_SUBSETS_ABOVE or dev is None or dev.type != 'cuda':
            for _ in range(500):
                chosen = torch.zeros(n_dr, dtype=torch.bool)
                chosen[torch.randperm(n_dr)[:k]] = True
                self.df_pos_edge.append(chosen)
            return
        gen = torch.Generator(device=dev)
        gen.manual_seed(int(torch.randint(0, 2 ** 31 - 1, (1,))))
        self._df_subset_index = torch.stack([torch.randperm(n_dr, device=dev, generator=gen)[:k] for _ in range(500)])
        self.df_pos_edge = [None] * 500          # placeholders: only the count is used on this path

    @torch.no_grad()
    def eval(self, model, data, stage='val', pred_all=False):
        _require_gpu()
        model.eval()
        model = model.to(device)        # --eval_on_cpu existed to dodge GPU OOM; not needed with 288 GB
        data = data.to(device)
        pos_edge_index = data[f'{stage}_pos_edge_index']
        neg_edge_index = data[f'{stage}_neg_edge_index']
        z = model(data.x, self._message_passing_edges(data))
        logits = model.decode(z, pos_edge_index, neg_edge_index).sigmoid()
        label = self.get_link_labels(pos_edge_index, neg_edge_index)

        loss = F.binary_cross_entropy_with_logits(logits, label).cpu().item()
        # tensor AUC / AP (metrics.py): identical to scikit-learn's values, no host round trip
        dt_auc = float(batched_roc_auc(logits, label)[0])
        dt_aup = float(batched_average_precision(logits, label)[0])

        if self.args.unlearning_model in ['original']:
            df_logit = []
        else:
            df_logit = model.decode(z, data.directed_df_edge_index).sigmoid().tolist()

        if len(df_logit) > 0:
            dr_edges = data.train_pos_edge_index[:, data.dr_mask]
            self._ensure_df_subsets(dr_edges.shape[1], len(df_logit), z.device)
            # one decode over all of Dr instead of 500 decodes of subsets (same scores), then the 500
            # resampled AUC / AUP as ONE batched sort on the device: Df labelled 0, Dr labelled 1
            k = len(df_logit)
            dr_score = model.decode(z, dr_edges).sigmoid()
            if getattr(self, '_df_subset_index', None) is None or self._df_subset_index.device != dr_score.device:
                self._df_subset_index = torch.stack([c.nonzero().flatten() for c in self.df_pos_edge]).to(dr_score.device)
            df_score = torch.tensor(df_logit, dtype=dr_score.dtype, device=dr_score.device)
            scores = torch.cat([df_score[None].expand(len(self.df_pos_edge), k), dr_score[self._df_subset_index]], dim=1)
            labels = torch.cat([torch.zeros(k), torch.ones(k)]).to(dr_score.device)
            df_auc = float(batched_roc_auc(scores, labels).mean())
            df_aup = float(batched_average_precision(scores, labels).mean())
        else:
            df_auc = df_aup = np.nan

        logit_all_pair = _all_pairs(z) if pred_all else None
        log = {
            f'{stage}_loss': loss,
            f'{stage}_dt_auc': dt_auc,
            f'{stage}_dt_aup': dt_aup,
            f'{stage}_df_auc': df_auc,
            f'{stage}_df_aup': df_aup,
            f'{stage}_df_logit_mean': np.mean(df_logit) if len(df_logit) > 0 else np.nan,
            f'{stage}_df_logit_std': np.std(df_logit) if len(df_logit) > 0 else np.nan,
        }
        return loss, dt_auc, dt_aup, df_auc, df_aup, df_logit, logit_all_pair, log

    @torch.no_grad()
    def test(self, model, data, model_retrain=None, attack_model_all=None, attack_model_sub=None, ckpt='best'):
        if ckpt == 'best':
            state = torch.load(os.path.join(self.args.checkpoint_dir, 'model_best.pt'), map_location='cpu')
            model.load_state_dict(state['model_state'])
        pred_all = not is_large(self.args.dataset)       # N x N logits only for the small graphs (base.py:314-317)
        loss, dt_auc, dt_aup, df_auc, df_aup, df_logit, logit_all_pair, test_log = self.eval(model, data, 'test', pred_all)
        self.logit_all_pair = logit_all_pair
        self.trainer_log.update({
            'dt_loss': loss, 'dt_auc': dt_auc, 'dt_aup': dt_aup, 'df_logit': df_logit,
            'df_auc': df_auc, 'df_aup': df_aup, 'auc_sum': dt_auc + df_auc, 'aup_sum': dt_aup + df_aup,
            'auc_gap': abs(dt_auc - df_auc), 'aup_gap': abs(dt_aup - df_aup)})
        if model_retrain is not None:
            self.trainer_log['ve'] = verification_error(model, model_retrain).cpu().item()
        return loss, dt_auc, dt_aup, df_auc, df_aup, df_logit, logit_all_pair, test_log

    def save_log(self):
        with open(os.path.join(self.args.checkpoint_dir, 'trainer_log.json'), 'w') as f:
            json.dump(self.trainer_log, f)
        torch.save(self.logit_all_pair, os.path.join(self.args.checkpoint_dir, 'pred_proba.pt'))


class NodeClassificationTrainer(Trainer):
    """NLL node classification (base.py:694-864); eval = accuracy + micro-F1 on val_mask."""

    def train(self, model, data, optimizer, args):
        _require_gpu()
        start = time.time()
        best_epoch, best_valid_acc = 0, 0
        model = model.to(device)
        data = data.to(device)
        for epoch in range(args.epochs):
            model.train()
            z = F.log_softmax(model(data.x, data.edge_index), dim=1)
            loss = F.nll_loss(z[data.train_mask], data.y[data.train_mask])
            loss.backward()
            optimizer.step()
            optimizer.zero_grad()
            if (epoch + 1) % args.valid_freq == 0:
                valid_loss, dt_acc, dt_f1, valid_log = self.eval(model, data, 'val')
                self._record({'epoch': epoch, 'train_loss': loss.item()}, valid_log)
                if dt_acc > best_valid_acc:
                    best_valid_acc, best_epoch = dt_acc, epoch
                    print(f'Save best checkpoint at epoch {epoch:04d}. Valid Acc = {dt_acc:.4f}')
                    torch.save({'model_state': model.state_dict(), 'optimizer_state': optimizer.state_dict()},
                               os.path.join(args.checkpoint_dir, 'model_best.pt'))
                    torch.save(z.detach(), os.path.join(args.checkpoint_dir, 'node_embeddings.pt'))
        self.trainer_log['training_time'] = time.time() - start
        torch.save({'model_state': model.state_dict(), 'optimizer_state': optimizer.state_dict()},
                   os.path.join(args.checkpoint_dir, 'model_final.pt'))
        self.trainer_log['best_epoch'] = best_epoch
        self.trainer_log['best_valid_acc'] = best_valid_acc

    @torch.no_grad()
    def eval(self, model, data, stage='val', pred_all=False):
        _require_gpu()
        model.eval()
        model = model.to(device)
        data = data.to(device)
        z = F.log_softmax(model(data.x, data.edge_index), dim=1)
        sel = data.val_mask                       # upstream reads val_mask for every stage
        loss = F.nll_loss(z[sel], data.y[sel]).cpu().item()
        pred = torch.argmax(z[sel], dim=1).cpu()
        dt_acc = accuracy_score(data.y[sel].cpu(), pred)
        dt_f1 = f1_score(data.y[sel].cpu(), pred, average='micro')
        log = {f'{stage}_loss': loss, f'{stage}_dt_acc': dt_acc, f'{stage}_dt_f1': dt_f1}
        return loss, dt_acc, dt_f1, log

    @torch.no_grad()
    def test(self, model, data, model_retrain=None, attack_model_all=None, attack_model_sub=None, ckpt='best'):
        if ckpt == 'best':
            state = torch.load(os.path.join(self.args.checkpoint_dir, 'model_best.pt'), map_location='cpu')
            model.load_state_dict(state['model_state'])
        loss, dt_acc, dt_f1, test_log = self.eval(model, data, 'test', not is_large(self.args.dataset))
        self.trainer_log.update({'dt_loss': loss, 'dt_acc': dt_acc, 'dt_f1': dt_f1})
        if model_retrain is not None:
            self.trainer_log['ve'] = verification_error(model, model_retrain).cpu().item()
        return loss, dt_acc, dt_f1, test_log
