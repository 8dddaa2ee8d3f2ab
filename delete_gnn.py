#!/usr/bin/env python3
"""Edge unlearning CLI (reference: delete_gnn.py:57-283), same flags, same checkpoint layout:

  python delete_gnn.py --dataset synth-dblp --gnn gcn --unlearning_model gnndelete_nodeemb \
         --df out --df_size 2.5 --random_seed 42

Loads data/<dataset>/d_<seed>.pt + df_<seed>.pt, samples Df, builds the S_Df masks, loads the
original backbone (strict=False: the Del weights keep their ones/1000 initialisation), trains the
Del operators on the HIP engine, evaluates and writes trainer_log.json / pred_proba.pt."""
import copy
import os

import torch

from gnndelete_amd.framework import get_model, get_trainer
from gnndelete_amd.framework.data import Data, prepare_edge_deletion, resolve_df_size
from gnndelete_amd.framework.trainer._log import wandb_init
from gnndelete_amd.framework.training_args import parse_args
from gnndelete_amd.framework.utils import seed_everything

device = torch.device('cuda' if torch.cuda.is_available() else 'cpu')


def build_optimizer(model, args):
    """delete_gnn.py:215-241: Adam over the parameters whose name contains 'del' (two Adams for
    the layer-wise loss types), Adam over everything for the non-GNNDelete methods."""
    if 'gnndelete' in args.unlearning_model:
        dels = [p for n, p in model.named_parameters() if 'del' in n]
        print('parameters_to_optimize', [n for n, p in model.named_parameters() if 'del' in n])
        relational = args.gnn in ['rgcn', 'rgat']
        if ('nodeemb' in args.unlearning_model or relational) and 'layerwise' in args.loss_type:
            return [torch.optim.Adam(model.deletion1.parameters(), lr=args.lr),
                    torch.optim.Adam(model.deletion2.parameters(), lr=args.lr)]
        return torch.optim.Adam([{'params': dels, 'weight_decay': 0.0}], lr=args.lr)
    return torch.optim.Adam([{'params': list(model.parameters()), 'weight_decay': 0.0}], lr=args.lr)


def main():
    args = parse_args()
    base_ckpt = args.checkpoint_dir
    original_path = os.path.join(base_ckpt, args.dataset, args.gnn, 'original', str(args.random_seed))
    seed_everything(args.random_seed)

    tail = '-'.join(str(i) for i in [args.df, args.df_size, args.random_seed])
    if 'gnndelete' in args.unlearning_model:
        variant = '-'.join(str(i) for i in [args.loss_fct, args.loss_type, args.alpha, args.neg_sample_random])
        args.checkpoint_dir = os.path.join(base_ckpt, args.dataset, args.gnn, args.unlearning_model, variant, tail)
    else:
        args.checkpoint_dir = os.path.join(base_ckpt, args.dataset, args.gnn, args.unlearning_model, tail)
    # Launched under torch.distributed (python -m torch.distributed.run --nproc-per-node N delete_gnn.py ... --minibatch): the
    # mini-batch loops are batch-data-parallel - every rank draws its own GraphSAINT batches, the Del gradients are averaged
    # before each optimizer step (framework/trainer/sampler.py: sync_gradients); rank 0 owns the checkpoint directory
    world, rank = int(os.environ.get('WORLD_SIZE', 1)), int(os.environ.get('RANK', 0))
    if world > 1:
        import torch.distributed as dist
        if torch.cuda.is_available() and torch.cuda.device_count() > 1:
            torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', 0)) % torch.cuda.device_count())
        dist.init_process_group(os.environ.get('GNNDELETE_DIST_BACKEND', 'nccl'))
        if rank:
            args.checkpoint_dir = os.path.join(args.checkpoint_dir, f'rank{rank}')
    os.makedirs(args.checkpoint_dir, exist_ok=True)

    data = Data.load(os.path.join(args.data_dir, args.dataset, f'd_{args.random_seed}.pt'))
    print('Directed dataset:', data)
    relational = args.gnn in ['rgcn', 'rgat']
    if not relational:
        args.in_dim = data.num_features
    print('Training args', args)
    wandb_init(args)

    assert args.df != 'none'
    df_size = resolve_df_size(args.df_size, data.train_pos_edge_index.shape[1])
    print(f'Original size: {data.train_pos_edge_index.shape[1]:,}')
    print(f'Df size: {df_size:,}')
    df_mask_all = torch.load(os.path.join(args.data_dir, args.dataset, f'df_{args.random_seed}.pt'))[args.df]
    prepare_edge_deletion(data, df_mask_all, df_size, relational, args.num_edge_type)
    print('Undirected dataset:', data)

    model = get_model(args, data.sdf_node_1hop_mask, data.sdf_node_2hop_mask, num_nodes=data.num_nodes,
                      num_edge_type=args.num_edge_type)
    logits_ori = None
    if args.unlearning_model != 'retrain':
        proba = os.path.join(original_path, 'pred_proba.pt')
        if os.path.exists(proba):
            logits_ori = torch.load(proba, map_location='cpu')
            if logits_ori is not None:
                logits_ori = logits_ori.to(device)
        ckpt = torch.load(os.path.join(original_path, 'model_best.pt'), map_location='cpu')
        model.load_state_dict(ckpt['model_state'], strict=False)
    else:
        data.dtrain_mask = data.dr_mask
    model = model.to(device)
    optimizer = build_optimizer(model, args)

    trainer = get_trainer(args)
    if args.unlearning_model == 'retrain':
        trainer.train(model, data, optimizer, args)
    else:
        trainer.train(model, data, optimizer, args, logits_ori, None, None)

    retrain = None
    if args.unlearning_model != 'retrain':
        retrain_path = os.path.join(base_ckpt, args.dataset, args.gnn, 'retrain', tail, 'model_best.pt')
        if os.path.exists(retrain_path):
            retrain_args = copy.deepcopy(args)
            retrain_args.unlearning_model = 'retrain'
            retrain = get_model(retrain_args, num_nodes=data.num_nodes, num_edge_type=args.num_edge_type)
            retrain.load_state_dict(torch.load(retrain_path, map_location='cpu')['model_state'])
            retrain = retrain.to(device).eval()
    results = trainer.test(model, data, model_retrain=retrain)
    print(results[-1])
    trainer.save_log()


if __name__ == '__main__':
    main()
