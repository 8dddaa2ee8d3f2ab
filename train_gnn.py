#!/usr/bin/env python3
"""Train the ORIGINAL link-prediction backbone (reference CLI: train_gnn.py:18-80): writes
checkpoint/<dataset>/<gnn>/original/<seed>/{model_best.pt, node_embeddings.pt, pred_proba.pt,
trainer_log.json, training_args.json}, which delete_gnn.py starts from."""
import os

import torch

from gnndelete_amd.framework import get_model, get_trainer
from gnndelete_amd.framework.data import Data
from gnndelete_amd.framework.graph_utils import is_undirected, to_undirected
from gnndelete_amd.framework.trainer._log import wandb_init
from gnndelete_amd.framework.training_args import parse_args
from gnndelete_amd.framework.utils import seed_everything

device = torch.device('cuda' if torch.cuda.is_available() else 'cpu')


def main():
    args = parse_args()
    args.unlearning_model = 'original'
    args.checkpoint_dir = os.path.join(args.checkpoint_dir, args.dataset, args.gnn, args.unlearning_model,
                                       str(args.random_seed))
    os.makedirs(args.checkpoint_dir, exist_ok=True)
    seed_everything(args.random_seed)

    data = Data.load(os.path.join(args.data_dir, args.dataset, f'd_{args.random_seed}.pt'))
    print('Directed dataset:', data)
    if args.gnn not in ['rgcn', 'rgat']:
        args.in_dim = data.num_features
    wandb_init(args)

    if args.gnn in ['rgcn', 'rgat']:
        # reverse edges carry relation type + R (train_gnn.py:46-56)
        rev = data.train_pos_edge_index.flip(0)
        data.edge_index = torch.cat([data.train_pos_edge_index, rev], dim=1)
        data.edge_type = torch.cat([data.train_edge_type, data.train_edge_type + args.num_edge_type], dim=0)
        data.dr_mask = torch.ones(data.edge_index.shape[1], dtype=torch.bool)
        assert is_undirected(data.edge_index, data.num_nodes)
    else:
        data.train_pos_edge_index = to_undirected(data.train_pos_edge_index, num_nodes=data.num_nodes)
        data.dtrain_mask = torch.ones(data.train_pos_edge_index.shape[1], dtype=torch.bool)
        data.dr_mask = data.dtrain_mask
        assert is_undirected(data.train_pos_edge_index, data.num_nodes)
    print('Undirected dataset:', data)

    model = get_model(args, num_nodes=data.num_nodes, num_edge_type=args.num_edge_type).to(device)
    optimizer = torch.optim.Adam(model.parameters(), lr=args.lr)
    trainer = get_trainer(args)
    trainer.train(model, data, optimizer, args)
    trainer.test(model, data)
    trainer.save_log()


if __name__ == '__main__':
    main()
