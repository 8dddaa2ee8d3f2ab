#!/usr/bin/env python3
"""Train the original node classifier (reference CLI: train_node.py:19-54) on a synthetic
node-classification stand-in; writes checkpoint_node/<dataset>/<gnn>/original/<seed>/."""
import os

import torch

from gnndelete_amd.framework import get_model, get_trainer
from gnndelete_amd.framework.synth import make_nodecls_dataset
from gnndelete_amd.framework.training_args import parse_args
from gnndelete_amd.framework.utils import seed_everything

device = torch.device('cuda' if torch.cuda.is_available() else 'cpu')


def main():
    args = parse_args()
    args.unlearning_model = 'original_node'
    args.checkpoint_dir = os.path.join('checkpoint_node', args.dataset, args.gnn, 'original', str(args.random_seed))
    os.makedirs(args.checkpoint_dir, exist_ok=True)
    seed_everything(args.random_seed)
    data = make_nodecls_dataset(args.dataset, seed=args.random_seed)
    args.in_dim, args.out_dim = data.x.shape[1], data.num_classes
    model = get_model(args, num_nodes=data.num_nodes, num_edge_type=args.num_edge_type).to(device)
    optimizer = torch.optim.Adam(model.parameters(), lr=args.lr)
    trainer = get_trainer(args)
    trainer.train(model, data, optimizer, args)
    trainer.test(model, data)
    trainer.save_log()


if __name__ == '__main__':
    main()
