"""Small helpers the reference keeps in framework/utils.py (labels, KG negatives, seeding)."""
import random

import numpy as np
import torch


@torch.no_grad()
def get_link_labels(pos_edge_index, neg_edge_index):
    """1 for the leading positive columns, 0 for the negatives (utils.py:31-36)."""
    n_pos, n_neg = pos_edge_index.size(1), neg_edge_index.size(1)
    labels = torch.zeros(n_pos + n_neg, dtype=torch.float, device=pos_edge_index.device)
    labels[:n_pos] = 1.
    return labels


get_link_labels_kg = get_link_labels


@torch.no_grad()
def negative_sampling_kg(edge_index, edge_type):
    '''Generate negative samples but keep the node type the same: within every relation type
    the head column is shuffled with torch.randperm (utils.py:46-58; same RNG consumption
    order as upstream: ascending relation id).'''
    corrupted = edge_index.clone()
    for rel in edge_type.unique():
        sel = (edge_type == rel).nonzero().flatten()
        heads = corrupted[0, sel]
        corrupted[0, sel] = heads[torch.randperm(heads.shape[0])]
    return corrupted


def seed_everything(seed):
    """torch_geometric.seed.seed_everything (delete_gnn.py:62)."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
