"""Data container + the unlearning-request preprocessing of the reference CLI.

``Data`` is the attribute bag the trainers consume (the reference uses torch_geometric.data.Data,
whose pickles cannot be read without PyG): attribute and ``data['key']`` access, ``.to(device)``,
AttributeError for missing keys so ``hasattr(data, 'dtrain_mask')`` behaves as upstream
(base.py:238).  On disk it is a plain ``torch.save``'d dict of tensors."""
import torch

from .graph_utils import is_undirected, k_hop_subgraph, to_undirected


class Data(dict):
    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError:
            raise AttributeError(key) from None

    def __setattr__(self, key, value):
        self[key] = value

    def __delattr__(self, key):
        del self[key]

    def to(self, device, *args, **kwargs):
        for k, v in self.items():
            if torch.is_tensor(v):
                self[k] = v.to(device, *args, **kwargs)
        return self

    def cpu(self):
        return self.to('cpu')

    def cuda(self):
        return self.to('cuda')

    def clone(self):
        return Data({k: (v.clone() if torch.is_tensor(v) else v) for k, v in self.items()})

    def save(self, path):
        torch.save(dict(self), path)

    @staticmethod
    def load(path):
        return Data(torch.load(path, map_location='cpu'))

    def __repr__(self):
        parts = [f'{k}={list(v.shape)}' if torch.is_tensor(v) else f'{k}={v}' for k, v in self.items()]
        return 'Data(' + ', '.join(parts) + ')'


def resolve_df_size(df_size, num_train_edges):
    """--df_size >= 100 is a count, otherwise a percentage of the train edges (delete_gnn.py:88-91)."""
    return int(df_size) if df_size >= 100 else int(df_size / 100 * num_train_edges)


def prepare_edge_deletion(data, df_mask_all, df_size, relational=False, num_edge_type=None):
    """delete_gnn.py:85-189: pick Df among the candidates with torch.randperm (global RNG, as
    upstream), build the 2-hop / 1-hop enclosing-subgraph masks on the DIRECTED train edges,
    then symmetrise edges and masks for message passing.  Mutates and returns ``data`` with
    df_mask, dr_mask, sdf_mask, sdf_node_{1,2}hop_mask, directed_df_edge_index[, _type],
    edge_index[, edge_type]."""
    E = data.train_pos_edge_index
    n = data.num_nodes
    candidates = df_mask_all.nonzero().squeeze()
    chosen = candidates[torch.randperm(candidates.shape[0])[:df_size]]
    df_mask = torch.zeros(E.shape[1], dtype=torch.bool)
    df_mask[chosen] = True
    dr_mask = ~df_mask

    data.directed_df_edge_index = E[:, df_mask]
    if relational:
        data.directed_df_edge_type = data.train_edge_type[df_mask]

    seeds = E[:, df_mask].flatten().unique()
    _, two_hop_edge, _, two_hop_mask = k_hop_subgraph(seeds, 2, E, num_nodes=n)
    _, one_hop_edge, _, _ = k_hop_subgraph(seeds, 1, E, num_nodes=n)
    sdf_node_1hop = torch.zeros(n, dtype=torch.bool)
    sdf_node_2hop = torch.zeros(n, dtype=torch.bool)
    sdf_node_1hop[one_hop_edge.flatten().unique()] = True
    sdf_node_2hop[two_hop_edge.flatten().unique()] = True
    data.sdf_node_1hop_mask = sdf_node_1hop
    data.sdf_node_2hop_mask = sdf_node_2hop

    assert not is_undirected(E, n)
    if relational:
        rev = E.flip(0)
        data.edge_index = torch.cat([E, rev], dim=1)
        data.edge_type = torch.cat([data.train_edge_type, data.train_edge_type + num_edge_type], dim=0)
        if 'train_mask' in data:
            data.train_mask = data.train_mask.repeat(2).view(-1)
        two_hop_mask, df_mask, dr_mask = (m.repeat(2).view(-1) for m in (two_hop_mask, df_mask, dr_mask))
    else:
        und, (df_i, hop_i) = to_undirected(E, [df_mask.int(), two_hop_mask.int()], n)
        two_hop_mask, df_mask = hop_i.bool(), df_i.bool()
        dr_mask = ~df_mask
        data.train_pos_edge_index = und
        data.edge_index = und
        assert is_undirected(und, n)
    data.sdf_mask = two_hop_mask
    data.df_mask = df_mask
    data.dr_mask = dr_mask
    return data
